#!/bin/bash
# Round-3 first pass on the GPU box: the new 20x20 tower tests, the graph / data-parallel tests, a 20x20 end-to-end figure and
# its kernel profile.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_actor_tower.py -x -q > $O/pytest_tower.log 2>&1; echo "tower rc=$?"; tail -3 $O/pytest_tower.log
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -x -q -k "graph or end_to_end or shadow" > $O/pytest_trainer.log 2>&1; echo "trainer rc=$?"; tail -3 $O/pytest_trainer.log
timeout -k 10 400 python tools/train_bench.py --layout mazes --envs 4096 --horizon 32 --minibatch 16384 --updates 2 > $O/train_mazes4096.json 2> $O/train_mazes4096.err; echo "train mazes rc=$?"; tail -c 600 $O/train_mazes4096.json
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -- python3 $ROOT/tools/train_bench.py --layout mazes --envs 2048 --horizon 16 --minibatch 16384 --updates 2 > /tmp/prof_m.log 2>&1; echo "prof rc=$?"
python3 $ROOT/tools/prof_summary.py $(find /tmp/prof_m -name "*kernel_stats.csv" | head -1) "" 70 > $ROOT/$O/kernel_stats_train_step_mazes_mb16384.txt
tail -1 /tmp/prof_m.log > $ROOT/$O/train_bench_of_the_profiled_run_mazes.json
cd $ROOT
head -40 $O/kernel_stats_train_step_mazes_mb16384.txt
