#!/bin/bash
# Round-3 closing pass, part A (GPU box, repo root): smoke, GPU tests, the bench lines.  Output: gpurun_out/r3f2/
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3f2; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
SECONDS=0
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "driver command rc=$? ${SECONDS}s"
SECONDS=0
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$? ${SECONDS}s"
for w in blox4096 mazes8192 tiny4096; do timeout -k 10 600 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; echo "bench $w rc=$?"; done
timeout -k 10 300 python bench.py --obs bfloat16 --no-ppo --no-emit --no-cpu-baseline > $O/bench_small_bfloat16.json 2>/dev/null
timeout -k 10 300 python bench.py --obs uint8 --no-ppo --no-emit --no-cpu-baseline > $O/bench_small_uint8.json 2>/dev/null
echo done
