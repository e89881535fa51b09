#!/bin/bash
# Round-2 measurement pass (run on the GPU box from the repo root): smoke, GPU tests, bench lines, rocprof summaries, PMC.
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r2final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench default rc=$?
for w in tiny4096 blox4096 mazes4096; do timeout -k 10 300 python bench.py --workload $w --no-ppo > $O/bench_$w.json 2>/dev/null; done
timeout -k 10 300 python bench.py --obs bfloat16 --no-ppo > $O/bench_small_bfloat16.json 2>/dev/null
timeout -k 10 300 python bench.py --obs uint8 --no-ppo > $O/bench_small_uint8.json 2>/dev/null
timeout -k 10 300 python bench.py --envs 65536 --no-ppo > $O/bench_small65536.json 2>/dev/null
echo benches done
bash tools/tick_prof.sh $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python3 $ROOT/tools/train_bench.py --envs 8192 --horizon 16 --minibatch 16384 --updates 3 > /tmp/prof_step.log 2>&1
python3 $ROOT/tools/prof_summary.py $(find /tmp/prof_step -name "*kernel_stats.csv" | head -1) "" 60 > $ROOT/$O/kernel_stats_train_step_mb16384.txt
tail -1 /tmp/prof_step.log > $ROOT/$O/train_bench_of_the_profiled_run.json
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_actor -- python3 $ROOT/tools/actor_bench.py --batch 8192 --iters 5 > /tmp/pmc_actor.log 2>&1 || echo "pmc actor failed"
python3 $ROOT/tools/pmc_mfma.py /tmp/pmc_actor > $ROOT/$O/pmc_mfma_actor_critic.md 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_critic -- python3 $ROOT/tools/critic_bench.py --iters 5 > /tmp/pmc_critic.log 2>&1 || echo "pmc critic failed"
python3 $ROOT/tools/pmc_mfma.py /tmp/pmc_critic >> $ROOT/$O/pmc_mfma_actor_critic.md 2>&1 || true
cd $ROOT
tools/actor_prof.sh 8192 > $O/actor_kernels_8192.txt 2>&1 || echo "actor prof failed"
tools/actor_pmc.sh 8192 > $O/pmc_actor_wave_cycles.txt 2>&1 || echo "actor pmc failed"
tools/critic_pmc.sh 8192 > $O/pmc_critic_wave_cycles.txt 2>&1 || echo "critic pmc failed"
tools/step512_prof.sh 80 > $O/kernel_stats_train_step_mb512_graph.txt 2>&1 || echo "step512 prof failed"
rm -f gpurun_out/traffic_new.json
bash tools/pmc_pass.sh small16384 float32 f32 > $O/pmc_f32.log 2>&1 || echo "pmc f32 failed"
bash tools/pmc_pass.sh small16384 uint8 u8 > $O/pmc_u8.log 2>&1 || echo "pmc u8 failed"
bash tools/pmc_pass.sh small16384 bfloat16 bf16 > $O/pmc_bf16.log 2>&1 || echo "pmc bf16 failed"
cp gpurun_out/traffic_new.json $O/traffic.json 2>/dev/null || true
echo done
