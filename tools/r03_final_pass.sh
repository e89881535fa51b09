#!/bin/bash
# Round-3 measurement pass (GPU box, repo root): smoke, GPU tests, bench lines, rocprof summaries, PMC.  Output: gpurun_out/r3final/
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2>/dev/null; echo "bench driver command rc=$?"
for w in blox4096 mazes8192; do timeout -k 10 900 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; echo "bench $w rc=$?"; done
timeout -k 10 300 python bench.py --workload tiny4096 --no-ppo > $O/bench_tiny4096.json 2>/dev/null
timeout -k 10 300 python bench.py --obs bfloat16 --no-ppo --no-emit > $O/bench_small_bfloat16.json 2>/dev/null
timeout -k 10 300 python bench.py --obs uint8 --no-ppo --no-emit > $O/bench_small_uint8.json 2>/dev/null
echo benches done
bash tools/tick_prof.sh $O
bash tools/r03_prof.sh r3final
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_actor -- python3 $ROOT/tools/actor_bench.py --batch 8192 --iters 5 --no-library > /tmp/pmc_actor.log 2>&1 || echo "pmc actor failed"
python3 $ROOT/tools/pmc_mfma.py /tmp/pmc_actor > $ROOT/$O/pmc_mfma_actor_critic.md 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_actor_b -- python3 $ROOT/tools/actor_bench.py --batch 8192 --layout bloxCapture --iters 5 --no-library > /tmp/pmc_actor_b.log 2>&1 || echo "pmc actor blox failed"
python3 $ROOT/tools/pmc_mfma.py /tmp/pmc_actor_b >> $ROOT/$O/pmc_mfma_actor_critic.md 2>&1 || true
for S in 154 400; do
  rm -rf /tmp/pmc_attn_$S
  ATTN_S=$S timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_attn_$S -- python3 $ROOT/tools/attn_time.py > /tmp/pmc_attn_$S.log 2>&1 || echo "pmc attn failed"
  python3 $ROOT/tools/pmc_mfma.py /tmp/pmc_attn_$S >> $ROOT/$O/pmc_mfma_actor_critic.md 2>&1 || true
done
cd $ROOT
for S in 154 400; do ATTN_S=$S python tools/attn_time.py 2>/dev/null; done > $O/attn_time.txt
for lay in smallCapture bloxCapture; do python tools/actor_bench.py --batch 512 8192 --layout $lay --iters 10 --no-library 2>/dev/null; done > $O/actor_bench.txt
bash tools/r03_pmc1.sh > /dev/null 2>&1; cp gpurun_out/r3pmc1/pmc_attn_S400.txt $O/pmc_wave_cycles_attention_S400.txt; cp gpurun_out/r3pmc1/pmc_actor_blox.txt $O/pmc_wave_cycles_actor_blox.txt
rm -f gpurun_out/traffic_new.json
bash tools/pmc_pass.sh small16384 float32 f32 > $O/pmc_f32.log 2>&1 || echo "pmc f32 failed"
cp gpurun_out/traffic_new.json $O/traffic.json 2>/dev/null || true
echo done
