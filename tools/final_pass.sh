set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -1 gpurun_out/final/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
for w in tiny4096 blox4096 mazes4096 mazes8192; do timeout -k 10 300 python bench.py --workload $w --no-ppo > gpurun_out/final/bench_$w.json 2>/dev/null; done
timeout -k 10 300 python bench.py --obs bfloat16 --no-ppo > gpurun_out/final/bench_small_bfloat16.json 2>/dev/null
timeout -k 10 300 python bench.py --obs uint8 --no-ppo > gpurun_out/final/bench_small_uint8.json 2>/dev/null
timeout -k 10 300 python bench.py --envs 65536 --no-ppo > gpurun_out/final/bench_small65536.json 2>/dev/null
timeout -k 10 300 python bench.py --workload blox4096 --envs 16384 --no-ppo > gpurun_out/final/bench_blox16384.json 2>/dev/null
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o fin -- python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-ppo --no-unidirectional > /tmp/prof_final.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/prof_final -name "*kernel_stats.csv" | head -1) gpurun_out/final/kernel_stats_small16384_f32.csv
bash tools/pmc_pass.sh small16384 float32 fin > gpurun_out/final/pmc.log 2>&1 || echo "pmc pass failed"
cp gpurun_out/traffic_new.json gpurun_out/final/traffic_small16384_f32.json 2>/dev/null || true
echo done
