#!/bin/bash
# the bench lines of the round's final code (default run, the driver's command, the two 20x20 workloads)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3lines; mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2>/dev/null; echo "driver command rc=$?"
for w in blox4096 mazes8192; do timeout -k 10 900 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; echo "bench $w rc=$?"; done
wc -l $O/*.json
