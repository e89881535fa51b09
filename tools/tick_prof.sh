#!/bin/bash
# usage (GPU box, repo root): tools/tick_prof.sh <outdir>  -> rocprofv3 kernel stats of the all-bytes-to-HBM tick AND the bench line of
# that very process (its HIP-event average for the expansion kernel), so that the two averages can be compared run for run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-gpurun_out/tickprof}
mkdir -p $ROOT/$O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tick
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tick -- python3 $ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-ppo --no-unidirectional --fixed-sweep > /tmp/prof_tick.log 2>&1
python3 $ROOT/tools/prof_summary.py $(find /tmp/prof_tick -name "*kernel_stats.csv" | head -1) "" 12 > $ROOT/$O/kernel_stats_tick_small16384_f32_fixed_sweep.txt
cp $(find /tmp/prof_tick -name "*kernel_stats.csv" | head -1) $ROOT/$O/kernel_stats_tick_small16384_f32_fixed_sweep.csv
grep '^{"metric"' /tmp/prof_tick.log | tail -1 > $ROOT/$O/bench_line_of_the_profiled_run.json
head -2 $ROOT/$O/kernel_stats_tick_small16384_f32_fixed_sweep.txt
python3 -c "
import json
d=json.loads(open('$ROOT/$O/bench_line_of_the_profiled_run.json').read())
r=d['roofline']; print('bench line of the same process: expand avg_launch_us', r['avg_launch_us'], 'rule', r['rule_kernel_avg_us'], 'frac', r['frac'])
"
