#!/bin/bash
# usage (GPU box, repo root): tools/step_prof.sh [minibatch] [top]  -> per-kernel totals of MAPPO updates (config-3 shape, fewer envs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
MB=${1:-16384}; TOP=${2:-45}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_step
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python3 $ROOT/tools/train_bench.py --envs 8192 --horizon 16 --minibatch $MB --updates 3 > /tmp/prof_step.log 2>&1
tail -2 /tmp/prof_step.log
cd $ROOT
python tools/prof_summary.py $(find /tmp/prof_step -name "*kernel_stats.csv" | head -1) "${3:-}" $TOP ${4:-}
