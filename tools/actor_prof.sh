#!/bin/bash
# usage (GPU box, repo root): tools/actor_prof.sh [batch]   -> per-kernel average durations of the fused actor tower
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-8192}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_actor
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_actor -- python3 $ROOT/tools/actor_bench.py --batch $B --iters 20 > /tmp/prof_actor.log 2>&1
cd $ROOT
python tools/prof_summary.py $(find /tmp/prof_actor -name "*kernel_stats.csv" | head -1) "pmx_actor" 12
