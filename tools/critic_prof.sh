#!/bin/bash
# usage (GPU box, repo root): tools/critic_prof.sh [batch]  -> per-kernel averages of the critic's hand-written kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-8192}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_critic3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_critic3 -- python3 $ROOT/tools/critic_bench.py --batch $B --iters 10 > /tmp/prof_critic3.log 2>&1
cd $ROOT
python tools/prof_summary.py $(find /tmp/prof_critic3 -name "*kernel_stats.csv" | head -1) "${2:-pmx_(ffn|tok)}" 8
