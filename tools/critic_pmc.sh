#!/bin/bash
# usage (GPU box, repo root): tools/critic_pmc.sh [batch]  -> wave-cycle split (SQ counters) and durations of the critic's kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-8192}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_critic2 /tmp/prof_critic2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_critic2 -- python3 $ROOT/tools/critic_bench.py --batch $B --iters 3 > /tmp/pmc_critic2.log 2>&1 || tail -5 /tmp/pmc_critic2.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_critic2 -- python3 $ROOT/tools/critic_bench.py --batch $B --iters 10 > /tmp/prof_critic2.log 2>&1
cd $ROOT
python tools/pmc_counters.py /tmp/pmc_critic2 pmx_
python tools/prof_summary.py $(find /tmp/prof_critic2 -name "*kernel_stats.csv" | head -1) "pmx_" 14
