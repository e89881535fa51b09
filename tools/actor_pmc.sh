#!/bin/bash
# usage (GPU box, repo root): tools/actor_pmc.sh [batch]   -> wave-cycle split of the fused actor tower's kernels (SQ counters)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-8192}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_actor2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_actor2 -- python3 $ROOT/tools/actor_bench.py --batch $B --iters 3 > /tmp/pmc_actor2.log 2>&1 || tail -5 /tmp/pmc_actor2.log
cd $ROOT
python tools/pmc_counters.py /tmp/pmc_actor2 pmx_actor
