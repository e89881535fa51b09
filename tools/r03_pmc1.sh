#!/bin/bash
# wave-cycle split (SQ counters) of the attention kernels at 400 tokens and of the 28-tile tower kernels
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3pmc1; mkdir -p $O
cd /tmp
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c /tmp/pmc_d
ATTN_S=400 ATTN_B=8192 timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_a -- python3 $ROOT/tools/attn_time.py > /tmp/pmc_a.log 2>&1 || tail -3 /tmp/pmc_a.log
ATTN_S=400 ATTN_B=8192 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/pmc_b -- python3 $ROOT/tools/attn_time.py > /tmp/pmc_b.log 2>&1 || tail -3 /tmp/pmc_b.log
timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_c -- python3 $ROOT/tools/actor_bench.py --batch 8192 --layout bloxCapture --iters 3 --no-library > /tmp/pmc_c.log 2>&1 || tail -3 /tmp/pmc_c.log
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/pmc_d -- python3 $ROOT/tools/actor_bench.py --batch 8192 --layout bloxCapture --iters 3 --no-library > /tmp/pmc_d.log 2>&1 || tail -3 /tmp/pmc_d.log
cd $ROOT
(python tools/pmc_counters.py /tmp/pmc_a pmx_attn; python tools/pmc_counters.py /tmp/pmc_b pmx_attn) > $O/pmc_attn_S400.txt
(python tools/pmc_counters.py /tmp/pmc_c pmx_actor; python tools/pmc_counters.py /tmp/pmc_d pmx_actor) > $O/pmc_actor_blox.txt
cat $O/pmc_attn_S400.txt $O/pmc_actor_blox.txt
