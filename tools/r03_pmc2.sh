#!/bin/bash
# wave-cycle split (SQ counters) and memory-side counters of the attention kernels at 154 tokens
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3pmc2; mkdir -p $O
cd /tmp
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
ATTN_S=154 ATTN_B=8192 timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_a -- python3 $ROOT/tools/attn_time.py > /tmp/pmc_a.log 2>&1 || tail -3 /tmp/pmc_a.log
ATTN_S=154 ATTN_B=8192 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/pmc_b -- python3 $ROOT/tools/attn_time.py > /tmp/pmc_b.log 2>&1 || tail -3 /tmp/pmc_b.log
ATTN_S=154 ATTN_B=8192 timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_c -- python3 $ROOT/tools/attn_time.py > /tmp/pmc_c.log 2>&1 || tail -3 /tmp/pmc_c.log
cd $ROOT
(python tools/pmc_counters.py /tmp/pmc_a pmx_attn; python tools/pmc_counters.py /tmp/pmc_b pmx_attn; python tools/pmc_counters.py /tmp/pmc_c pmx_attn) > $O/pmc_attn_S154.txt
cat $O/pmc_attn_S154.txt
