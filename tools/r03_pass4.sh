#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -x -q -k "attention or critic" > $O/pytest_attn.log 2>&1; echo "attn tests rc=$?"; tail -2 $O/pytest_attn.log
for S in 154 400; do ATTN_S=$S timeout -k 10 120 python tools/attn_time.py 2>/dev/null; done | tee $O/attn_time.txt
