#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3ab1; mkdir -p $O
bash tools/ab_lib.sh "python tools/actor_bench.py --batch 8192 --layout bloxCapture --iters 10 --no-library 2>/dev/null | cut -c1-260; python tools/actor_bench.py --batch 8192 --layout smallCapture --iters 10 --no-library 2>/dev/null | cut -c1-260" base noslp_actor > $O/actor_ab.txt 2>&1
bash tools/ab_lib.sh "ATTN_S=154 python tools/attn_time.py 2>/dev/null; ATTN_S=400 python tools/attn_time.py 2>/dev/null" base noslp_train > $O/attn_ab.txt 2>&1
cat $O/actor_ab.txt $O/attn_ab.txt
timeout -k 10 300 python -m pytest tests/test_gpu_trainer.py -q -k "reference_fixture" > $O/pytest_fixture.log 2>&1; echo "fixture rc=$?"; grep -E "^E  |passed|failed" $O/pytest_fixture.log | head -20
