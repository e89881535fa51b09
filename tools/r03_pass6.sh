#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_actor_tower.py -x -q > $O/pytest_tower.log 2>&1; echo "tower rc=$?"; tail -2 $O/pytest_tower.log
for lay in bloxCapture smallCapture; do python tools/actor_bench.py --batch 512 8192 --layout $lay --iters 10 --no-library 2>/dev/null | cut -c1-200; done | tee $O/actor_bench.txt
