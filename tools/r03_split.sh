#!/bin/bash
# the data-gradient kernel at large batches: one wave per sample (one wave per SIMD) against two waves per sample at two waves per SIMD
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3split; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_actor_tower.py -q -x > $O/pytest_tower.log 2>&1; echo "tower rc=$?"; tail -3 $O/pytest_tower.log
run() { timeout -k 10 200 python tools/actor_bench.py --batch 1024 2048 8192 16384 --layout smallCapture --iters 20 --no-library 2>/dev/null | cut -c1-400; }
{
echo "== one wave per sample (PMX_ACTOR_BWD_TWO_WAVE_MIN=0)"; PMX_ACTOR_BWD_TWO_WAVE_MIN=0 run
echo "== two waves per sample, two waves per SIMD (default)"; run
} > $O/ab3.txt 2>&1
cat $O/ab3.txt
