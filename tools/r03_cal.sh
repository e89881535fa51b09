#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 300 python -m pytest tests/test_gpu_trainer.py -q -k "reference_fixture" > gpurun_out/r3d/pytest_fixture.log 2>&1; echo "fixture rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r3d/pytest_fixture.log | head -20
bash tools/r03_prof.sh r3prof1
