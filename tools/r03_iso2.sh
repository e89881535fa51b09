#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cfg in "X=1" "PMX_NO_FUSED_PROJECTOR=1" "PMX_NO_DEFER_SUMS=1" "PMX_NO_PREPACK=1" "PMX_NO_FUSED_PROJECTOR=1 PMX_NO_DEFER_SUMS=1 PMX_NO_PREPACK=1"; do
  env $cfg timeout -k 10 200 python tools/step512_time.py 2>/dev/null | tail -1
done
