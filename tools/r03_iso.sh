#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3iso; mkdir -p $O
for cfg in "PMX_PREPACK_NO_RECORD=1" "PMX_PREPACK_NO_RECORD=1 PMX_NO_TWO_STREAMS=1" "PMX_NO_TWO_STREAMS=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python -m pytest tests/test_gpu_trainer.py -x -q -k "past_230 or graph_trainer" > $O/log.txt 2>&1; echo "rc=$?"; tail -1 $O/log.txt
done
