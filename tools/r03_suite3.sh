#!/bin/bash
# the GPU test suite three times over: flaky tests show up as differing outcomes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r3suite
for i in 1 2 3; do
  timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3suite/run$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/r3suite/run$i.log)"; grep -E "^FAILED" gpurun_out/r3suite/run$i.log | cut -c1-200
done
