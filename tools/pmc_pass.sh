#!/bin/bash
# usage: tools/pmc_pass.sh <workload> <obs> <tag>   (run on the GPU box from the repo root)
# Separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slots), --kernel-trace only, then tools/pmc_traffic.py.
set -e
W=$1; O=$2; TAG=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$c -- python3 $ROOT/bench.py --workload $W --obs $O --steps 40 --warmup 10 --no-cpu-baseline --no-ppo --no-unidirectional --fixed-sweep > /tmp/pmc_${TAG}_$c.log 2>&1
done
cd $ROOT
python tools/pmc_traffic.py /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE $W:$O gpurun_out/traffic_new.json
# the training loop's observation kernel (byte planes) runs in the same bench process (roofline_emit_team)
python tools/pmc_traffic.py /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE $W:emit_uint8 gpurun_out/traffic_new.json pmx_emit_team
