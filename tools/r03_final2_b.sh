#!/bin/bash
# Round-3 closing pass, part B (GPU box, repo root): rocprofv3 summaries of the tick and of the optimizer steps, PMC traffic.
# Output: gpurun_out/r3f2/
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3f2; mkdir -p $O
bash tools/tick_prof.sh $O
rm -f gpurun_out/traffic_new.json
bash tools/pmc_pass.sh small16384 float32 f32 > $O/pmc_f32.log 2>&1 || echo "pmc f32 failed"
cp gpurun_out/traffic_new.json $O/traffic.json 2>/dev/null || true
echo pmc done
bash tools/r03_prof.sh r3f2
echo done
