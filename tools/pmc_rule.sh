#!/bin/bash
# usage (GPU box, repo root): tools/pmc_rule.sh  -> gpurun_out/pmc_rule.txt : per-wave cycle accounting of pmx_rule_kernel
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcr_$tag -- python3 $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-ppo --no-unidirectional > /tmp/pmcr_$tag.log 2>&1
done
python3 - <<'PY' > $ROOT/gpurun_out/pmc_rule.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcr_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pmx_rule_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print(f"{k:26s} launches {len(v):4d}  mean per launch {sum(v)/len(v):14.1f}")
PY
cat $ROOT/gpurun_out/pmc_rule.txt
