#!/bin/bash
# usage (GPU box, repo root): tools/step512_prof.sh [top]  -> per-kernel totals of 200 graph-replayed 512-sample optimizer steps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_512
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_512 -- python3 $ROOT/tools/step512_prof.py 190 16384 8 > /tmp/prof_512.log 2>&1
tail -1 /tmp/prof_512.log
cd $ROOT
python tools/prof_summary.py $(find /tmp/prof_512 -name "*kernel_stats.csv" | head -1) "" ${1:-60}
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_512/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"all kernels: {tot/1e6:.1f} ms in {calls} launches")
PY
