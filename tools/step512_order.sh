#!/bin/bash
# usage (GPU box, repo root): tools/step512_order.sh  -> the kernels of ONE graph-replayed 512-sample optimizer step in start order
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_512o
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_512o -- python3 $ROOT/tools/step512_prof.py 60 16384 8 > /tmp/prof_512o.log 2>&1
tail -1 /tmp/prof_512o.log
cd $ROOT
python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/prof_512o/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]) for r in rows]
# the last complete step: from the last pmx_gather_rows_kernel but one to the last one
idx = [i for i, n in enumerate(names) if "pmx_gather_rows_kernel" in n]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"one step: {b - a} launches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r, n in zip(rows[a:b], names[a:b]):
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r.get('Queue_Id','?'):>3} {n[:150]}")
PY
