#!/usr/bin/env python3
"""Per-kernel sums of arbitrary rocprofv3 --pmc counters.   python tools/pmc_counters.py <dir> [kernel substring]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k)[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"    {n:32s} {v:.4g}")
