#!/usr/bin/env python3
"""MFMA-busy share per kernel from a rocprofv3 --pmc run: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs),
the method of profiles/r01_pmc_mfma_nn_step.md.   python tools/pmc_mfma.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k)[:80]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
rows = []
for k, c in acc.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"], k, cnt[k], c["GRBM_GUI_ACTIVE"], c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
print("| kernel | launches | SQ_VALU_MFMA_BUSY_CYCLES | GRBM_GUI_ACTIVE | mfma share |\n|---|---|---|---|---|")
for b, k, n, gui, share in sorted(rows, reverse=True)[:16]:
    print(f"| `{k}` | {n} | {b:.0f} | {gui:.0f} | {share:.3f} |")
