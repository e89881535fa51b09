#!/usr/bin/env python3
"""Shortens a rocprofv3 kernel_stats.csv: prof_summary.py <csv> [pattern] [top N] [full-name width]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else None
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n = 0
for r in rows:
    if pat and not pat.search(r["Name"]):
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = name[:int(sys.argv[4])] if len(sys.argv) > 4 else re.sub(r"\(.*", "", name)[:70]
    print(f'{float(r["TotalDurationNs"]) / 1e6:10.3f} ms {int(r["Calls"]):7d} calls {float(r["AverageNs"]) / 1e3:10.1f} us avg  {name}')
    n += 1
    if n >= top:
        break
