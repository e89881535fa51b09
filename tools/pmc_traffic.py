#!/usr/bin/env python3
"""Turn rocprofv3 --pmc CSV output into per-launch HBM traffic of the expansion kernel.

MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB and need separate passes (TCC slots);
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced read stream, WRITE_SIZE is exact for
16-byte-per-lane streaming stores.  So   hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   per launch.

    python tools/pmc_traffic.py <dir with FETCH pass> <dir with WRITE pass> <workload key> [out.json] [kernel substring]
"""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel="pmx_expand"):     # pmx_expand_kernel and pmx_expand4_kernel
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and kernel in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    return vals


kern = sys.argv[5] if len(sys.argv) > 5 else "pmx_expand"
fetch = per_launch(sys.argv[1], "FETCH_SIZE", kern)
write = per_launch(sys.argv[2], "WRITE_SIZE", kern)
key = sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
med = lambda v: sorted(v)[len(v) // 2] if v else None
f, w = med(fetch), med(write)
entry = {"fetch_size_kib_raw": f, "write_size_kib": w, "launches_seen": [len(fetch), len(write)],
         "correction": "hbm = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
         "expand_hbm_bytes_per_launch": (2 * f * 1024 + w * 1024) if f is not None and w is not None else None}
data = json.load(open(out)) if os.path.exists(out) else {}
data[key] = entry
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(entry))
