#!/usr/bin/env python3
"""Calibration: what a pure streaming WRITE (and a copy) reaches on this MI355X at the bench's buffer sizes.
The observation expansion is a pure store stream, so this is the practical ceiling next to the 8 TB/s spec."""
import json
import sys

import torch

sizes = [int(s) for s in sys.argv[1:]] or [80740352, 322961408, 1291845632]
out = {}
for nbytes in sizes:
    n = nbytes // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    res = {}
    for name, fn, moved in (("fill", lambda: a.fill_(1.0), nbytes), ("copy", lambda: b.copy_(a), 2 * nbytes)):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[name] = {"us": ms * 1e3, "GBps": moved / ms / 1e6}
    out[str(nbytes)] = res
    del a, b
print(json.dumps(out))
