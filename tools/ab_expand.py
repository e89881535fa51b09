#!/usr/bin/env python3
"""In-process A/B of expansion-kernel launch options (cdna_hip_programming.md rule 24: interleaved rounds, one process).
    python tools/ab_expand.py --envs 16384 --obs float32 --var expand_nt --values 0,1      (keys of pmx_set_tuning)"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pmx

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=16384)
ap.add_argument("--obs", default="float32")
ap.add_argument("--layout", default="smallCapture")
ap.add_argument("--var", default="expand_nt")
ap.add_argument("--values", default="0,1")
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--ticks", type=int, default=100)
a = ap.parse_args()
env = pmx.PmxVecEnv(a.layout, a.envs, length=300, auto_reset=True, obs_dtype=a.obs)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.randint(0, 5, (32, a.envs, 4), generator=g, device="cuda", dtype=torch.int8)
vals = a.values.split(",")
res = {v: [] for v in vals}
tick = {v: [] for v in vals}
for r in range(a.rounds):
    for v in vals:
        env.set_tuning(a.var, int(v))
        for k in range(10):
            env.step(acts[k % 32])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(a.ticks):
            env.step(acts[k % 32])
        e1.record(); torch.cuda.synchronize()
        tick[v].append(e0.elapsed_time(e1) / a.ticks * 1e3)
        env.profile_begin(a.ticks + 4)
        for k in range(a.ticks):
            env.step(acts[k % 32])
        p = env.profile_end()
        res[v].append(p["expand_ms"] / p["expand_launches"] * 1e3)
for v in vals:
    print(f"{a.var}={v}: expand median {statistics.median(res[v]):.1f} us (min {min(res[v]):.1f}), tick median {statistics.median(tick[v]):.1f} us")
