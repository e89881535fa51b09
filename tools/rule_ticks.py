#!/usr/bin/env python3
"""Development aid: s_memtime cycles per phase of one wave of pmx_rule_kernel (library built with -DPMX_RULE_TIMING:
tools/ab_build.sh rt pmx_step.hip -DPMX_RULE_TIMING; cp ab/rt.so pacman-marl-2025_amd/libpmx_hip.so)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmx
from pmx import _lib

lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
DT = sys.argv[2] if len(sys.argv) > 2 else "float32"
env = pmx.PmxVecEnv("smallCapture", n_envs=N, length=300, auto_reset=True, obs_dtype=DT, device="cuda:0", seed=1)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.randint(0, 5, (N, 4), device="cuda", generator=g, dtype=torch.int8) for _ in range(16)]
buf = (ctypes.c_ulonglong * 16)()
for i in range(50):
    env.step(acts[i % 16])
torch.cuda.synchronize()
assert lib.pmx_rule_ticks_read(buf, 1) == 0
n = 400
for i in range(n):
    env.step(acts[i % 16])
torch.cuda.synchronize()
assert lib.pmx_rule_ticks_read(buf, 1) == 0
names = ["issue_loads+ctx", "commit(wait loads)", "substep0+snap", "substep1+snap", "substep2+snap", "substep3", "finish", "store_env"]
launches = buf[15]
per = {names[i]: round(buf[i] / launches, 1) for i in range(8)}
per["total_ticks"] = round(sum(buf[i] for i in range(8)) / launches, 1)
out = {"envs": N, "obs": DT, "launches": int(launches), "s_memtime_ticks_per_launch_block0": per}
if buf[14]:
    out["expand4_wave_ticks (3 sampled blocks)"] = {n: round(buf[8 + i] / buf[14], 1) for i, n in enumerate(["loads+table_init", "build", "stream"])}
print(json.dumps(out))
