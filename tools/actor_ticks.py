#!/usr/bin/env python3
"""Development aid: per-phase s_memtime cycles of one wave of the fused actor tower's forward kernel.  Needs a library whose
pmx_actor.hip was compiled with -DPMX_ACTOR_TIMING (tools/actor_ticks_build.sh); the shipped build has no such symbol."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmx
from pmx import _lib, actor_tower, mappo

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lay = pmx.get_layout("smallCapture")
H, W = lay.height, lay.width
dev = torch.device("cuda")
m = mappo.MAPPOAgent((8, H, W)).to(dev)
params = actor_tower._tower_params(m.actor_backbone)
obs = (torch.rand(B, 8, H, W, device=dev) < 0.25).to(torch.bfloat16)
pack = actor_tower.pack_params(params)
buf = (ctypes.c_ulonglong * 16)()
names = ["conv", "stats", "pass2", "sample_head"]
for mode in ("infer", "train"):
    for _ in range(2):
        actor_tower.tower_forward(obs, pack) if mode == "infer" else actor_tower.actor_tower(m.actor_backbone, obs)
    torch.cuda.synchronize()
    assert lib.pmx_actor_ticks_read(buf, 1) == 0
    n = 5
    for _ in range(n):
        actor_tower.tower_forward(obs, pack) if mode == "infer" else actor_tower.actor_tower(m.actor_backbone, obs)
    torch.cuda.synchronize()
    assert lib.pmx_actor_ticks_read(buf, 1) == 0
    off = 0 if mode == "infer" else 4
    per = {names[i]: buf[off + i] / n for i in range(4)}
    per["total"] = sum(per.values())
    print(json.dumps({"mode": mode, "batch": B, "ticks_per_launch_one_wave": per}))
