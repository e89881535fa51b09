#!/usr/bin/env python3
"""End-to-end MAPPO throughput on one GPU: rollout ticks/s (env + policy inference), optimizer steps/s, updates/s.
    python tools/train_bench.py [--envs 16384] [--horizon 32] [--minibatch 512] [--updates 2] [--obs bfloat16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--layout", default="smallCapture")
ap.add_argument("--envs", type=int, default=16384)
ap.add_argument("--horizon", type=int, default=32)
ap.add_argument("--minibatch", type=int, default=512)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--updates", type=int, default=2)
ap.add_argument("--max-steps", type=int, default=0, help="cap optimizer steps per update (0 = all) for a quick probe")
ap.add_argument("--obs", default=None, help="observation planes (default: the trainer's choice, uint8 under autocast)")
ap.add_argument("--algorithm", default="mappo", choices=["mappo", "ippo"])
ap.add_argument("--opponent", default="random")
ap.add_argument("--no-autocast", action="store_true")
ap.add_argument("--graph", action="store_true", help="replay the optimizer step from a hipGraph (what bench.py does at 512 samples)")
args = ap.parse_args()

import pmx
from pmx import trainer

layout = args.layout
if layout == "mazes":       # one generated 20x20 maze per env (BASELINE config 5)
    layout = [pmx.Layout.from_text(pmx.maze_generator.generate_maze(s)) for s in range(1, args.envs + 1)]
tr = trainer.VecMAPPOTrainer(layout, args.envs, algorithm=args.algorithm, horizon=args.horizon, minibatch=args.minibatch, epochs=args.epochs,
                             obs_dtype=args.obs, opponent=args.opponent, use_autocast=not args.no_autocast, use_graph=args.graph)
sync = lambda: torch.cuda.synchronize()
tr.rollout(); tr.compute_gae(); sync()          # warm-up (MIOpen find, allocator)
res = []
for u in range(args.updates):
    sync(); t0 = time.perf_counter()
    tr.rollout(); sync(); t1 = time.perf_counter()
    tr.compute_gae(); sync(); t2 = time.perf_counter()
    if args.max_steps:
        # time a bounded number of optimizer steps of the real update loop
        S = tr.T * tr.N * 2
        obs = tr.obs_buf.view((S,) + tr.obs_shape); merged = tr.merged_buf.view((tr.T * tr.N,) + tr.obs_shape)
        perm = torch.randperm(S, device=tr.device)
        for k in range(3):
            mb = perm[k * args.minibatch:(k + 1) * args.minibatch]
            tr.learner.update_minibatch(tr._net_in(obs[mb]), tr._net_in(merged[mb // 2]), tr.act_buf.view(S)[mb], tr.logp_buf.view(S)[mb], tr.adv_buf.view(S)[mb], tr.ret_buf.view(S)[mb])
        sync(); t2 = time.perf_counter()
        for k in range(args.max_steps):
            mb = perm[k * args.minibatch:(k + 1) * args.minibatch]
            tr.learner.update_minibatch(tr._net_in(obs[mb]), tr._net_in(merged[mb // 2]), tr.act_buf.view(S)[mb], tr.logp_buf.view(S)[mb], tr.adv_buf.view(S)[mb], tr.ret_buf.view(S)[mb])
        steps = args.max_steps
    else:
        tr.update(); steps = tr.stats["optimizer_steps"]
    sync(); t3 = time.perf_counter()
    res.append(dict(rollout_s=t1 - t0, gae_s=t2 - t1 if not args.max_steps else None, update_s=t3 - t2, steps=steps,
                    rollout_env_steps_per_s=tr.N * tr.T / (t1 - t0), optimizer_steps_per_s=steps / (t3 - t2),
                    samples_per_s=steps * args.minibatch / (t3 - t2)))
print(json.dumps(dict(config=vars(args), runs=res)))
