#!/usr/bin/env python3
"""Ticks/s of the drop-in dict API (one GPU-resident game, host CaptureAgent bots as red) -- BASELINE config 0's shape.
The reference's own loop runs 360-390 ticks/s with baselineTeam and 530-590 with randomTeam on one host core."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--layout", default="tinyCapture")
ap.add_argument("--ticks", type=int, default=900)
args = ap.parse_args()
from pmx import gymPacMan_parallel_env

for team in ("baselineTeam", "randomTeam"):
    random.seed(0)
    env = gymPacMan_parallel_env(layout_file=f"layouts/{args.layout}.lay", length=299, enemieName=team, self_play=False)
    env.reset()
    rng = np.random.RandomState(1)
    for _ in range(30):
        _, _, term, _ = env.step({env.agents[1]: int(rng.randint(5)), env.agents[3]: int(rng.randint(5))})
    t0 = time.perf_counter()
    for t in range(args.ticks):
        _, _, term, _ = env.step({env.agents[1]: int(rng.randint(5)), env.agents[3]: int(rng.randint(5))})
        if any(term.values()):
            env.reset()
    dt = time.perf_counter() - t0
    print(json.dumps({"layout": args.layout, "red": team, "ticks_per_s": args.ticks / dt}), flush=True)
