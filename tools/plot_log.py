#!/usr/bin/env python3
"""Training curves from tools/train.py --log files (the reference draws the same kind of figure, pacman_mappo_resnet.py:653-699).

    python tools/plot_log.py run_a.jsonl [run_b.jsonl ...] --out curves.png
"""
import argparse
import json
import os

import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt

ap = argparse.ArgumentParser()
ap.add_argument("logs", nargs="+")
ap.add_argument("--out", default="curves.png")
args = ap.parse_args()

panels = [("reward", "rollout reward per env"), ("win_rate", "win rate of finished episodes"), ("entropy", "policy entropy"),
          ("vl", "value loss"), ("env_steps_per_s", "env-steps/s (rollout + all PPO epochs)")]
fig, axes = plt.subplots(1, len(panels), figsize=(4.2 * len(panels), 3.4))
for path in args.logs:
    rows = [json.loads(l) for l in open(path) if l.strip().startswith("{")]
    rows = [r for r in rows if "reward" in r]
    label = os.path.splitext(os.path.basename(path))[0]
    for ax, (key, title) in zip(axes, panels):
        ax.plot([r["update"] for r in rows], [r.get(key) for r in rows], label=label)
        ax.set_title(title, fontsize=9)
        ax.set_xlabel("update")
axes[0].legend(fontsize=7)
fig.tight_layout()
fig.savefig(args.out, dpi=110)
print("wrote", args.out)
