#!/bin/bash
# run-to-run differences of a replayed update (gather against indexing), whole update vs four steps: how often do the test's bounds miss?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python - <<'PY' 2>&1 | tail -12
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pmx import trainer
def run(gather, max_steps):
    tr = trainer.VecMAPPOTrainer("tinyCapture", 64, horizon=8, minibatch=128, opponent="random", use_graph=True, seed=11)
    tr.graph_gather = gather
    tr.rollout(); tr.compute_gae(); tr.update(max_steps=max_steps)
    out = (tr.learner.bucket.data.clone(), tr.learner.ema.clone(), {k: float(v) for k, v in tr.stats.items() if k in ("pg", "vl", "entropy", "loss", "grad_norm")})
    tr.env.close()
    return out
for max_steps in (None, 4):
    fails, worst_w, worst_s = 0, 0.0, 0.0
    for rep in range(20):
        a, b = run(True, max_steps), run(False, max_steps)
        w = max(float((x - y).norm() / y.norm()) for x, y in ((a[0], b[0]), (a[1], b[1])))
        s = max(abs(a[2][k] - b[2][k]) / (abs(b[2][k]) + 1e-2) for k in a[2])
        fails += (w > 2e-3 or s > 2e-2); worst_w, worst_s = max(worst_w, w), max(worst_s, s)
    print(f"max_steps {max_steps}: {fails} of 20 pairs miss the bounds; worst weights {worst_w:.1e}, worst report {worst_s:.1e}", flush=True)
PY
