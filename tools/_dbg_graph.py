import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pmx import trainer
for rep in range(3):
    tr = trainer.VecMAPPOTrainer("smallCapture", n_envs=256, horizon=12, minibatch=512, epochs=2, obs_dtype="bfloat16", seed=5 + rep, length=20, opponent="random", use_graph=True)
    bad = 0
    for u in range(3):
        st = tr.train_update()
        ok = all(bool(torch.isfinite(st[k]).all()) for k in ("pg", "vl", "grad_norm"))
        bad += (not ok)
    print("rep", rep, "bad updates", bad, "params finite", bool(torch.isfinite(tr.learner.bucket.data).all()), flush=True)
    tr.env.close()
