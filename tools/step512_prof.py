#!/usr/bin/env python3
"""200 graph-replayed optimizer steps at the reference's minibatch of 512, for rocprofv3 --kernel-trace --stats (tools/step512_prof.sh)."""
import json, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmx import trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
horizon = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tr = trainer.VecMAPPOTrainer("smallCapture", envs, horizon=horizon, minibatch=512, opponent="random", use_graph=True)
tr.rollout(); tr.compute_gae(); tr.update(max_steps=10)
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.update(max_steps=steps)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
steps = int(tr.stats["optimizer_steps"])          # an update stops after its epochs even when more steps were asked for
print(json.dumps({"steps": steps, "envs": envs, "horizon": horizon, "steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3}))
