#!/usr/bin/env python3
"""200 graph-replayed optimizer steps at the reference's minibatch of 512, for rocprofv3 --kernel-trace --stats (tools/step512_prof.sh)."""
import json, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmx import trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tr = trainer.VecMAPPOTrainer("smallCapture", 2048, horizon=8, minibatch=512, opponent="random", use_graph=True)
tr.rollout(); tr.compute_gae(); tr.update(max_steps=10)
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.update(max_steps=steps)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"steps": steps, "steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3}))
