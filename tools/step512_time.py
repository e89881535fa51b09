"""Optimizer steps/s of the hipGraph-replayed 512-sample step (smallCapture).  Feature switches via the environment:
PMX_NO_FUSED_PROJECTOR, PMX_NO_DEFER_SUMS, PMX_NO_PREPACK, PMX_NO_TWO_STREAMS."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmx import trainer
tr = trainer.VecMAPPOTrainer("smallCapture", 4096, horizon=8, minibatch=512, opponent="random", use_graph=True)
tr.rollout(); tr.compute_gae(); tr.update(max_steps=16)
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.update()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = int(tr.stats["optimizer_steps"])
print({k: v for k, v in os.environ.items() if k.startswith("PMX_")}, f"{n / dt:.0f} steps/s ({dt / n * 1e6:.0f} us per step)")
