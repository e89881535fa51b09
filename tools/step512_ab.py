#!/usr/bin/env python3
"""Optimizer steps/s at the reference's minibatch of 512 (graph replay) with kernel families toggled: where does the step go?"""
import json, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmx import mappo, trainer

def run(tag, fused_tower=True, fused_ffn=True, graph=True, steps=150):
    mappo.MAPPOAgent.fused_tower, mappo.MAPPOAgent.fused_ffn = fused_tower, fused_ffn
    tr = trainer.VecMAPPOTrainer("smallCapture", 2048, horizon=8, minibatch=512, opponent="random", use_graph=graph)
    tr.rollout(); tr.compute_gae(); tr.update(max_steps=10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.update(max_steps=steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tr.env.close()
    print(json.dumps({"variant": tag, "graph": graph, "steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3}), flush=True)

run("all fused")
run("no fused ffn/tok", fused_ffn=False)
run("no fused tower", fused_tower=False)
run("none", fused_tower=False, fused_ffn=False)
run("all fused, eager", graph=False)
os.environ["PMX_ATTN_BWD_GENERIC"] = "1"
