#!/usr/bin/env python3
"""Micro-benchmark of the MAPPO network step variants on one GPU (fwd+bwd of ppo_loss + optimizer step)."""
import argparse, json, os, sys, time
import torch
import torch.nn as nn
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmx import mappo

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--variants", default="base,cl")
ap.add_argument("--no-infer", action="store_true")
args = ap.parse_args()
H, W = 11, 14
dev = torch.device("cuda")


class UnfoldConv(nn.Module):
    """3x3 same conv as ONE GEMM over an unfolded input: [B*HW, C*9] x [C*9, Cout]."""
    def __init__(self, conv):
        super().__init__()
        self.weight, self.bias = conv.weight, conv.bias
    def forward(self, x):
        B, C, Hh, Ww = x.shape
        cols = F.unfold(x, 3, padding=1)                       # [B, C*9, HW]
        out = torch.matmul(self.weight.view(self.weight.shape[0], -1), cols)   # [B, Cout, HW]
        return (out + self.bias.view(1, -1, 1)).view(B, -1, Hh, Ww)


def swap_convs(m):
    for name, ch in m.named_children():
        if isinstance(ch, nn.Conv2d):
            setattr(m, name, UnfoldConv(ch))
        else:
            swap_convs(ch)


def run(variant):
    torch.manual_seed(0)
    model = mappo.MAPPOAgent((8, H, W)).to(dev)
    if variant == "unfold":
        swap_convs(model)
    if variant == "cl":
        model = model.to(memory_format=torch.channels_last)
    learner = mappo.PPOLearner(model, autocast_dtype=torch.bfloat16)
    B = args.batch
    obs = (torch.rand(B, 8, H, W, device=dev) < 0.2).to(torch.bfloat16)
    merged = (torch.rand(B, 8, H, W, device=dev) < 0.2).to(torch.bfloat16)
    if variant == "cl":
        obs, merged = obs.contiguous(memory_format=torch.channels_last), merged.contiguous(memory_format=torch.channels_last)
    act = torch.randint(0, 5, (B,), device=dev)
    logp = torch.full((B,), -1.6, device=dev); adv = torch.randn(B, device=dev); ret = torch.randn(B, device=dev)
    step = learner.update_minibatch
    if variant in ("flat", "flatgraph"):
        learner.enable_bf16_flat()
    if variant in ("graph", "flatgraph"):
        learner.capture(B, (8, H, W), torch.bfloat16)
        step = learner.update_minibatch_graph
    for _ in range(3):
        step(obs, merged, act, logp, adv, ret)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step(obs, merged, act, logp, adv, ret)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    last = step(obs, merged, act, logp, adv, ret)
    extra = {k: float(v) for k, v in last.items()}
    if args.no_infer:
        return dict(variant=variant, batch=B, train_ms=dt * 1e3, train_samples_per_s=B / dt, **extra)
    # inference
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(2):
            model.act(obs); model.value(merged)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            model.act(obs); model.value(merged)
        torch.cuda.synchronize(); di = (time.perf_counter() - t0) / args.steps
    return dict(variant=variant, batch=B, train_ms=dt * 1e3, train_samples_per_s=B / dt, infer_ms=di * 1e3, infer_samples_per_s=B / di)


for v in args.variants.split(","):
    print(json.dumps(run(v)), flush=True)
