#!/usr/bin/env python3
"""Times the critic (value head) forward / forward+backward with and without the fused feed-forward kernels, and the FFN
kernels alone.   python tools/critic_bench.py [--batch 8192]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pmx
from pmx import mappo

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--layout", default="smallCapture")
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
lay = pmx.get_layout(args.layout)
H, W = lay.height, lay.width
dev = torch.device("cuda")
torch.manual_seed(0)
m = mappo.MAPPOAgent((8, H, W)).to(dev)
B = args.batch
merged = (torch.rand(B, 8, H, W, device=dev) < 0.25).to(torch.bfloat16)
cparams = [p for n, p in m.named_parameters() if n.startswith("critic")]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.iters * 1e3


def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        m.value(merged)


def fwd_bwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v = m.value(merged)
    torch.autograd.grad(v.float().sum(), cparams)


res = {"batch": B}
for fused in (True, False):
    mappo.MAPPOAgent.fused_ffn = fused
    k = "fused_ffn" if fused else "separate_ops"
    res[k + "_fwd_ms"] = timeit(fwd)
    res[k + "_fwd_bwd_ms"] = timeit(fwd_bwd)
mappo.MAPPOAgent.fused_ffn = True
layer = m.critic_transformer.layers[0]
x = torch.randn(H * W, B, 32, device=dev).to(torch.bfloat16).requires_grad_(True)
res["ffn_kernel_fwd_ms"] = timeit(lambda: mappo.ffn_layer_norm(x.detach(), layer.linear1, layer.linear2, layer.norm2))


def ffn_fb():
    y = mappo.ffn_layer_norm(x, layer.linear1, layer.linear2, layer.norm2)
    torch.autograd.grad(y.float().sum(), [x, layer.linear1.weight])


res["ffn_kernel_fwd_bwd_ms"] = timeit(ffn_fb)
print(json.dumps(res))
