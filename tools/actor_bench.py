#!/usr/bin/env python3
"""Times the fused actor tower (csrc/pmx_actor.hip) against the library path (MIOpen convolutions + GroupNorm/GELU kernels):
inference, training forward, forward + backward.   python tools/actor_bench.py [--batch 8192] [--layout smallCapture]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pmx
from pmx import mappo, actor_tower

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[512, 8192, 32768])
ap.add_argument("--layout", default="smallCapture")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-library", action="store_true", help="time the fused kernels only")
args = ap.parse_args()
lay = pmx.get_layout(args.layout)
H, W = lay.height, lay.width
dev = torch.device("cuda")
torch.manual_seed(0)
m = mappo.MAPPOAgent((8, H, W)).to(dev)
params = actor_tower._tower_params(m.actor_backbone)
FLOP = 2 * H * W * 9 * (8 * 16 + 16 * 32 + 6 * 32 * 32)      # algorithmic forward FLOPs per sample (real channel counts)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for B in args.batch:
    obs = (torch.rand(B, 8, H, W, device=dev) < 0.25).to(torch.bfloat16)
    pack = actor_tower.pack_params(params)
    res = {"batch": B, "layout": args.layout}
    res["fused_infer_ms"] = timeit(lambda: actor_tower.tower_forward(obs, pack), args.iters) * 1e3

    def fused_train():
        feat = actor_tower.actor_tower(m.actor_backbone, obs)
        torch.autograd.grad(feat.float().sum(), params)

    def fused_fwd_save():
        actor_tower.actor_tower(m.actor_backbone, obs)
    res["fused_train_fwd_ms"] = timeit(fused_fwd_save, args.iters) * 1e3
    res["fused_fwd_bwd_ms"] = timeit(fused_train, args.iters) * 1e3

    def lib_infer():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            m.actor_backbone(obs.contiguous(memory_format=torch.channels_last))

    def lib_train():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m.actor_backbone(obs.contiguous(memory_format=torch.channels_last))
        torch.autograd.grad(out.float().sum(), params)
    if not args.no_library:
        res["library_infer_ms"] = timeit(lib_infer, args.iters) * 1e3
        res["library_fwd_bwd_ms"] = timeit(lib_train, args.iters) * 1e3
    res["fused_infer_TFLOPs"] = B * FLOP / (res["fused_infer_ms"] * 1e-3) / 1e12
    res["fused_fwd_bwd_TFLOPs"] = 3 * B * FLOP / (res["fused_fwd_bwd_ms"] * 1e-3) / 1e12
    print(json.dumps(res), flush=True)
