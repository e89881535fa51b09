#!/usr/bin/env python3
"""Localise the non-finite gradient seen when the bf16 optimizer step is replayed from a hipGraph.

Update-only loop (no policy sampling, so a NaN cannot reach torch.multinomial's device assert): capture the step for a
512-sample minibatch, replay it --replays times on fresh random inputs and stop at the first non-finite gradient norm,
printing which parameters hold non-finite gradients.  --disable {attn,gn,ln,tl} swaps one family of hand-written kernels
for the stock torch ops to bisect.

    python tools/graph_debug.py --replays 3000 [--disable attn] [--eager]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=3000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--disable", default="", help="comma list of attn,gn,ln,tl")
    ap.add_argument("--eager", action="store_true", help="run the same loop eagerly (control)")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--verbose-from", type=int, default=1 << 30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--step-offset", type=int, default=0, help="start the Adam step counter here")
    ap.add_argument("--frozen-scalars", action="store_true", help="replay the graph directly: step scalars stay at step 1")
    ap.add_argument("--gc", default="auto", choices=["auto", "off", "every"], help="Python cyclic GC: default, disabled, or collect before every replay")
    ap.add_argument("--keep", action="store_true", help="keep the token_linear backward tensors of the captured run for inspection")
    args = ap.parse_args()
    from pmx import mappo
    mappo.PPOLearner.fused_optimizer = False     # this tool inspects the per-tensor norms of the torch path
    off = set(filter(None, args.disable.split(",")))
    if "attn" in off:
        def sdpa(qkv):
            S, B, _ = qkv.shape
            q, k, v = qkv.chunk(3, dim=-1)
            q, k, v = (t.reshape(S, B * 4, 8).transpose(0, 1).reshape(B, 4, S, 8) for t in (q, k, v))
            return F.scaled_dot_product_attention(q, k, v).permute(2, 0, 1, 3).reshape(S, B, 32)
        mappo.attention8 = sdpa
    if "gn" in off:
        mappo.group_norm_gelu = lambda h, res, gn: F.gelu(gn(h) if res is None else gn(h) + res)
    if "ln" in off:
        mappo.add_layer_norm_small = lambda x, a, ln: F.layer_norm(x + a, (32,), ln.weight, ln.bias, ln.eps)
    if "tl" in off:
        mappo.token_linear = F.linear
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shape = (8, 11, 14)
    model = mappo.MAPPOAgent(shape, 5, 2).to(dev)
    ac = None if args.fp32 else torch.bfloat16
    L = mappo.PPOLearner(model, autocast_dtype=ac)
    in_dt = torch.float32 if args.fp32 else torch.bfloat16
    B = args.batch
    if not args.eager:
        if args.keep:
            import types
            orig = L.capture
            # keep only the tensors of the captured run: switch the hook on around the capture itself
            class _Lst(list):
                pass
            keep = _Lst()
            real_graph = torch.cuda.graph

            class hooked(real_graph):
                def __enter__(self_):
                    mappo._DEBUG_KEEP = keep
                    return super().__enter__()

                def __exit__(self_, *a):
                    r = super().__exit__(*a)
                    mappo._DEBUG_KEEP = None
                    return r
            torch.cuda.graph = hooked
            L.capture(B, shape, in_dt)
            torch.cuda.graph = real_graph
            print("kept", len(keep), "token_linear backward records", flush=True)
        else:
            L.capture(B, shape, in_dt)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    g = torch.Generator(device=dev).manual_seed(args.seed)
    bad_at = -1
    import gc
    if args.gc == "off":
        gc.disable()
    for it in range(args.replays):
        obs = (torch.rand((B,) + shape, device=dev, generator=g) < 0.2).to(in_dt)
        mg = (torch.rand((B,) + shape, device=dev, generator=g) < 0.2).to(in_dt)
        act = torch.randint(0, 5, (B,), device=dev, generator=g)
        logp = -1.6 + 0.05 * torch.randn(B, device=dev, generator=g)
        adv = torch.randn(B, device=dev, generator=g)
        ret = torch.randn(B, device=dev, generator=g)
        fn = L.update_minibatch if args.eager else L.update_minibatch_graph
        if it == 0 and args.step_offset:
            L.step_count = args.step_offset
        if args.frozen_scalars and not args.eager:
            def fn(o, m, a, lp, ad, r):
                i = L._g_in
                i["obs"].copy_(o); i["merged"].copy_(m); i["act"].copy_(a); i["logp"].copy_(lp); i["adv"].copy_(ad); i["ret"].copy_(r)
                L._graph.replay()
                return L._g_stats
        if args.gc == "every":
            n_free = gc.collect()
            if it < 3:
                print(f"gc.collect() before replay {it}: {n_free} objects", flush=True)
        st = fn(obs, mg, act, logp, adv, ret)
        gn = float(st["grad_norm"].item())
        if not (gn == gn and abs(gn) != float("inf")):
            bad_at = it
            print(f"non-finite grad_norm {gn} at replay {it}", flush=True)
            for n, p in zip(names, L.bucket.params):
                gr = p.grad
                nf = int((~torch.isfinite(gr)).sum().item())
                big = float(gr[torch.isfinite(gr)].abs().max().item()) if nf < gr.numel() else float("nan")
                if nf or big > 1e6:
                    print(f"   {n}: {nf}/{gr.numel()} non-finite, max finite |g| = {big:.3e}", flush=True)
            if not args.eager:
                nr = L._g_norms.tolist()
                print("   per-tensor grad norms (pre-clip):", [(n, v) for n, v in zip(names, nr) if not (v == v and abs(v) < 1e6)], flush=True)
                print("   largest finite:", sorted(((v, n) for n, v in zip(names, nr) if v == v and abs(v) != float("inf")), reverse=True)[:3], flush=True)
            if not args.eager and args.keep:
                for k, (gy, gb, gw) in enumerate(keep):
                    print(f"   tl-backward #{k}: gy {tuple(gy.shape)} nonfinite {int((~torch.isfinite(gy)).sum())} max|gy| {float(gy.float().abs().max()):.3e}; "
                          f"gb nonfinite {int((~torch.isfinite(gb)).sum())} of {gb.numel()}  gw nonfinite {int((~torch.isfinite(gw)).sum())}", flush=True)
                    if int((~torch.isfinite(gb)).sum()):
                        print("      gb:", gb.float().tolist()[:100], flush=True)
                        print("      recomputed now:", gy.sum((0, 1)).float().tolist()[:8], flush=True)
            print("   loss stats:", {k: float(v.item()) for k, v in st.items()}, flush=True)
            print("   params finite:", bool(torch.isfinite(L.bucket.data).all().item()), flush=True)
            break
        if it % 500 == 0 or args.verbose_from <= it:
            print(f"replay {it}: grad_norm {gn:.4f} loss {float(st['loss'].item()):.4f}", flush=True)
    if bad_at < 0:          # throughput of the same loop without the per-step host synchronisation
        import time
        data = [(obs, mg, act, logp, adv, ret)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn(*data[0])
        torch.cuda.synchronize()
        print(f"TIMING {'eager' if args.eager else 'graph'} batch {B}: {200 / (time.perf_counter() - t0):.1f} optimizer steps/s", flush=True)
    print(f"RESULT disable={sorted(off)} eager={args.eager} fp32={args.fp32}: " + ("clean" if bad_at < 0 else f"bad at {bad_at}"), flush=True)


if __name__ == "__main__":
    main()
