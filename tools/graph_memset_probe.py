#!/usr/bin/env python3
"""Does a captured hipMemsetAsync (tensor.zero_()) stay correct over thousands of replays of a hipGraph?

Probe behind DESIGN.md section 5's note on the graph-replayed optimizer step: a graph holding M (zero_ -> add_(1))
pairs on tensors of assorted sizes is replayed R times; after every replay each tensor must equal 1 everywhere.
--mode fill uses fill_(0) (an elementwise kernel) instead of zero_() (a memset node) as the control.
"""
import argparse

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--replays", type=int, default=6000)
    ap.add_argument("--mode", default="zero", choices=["zero", "fill", "zeros_alloc"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sizes = [96, 384, 1 << 12, 1 << 16, 2_634_070, 33, 1 << 20, 7]
    bufs = [torch.ones(sizes[k % len(sizes)], device=dev) for k in range(args.pairs)]
    outs = [None] * args.pairs

    def body():
        for k, b in enumerate(bufs):
            if args.mode == "zero":
                b.zero_()
                b.add_(1.0)
            elif args.mode == "fill":
                b.fill_(0.0)
                b.add_(1.0)
            else:                               # a fresh zero-initialised temporary inside the graph (what reductions do)
                t = torch.zeros_like(b)
                outs[k] = t.add_(1.0)
                b.copy_(outs[k])

    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    bad = None
    for r in range(args.replays):
        g.replay()
        flags = torch.stack([(b != 1.0).sum() for b in bufs]).tolist()
        if any(flags):
            bad = (r, [(k, bufs[k].numel(), int(f), float(bufs[k].max().item()), float(bufs[k].min().item())) for k, f in enumerate(flags) if f])
            break
        for b in bufs:                          # perturb so that a skipped memset shows
            b.add_(1.0)
    print(f"RESULT mode={args.mode} pairs={args.pairs}: " + ("clean" if bad is None else f"first bad replay {bad[0]}: {bad[1][:4]}"), flush=True)


if __name__ == "__main__":
    main()
