#!/usr/bin/env python3
"""Optimizer steps/s with and without CriticEncoderLayer.fold_residual_gradient (the residual branch's gradient added inside
pmx_tok96_backward_res instead of by an add kernel of autograd's): 512-sample graph-replayed steps and 16 384-sample steps,
interleaved rounds in one process.  usage (GPU box): python tools/residual_fold_ab.py"""
import json, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmx import mappo, trainer


def run(fold, minibatch, envs, horizon, steps):
    mappo.CriticEncoderLayer.fold_residual_gradient = fold
    tr = trainer.VecMAPPOTrainer("smallCapture", envs, horizon=horizon, minibatch=minibatch, opponent="random", use_graph=True)
    tr.rollout(); tr.compute_gae(); tr.update(max_steps=10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.update(max_steps=steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tr.env.close()
    return steps / dt


if __name__ == "__main__":
    for minibatch, envs, horizon, steps in ((512, 2048, 16, 180), (16384, 8192, 16, 45)):
        rows = {False: [], True: []}
        for r in range(4):
            for fold in (False, True):
                rows[fold].append(run(fold, minibatch, envs, horizon, steps))
        for fold in (False, True):
            v = sorted(rows[fold])
            print(json.dumps({"minibatch": minibatch, "fold_residual_gradient": fold, "steps_per_s_median": round((v[1] + v[2]) / 2, 1),
                              "min": round(v[0], 1), "max": round(v[-1], 1)}), flush=True)
