#!/usr/bin/env python3
"""Time the reference's own Python tick loop beside the CPU oracle port -- build container only.

The reference cannot travel to the GPU box, so `bench.py`'s `cpu_baseline` there is the oracle port
(oracle/pmx_oracle.c).  This tool measures, in the container that has /root/reference, both
  * the reference loop: gymPacMan_parallel_env(self_play=True).step with uniform random actions, reset on done
    (gymPacMan.py:143-193; one core, it is single-threaded Python), and
  * the oracle port on the same workload shape on 1 core and on all cores (threads; ctypes releases the GIL),
and writes their ratio to profiles/r01_cpu_reference_ratio.json.  bench.py divides its on-box port figure by that ratio
and reports the result as `cpu_baseline.reference_equiv_derived` -- a derived number, labelled as such.

Optionally (--mappo) it also times the reference's MAPPO policy inference and one optimizer step on the container's
cores, the two figures SURVEY.md section 6 derives the reference's end-to-end rate from.

Nothing of the reference is copied: the script imports it, drives it and records timings.
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/time_reference.py [--ticks 3000] [--mappo]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present: this timing can only be taken in the build container")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

LAYOUTS = {"tinyCapture": f"{REF}/layouts/tinyCapture.lay", "smallCapture": f"{REF}/layouts/smallCapture.lay",
           "bloxCapture": f"{REF}/layouts/bloxCapture.lay"}


def time_reference_loop(path, ticks, length=300):
    with contextlib.redirect_stdout(io.StringIO()):
        import gymPacMan
        env = gymPacMan.gymPacMan_parallel_env(layout_file=path, display=False, length=length, reward_forLegalAction=True,
                                               defenceReward=True, random_layout=False, self_play=True)
        env.reset()
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 5, size=(ticks, 4))
    sink = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink):
        for t in range(ticks):
            _, _, term, _ = env.step({k: int(acts[t, i]) for i, k in enumerate(env.agents)})
            if any(term.values()):
                env.reset()
    return ticks / (time.perf_counter() - t0)


def time_port(text, seconds, threads, length=300, n=256):
    from oracle import oracle as O

    def worker(seed):
        env = O.BatchEnv(text, n, length=length, auto_reset=True)
        rng = np.random.RandomState(seed)
        acts = rng.randint(0, 5, size=(64, n, 4)).astype(np.int8)
        obs = np.zeros((n, 4, 8, env.L.H, env.L.W), np.float32)
        for k in range(4):
            env.tick(acts[k], obs)
        t0 = time.perf_counter()
        ticks = 0
        while time.perf_counter() - t0 < seconds:
            for k in range(16):
                env.tick(acts[(ticks + k) % 64], obs)
            ticks += 16
        return n * ticks / (time.perf_counter() - t0)

    if threads == 1:
        return worker(0)
    with ThreadPoolExecutor(threads) as ex:
        return float(sum(ex.map(worker, range(threads))))


def time_reference_mappo(path, steps=3):
    """Policy inference (batch 1, pacman_mappo_resnet.py:474-480) and one optimizer step at 512 samples (:571-590)."""
    import torch
    with contextlib.redirect_stdout(io.StringIO()):
        import gymPacMan
        import pacman_mappo_resnet as R
        env = gymPacMan.gymPacMan_parallel_env(layout_file=path, display=False, length=300, self_play=True)
        env.reset()
    shape = tuple(env.get_Observation(0).shape)
    torch.manual_seed(0)
    model = R.MAPPOAgent(shape, 5)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, eps=1e-5)
    o1 = torch.rand((1,) + shape)
    with torch.no_grad():
        for _ in range(3):
            model.get_action_and_value(o1, [o1, o1])
        t0 = time.perf_counter()
        for _ in range(50):
            model.get_action_and_value(o1, [o1, o1])
        t_inf = (time.perf_counter() - t0) / 50
    B = 512
    ob, mg = torch.rand((B,) + shape), torch.rand((B,) + shape)
    act = torch.randint(0, 5, (B,))
    adv, ret, old = torch.randn(B), torch.randn(B), -1.6 * torch.ones(B)
    ts = []
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        val, logp, ent = model.evaluate(ob, mg, act)
        ratio = (logp - old).exp()
        pg = -torch.min(ratio * adv, ratio.clamp(0.85, 1.15) * adv).mean()
        loss = pg + 0.5 * 0.5 * (val - ret).pow(2).mean() - 0.01 * ent.mean()
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        ts.append(time.perf_counter() - t0)
    return {"obs_shape": list(shape), "inference_ms_batch1": 1e3 * t_inf, "optimizer_step_s_batch512": float(np.median(ts[1:])),
            "torch_threads": torch.get_num_threads()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=3000)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--mappo", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_cpu_reference_ratio.json"))
    args = ap.parse_args()
    os.chdir(os.environ.get("TMPDIR", "/tmp"))
    from pmx import layout as PL
    cores = os.cpu_count()
    res = {"host": {"cores": cores, "note": "build container, no GPU"}, "layouts": {}}
    for name, path in LAYOUTS.items():
        ref = time_reference_loop(path, args.ticks)
        text = PL.get_layout(name).text
        p1 = time_port(text, args.seconds, 1)
        pn = time_port(text, args.seconds, cores)
        res["layouts"][name] = {
            "reference_env_steps_per_s_1core": ref, "port_env_steps_per_s_1core": p1,
            f"port_env_steps_per_s_{cores}threads": pn, "ratio_port_over_reference_1core": p1 / ref,
            "sample": f"reference: {args.ticks} ticks of one env, uniform random actions, reset on done; "
                      f"port: 256 envs per thread for {args.seconds:.0f} s, float32 planes, auto-reset"}
        print(name, json.dumps(res["layouts"][name]), flush=True)
    if args.mappo:
        res["reference_mappo"] = {k: time_reference_mappo(LAYOUTS[k]) for k in ("smallCapture", "bloxCapture")}
        print(json.dumps(res["reference_mappo"]), flush=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
