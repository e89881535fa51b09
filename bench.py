#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Capture-the-Flag tick on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload small16384|tiny4096|blox4096] [--obs float32|bfloat16|uint8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: one gymPacMan tick (4 agent sub-steps, rewards, termination,
4 observation tensors, auto-reset) of every env on the rank's GPU.  Inputs (the action stream) are resident in HBM
before the timed region.  Envs are independent, so N GPUs hold N disjoint shards ("weak" scaling, no data-path
collective); the timed region is bracketed by barrier + synchronize and the reported time is the max over ranks.

The JSON line also carries
  roofline      the observation-expansion kernel (>95 % of the bytes): algorithmic bytes per launch / its average
                duration from start/stop HIP events attached to each dispatch on the launch stream (pmx_profile_begin/end) vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference's tick, oracle/pmx_oracle.c) on one host core, bounded sample
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")    # before HIP initialises: see pacman-marl-2025_amd/__init__.py

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "small16384": ("smallCapture", 16384),   # BASELINE.json configs[2]: the layout the metric is quoted on
    "tiny4096": ("tinyCapture", 4096),       # configs[1]
    "blox4096": ("bloxCapture", 4096),       # the layout the reference actually trains on (20x20)
    "mazes4096": ("mazeGenerator", 4096),    # configs[4] in miniature: every env its own generated 20x20 maze
    "mazes8192": ("mazeGenerator", 8192),    # configs[4] per GPU: 65 536 envs on 8 GPUs, seeds 1..65536 (rank r: r*8192+1 ..)
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)
ELEM = {"float32": 4, "bfloat16": 2, "uint8": 1}


def algorithmic_bytes(H, W, e, n_obs=4):
    """B = 2*S + 41 + n_obs*8*H*W*e, S = 4*H + 48 (packed state, read and written once)."""
    S = 4 * H + 8 + 32 + 8
    return 2 * S + 41 + n_obs * 8 * H * W * e


def reference_ratio(layname):
    """port / reference speed ratio measured in the build container by tools/time_reference.py (the reference cannot
    travel to the GPU box); None when the layout was not timed."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_cpu_reference_ratio.json")) as f:
            return json.load(f)["layouts"][layname]["ratio_port_over_reference_1core"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(layout_rows, length, seconds=12.0, layname=None):
    """Times the oracle port on ONE host core on a bounded sample of the same workload (256 envs, f32 planes)."""
    from oracle import oracle as O
    n = 256
    env = O.BatchEnv(layout_rows, n, length=length, auto_reset=True)
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 5, size=(64, n, 4)).astype(np.int8)
    obs = np.zeros((n, 4, 8, env.L.H, env.L.W), np.float32)
    for k in range(5):
        env.tick(acts[k], obs)
    t0 = time.perf_counter()
    ticks = 0
    while True:
        for k in range(16):
            env.tick(acts[(ticks + k) % 64], obs)
        ticks += 16
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    # the same port on several host cores at once (one BatchEnv per thread; the C call releases the GIL): a fairer CPU figure
    # than one scalar core.  The pool is sized to the GPU box's CPU share for one GPU (16), not to its core count.
    import concurrent.futures as cf
    n_thr = max(1, min(16, os.cpu_count() or 1))

    def worker(seed):
        e2 = O.BatchEnv(layout_rows, n, length=length, auto_reset=True)
        a2 = np.random.RandomState(seed).randint(0, 5, size=(64, n, 4)).astype(np.int8)
        o2 = np.zeros((n, 4, 8, e2.L.H, e2.L.W), np.float32)
        t1, k = time.perf_counter(), 0
        while time.perf_counter() - t1 < 5.0:
            for j in range(16):
                e2.tick(a2[(k + j) % 64], o2)
            k += 16
        return n * k / (time.perf_counter() - t1)
    with cf.ThreadPoolExecutor(n_thr) as ex:
        mt = float(sum(ex.map(worker, range(n_thr))))
    ratio = reference_ratio(layname)
    derived = {} if ratio is None else {
        "reference_equiv_derived": n * ticks / dt / ratio,
        "derived_note": f"port figure / {ratio:.0f} = the port-to-reference speed ratio on one core of the build container "
                        "(tools/time_reference.py, profiles/r01_cpu_reference_ratio.json); derived, not measured here"}
    return {"value": n * ticks / dt, "unit": "env-steps/s", "cores": 1, "kind": "port", **derived,
            "multi_thread": {"value": mt, "unit": "env-steps/s", "cores": n_thr, "sample": f"{n} envs per thread, 5 s"},
            "sample": f"{n} envs x {ticks} ticks of the same layout, float32 planes, auto-reset, uniform random actions "
                      f"({dt:.1f} s on 1 of {os.cpu_count()} host cores)"}


def e2e_probe(layname, dev, rank, world, dist, n_envs, horizon, minibatch, use_graph):
    """One full MAPPO update -- rollout of `horizon` ticks of `n_envs` envs with policy inference, GAE, UPDATE_EPOCHS epochs of
    `minibatch`-sample optimizer steps -- timed end to end after a short warm-up (pacman_mappo_resnet.py:461-600).  With more
    than one rank every optimizer step all-reduces the flat fp32 gradient bucket over RCCL; the times are
    bracketed by barriers, the maximum over ranks is reported and the rates are whole-job."""
    from pmx import trainer
    force_pg = dist is not None and (world > 1 or os.environ.get("PMX_BENCH_FORCE_DP") == "1")
    tr = trainer.VecMAPPOTrainer(layname, n_envs, horizon=horizon, minibatch=minibatch, device=dev, opponent="random", rank=rank,
                                 world_size=world, process_group=dist.group.WORLD if force_pg else None,
                                 use_graph=use_graph and not force_pg)
    if force_pg and world == 1:
        tr.learner.world_size = 2          # rehearsal on a one-GPU box: issue the RCCL all-reduce although there is one rank
        tr.learner.pg = dist.group.WORLD

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def tmax(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    tr.rollout(); tr.compute_gae(); tr.update(max_steps=8)      # warm-up: allocator, library solver search, graph capture
    tr.update_idx = 0
    fence()
    t0 = time.perf_counter()
    tr.rollout()
    fence()
    t1 = time.perf_counter()
    tr.compute_gae()
    tr.update()
    fence()
    t2 = time.perf_counter()
    t_roll, t_upd = tmax(t1 - t0), tmax(t2 - t1)
    steps = int(tr.stats["optimizer_steps"])
    finite = bool(torch.isfinite(tr.stats["grad_norm"]).item())
    grad_bytes = tr.learner.bucket.grad.numel() * 4
    fused = bool(tr.model._use_fused_tower(tr.obs_buf[0, :1, 0]))
    tr.env.close()
    del tr
    torch.cuda.empty_cache()
    env_steps = n_envs * horizon * world
    return {"end_to_end_env_steps_per_s": env_steps / (t_roll + t_upd), "minibatch_per_gpu": minibatch,
            "envs_per_gpu": n_envs, "horizon": horizon, "epochs": 3, "optimizer_steps": steps, "rollout_s": t_roll,
            "gae_plus_update_s": t_upd, "rollout_env_steps_per_s": env_steps / t_roll, "optimizer_steps_per_s": steps / t_upd,
            "train_samples_per_s": steps * minibatch * world / t_upd, "hipgraph_replay": bool(use_graph and not force_pg),
            "finite": finite, "fused_actor_tower": fused,
            "grad_allreduce": (f"one RCCL all-reduce of the flat fp32 gradient bucket ({grad_bytes / 1e6:.1f} MB) per optimizer step"
                               if (world > 1 or force_pg) else "none (1 GPU)")}


def ppo_probe(layname, dev, rank=0, world=1, dist=None, n_envs=16384, horizon=32, large_minibatch=16384):
    """The second half of BASELINE.json's metric: end-to-end MAPPO on BASELINE config 3 (smallCapture, 16 384 envs per GPU,
    T = 32, 3 epochs), once at the reference's minibatch of 512 samples per GPU (pacman_mappo_resnet.py:18; replayed from a
    hipGraph on one GPU, where the step is launch-bound) and once at a large minibatch (fewer, larger optimizer steps -- a
    different optimisation schedule than the reference's, stated as such)."""
    runs = []
    for mb, graph in ((512, world == 1), (large_minibatch, False)):
        try:
            runs.append(e2e_probe(layname, dev, rank, world, dist, n_envs, horizon, mb, graph))
        except Exception as e:                      # the probe must not lose the bench line
            runs.append({"minibatch_per_gpu": mb, "error": f"{type(e).__name__}: {e}"})
    ref, big = runs
    return {"config": f"{layname}, {n_envs} envs/GPU, horizon {horizon}, 3 epochs, paired minibatches (the centralised critic runs once "
                      "per env-tick pair), bf16 autocast, byte observation planes, in-kernel randomTeam opponent",
            "end_to_end": runs,
            "end_to_end_env_steps_per_s": big.get("end_to_end_env_steps_per_s"), "end_to_end_minibatch": large_minibatch,
            "end_to_end_env_steps_per_s_mb512": ref.get("end_to_end_env_steps_per_s"),
            "optimizer_steps_per_s": ref.get("optimizer_steps_per_s"), "samples_per_gpu_per_step": 512,
            "rollout_env_steps_per_s": big.get("rollout_env_steps_per_s"),
            "network": "MAPPOAgent: actor tower = one fused HIP forward kernel + two backward kernels on bf16 MFMA "
                       "(csrc/pmx_actor.hip); critic batch-major / channels-last: MFMA attention, fused in-projection, out-projection + LayerNorm and "
                       "feed-forward + LayerNorm kernels (csrc/pmx_critic.hip, pmx_train.hip); heads and the projector convolution on hipBLASLt / MIOpen",
            "reference_cpu": "0.49 s per optimizer step (2 steps/s) and about 80 env-steps/s end to end on 8 host cores, smallCapture "
                             "(tools/time_reference.py, profiles/r01_cpu_reference_ratio.json)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="small16384", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--obs", default="float32", choices=sorted(ELEM))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unidirectional", action="store_true", help="skip the pass with a consumer between ticks (for profiler runs)")
    ap.add_argument("--no-ppo", action="store_true", help="skip the short MAPPO rollout/update probe")
    ap.add_argument("--fixed-sweep", action="store_true",
                    help="run EVERY pass with the sweep direction fixed (all bytes to HBM): for profiler runs, so that the trace's average "
                         "duration of the expansion kernel is the one `roofline` is quoted on")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run the collective path is used even at N = 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import pmx
    layname, n_envs = WORKLOADS[args.workload]
    if args.envs:
        n_envs = args.envs
    if layname == "mazeGenerator":
        from pmx import maze_generator
        # one distinct maze per env, the reference generator's seeds 1..N*world; about 1 ms each on the host
        lay = [pmx.Layout.from_text(maze_generator.generate_maze(seed)) for seed in range(rank * n_envs + 1, (rank + 1) * n_envs + 1)]
        H, W = lay[0].height, lay[0].width
    else:
        lay = pmx.get_layout(layname)
        H, W = lay.height, lay.width
    length = 300
    dev = torch.device("cuda", local)
    env = pmx.PmxVecEnv(lay, n_envs, length=length, auto_reset=True, obs_dtype=args.obs, device=dev)
    env.reset()
    if args.fixed_sweep:
        env.set_tuning("expand_alt", 0)
    # the action stream: uniform over the 5 actions from the device Philox generator, resident before timing
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = 64
    actions = torch.randint(0, 5, (n_act, n_envs, 4), generator=g, device=dev, dtype=torch.int8)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        env.step(actions[k % n_act])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        env.step(actions[k % n_act])
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    checksum = int(env.obs.sum(dtype=torch.float64).item()) if args.obs != "uint8" else int(env.obs.sum().item())

    # second pass, same K steps, with start/stop HIP events attached to each kernel dispatch: per-kernel durations for the roofline
    env.profile_begin(args.steps + 8)
    for k in range(args.steps):
        env.step(actions[k % n_act])
    prof = env.profile_end()
    # third pass, the one the roofline is quoted on: the alternating sweep switched off (pmx_set_tuning "expand_alt" 0), so that
    # every byte of the planes has to reach HBM.  The default alternates the sweep direction, and a loop like this one -- the
    # same buffer re-stepped with nothing in between -- then overwrites the tail of tick t while it is still in the 256 MiB
    # Infinity Cache; that figure is reported separately as `cache_assisted`, it is not an HBM rate.
    # fourth pass: the default sweep again, but with a CONSUMER between ticks that reads the planes and writes a rollout slot
    # (what VecMAPPOTrainer.rollout does first with every tick's observations): the cache holds the consumer's traffic then.
    k_uni = min(args.steps, 500)
    env.set_tuning("expand_alt", 0)
    t_uni0 = None
    for k in range(20):
        env.step(actions[k % n_act])
    torch.cuda.synchronize(dev)
    t_uni0 = time.perf_counter()
    for k in range(k_uni):
        env.step(actions[k % n_act])
    torch.cuda.synchronize(dev)
    tick_uni_s = (time.perf_counter() - t_uni0) / k_uni
    env.profile_begin(k_uni + 8)
    for k in range(k_uni):
        env.step(actions[k % n_act])
    prof_uni = env.profile_end()
    env.set_tuning("expand_alt", 0 if args.fixed_sweep else -1)
    prof_cons = None
    if not args.no_unidirectional:
        slots = torch.empty((4,) + tuple(env.obs.shape), dtype=env.obs.dtype, device=dev)
        k_c = min(args.steps, 200)
        for k in range(10):
            env.step(actions[k % n_act]); slots[k % 4].copy_(env.obs)
        env.profile_begin(k_c + 8)
        for k in range(k_c):
            env.step(actions[k % n_act]); slots[k % 4].copy_(env.obs)
        prof_cons = env.profile_end()
        del slots
    e = ELEM[args.obs]
    expand_bytes = n_envs * 4 * 8 * H * W * e                      # algorithmic bytes of one expansion launch
    expand_alt_s = prof["expand_ms"] / 1e3 / max(prof["expand_launches"], 1)
    expand_s = prof_uni["expand_ms"] / 1e3 / max(prof_uni["expand_launches"], 1)
    rule_s = prof_uni["rule_ms"] / 1e3 / max(prof_uni["rule_launches"], 1)
    achieved = expand_bytes / expand_s / 1e9
    # same-process, same-device calibration: a pure streaming write of the same number of bytes (torch fill_)
    cal = torch.empty(expand_bytes // 4, dtype=torch.float32, device=dev)
    for _ in range(5):
        cal.fill_(1.0)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(20):
        cal.fill_(1.0)
    c1.record()
    torch.cuda.synchronize(dev)
    write_ceiling = expand_bytes / (c0.elapsed_time(c1) / 20 / 1e3) / 1e9
    del cal
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.workload}:{args.obs}"
            if n_envs == WORKLOADS[args.workload][1]:           # the PMC passes were taken at the workload's own env count
                traffic = tj.get(key, {}).get("expand_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        B = algorithmic_bytes(H, W, e)
        value = world * n_envs * args.steps / dt
        line = {
            "metric": "env-steps/s (4-agent smallCapture)" if layname == "smallCapture" else f"env-steps/s (4-agent {layname})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{layname}.lay 2v2, {n_envs} envs/GPU, gymPacMan tick (self_play), length {length}, "
                                   f"auto-reset, uniform random actions (Philox seed 1234), 4 observations/env-tick",
                       "envs_per_gpu": n_envs, "layout": layname, "obs_dtype": args.obs, "obs_bytes_per_elem": e,
                       "parallelism": f"env-sharded x{world}, no data-path collective"},
            "agent_steps_per_s": 4 * value,
            "algorithmic_bytes_per_env_step": B,
            "tick_GBps": value / world * B / 1e9,
            "tick_all_bytes_to_hbm": {"us_per_tick": tick_uni_s * 1e6, "GBps": n_envs * B / tick_uni_s / 1e9,
                                      "frac_of_peak": n_envs * B / tick_uni_s / 1e9 / HBM_PEAK_GBS,
                                      "note": "whole tick (rule + expansion kernels, host launch path) with the sweep direction fixed"},
            "roofline": {"bound": "hbm", "kernel": "pmx_expand_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_with": "sweep direction fixed (pmx_set_tuning expand_alt 0): every byte of the planes reaches HBM",
                         "traffic_scope": "PMC TCC FETCH_SIZE/WRITE_SIZE = L2 <-> fabric requests; the Infinity Cache sits behind them",
                         "algorithmic_bytes_per_launch": expand_bytes, "avg_launch_us": expand_s * 1e6,
                         "launches": prof_uni["expand_launches"], "rule_kernel_avg_us": rule_s * 1e6,
                         "same_device_fill_GBps": write_ceiling,
                         "cache_assisted": {
                             "avg_launch_us": expand_alt_s * 1e6, "GBps": expand_bytes / expand_alt_s / 1e9,
                             "note": "the default path in THIS loop (same buffer re-stepped, nothing in between): the sweep direction "
                                     "alternates and tick t+1 overwrites the tail of tick t while it is still in the 256 MiB Infinity "
                                     "Cache; algorithmic bytes / time, NOT an HBM rate (`value` and `ms_per_step` are from this path)"},
                         "with_consumer": None if prof_cons is None else {
                             "avg_launch_us": prof_cons["expand_ms"] * 1e3 / max(prof_cons["expand_launches"], 1),
                             "GBps": expand_bytes / (prof_cons["expand_ms"] / 1e3 / max(prof_cons["expand_launches"], 1)) / 1e9,
                             "note": "default path with a copy of the planes into a rollout slot between ticks (reads them, writes as "
                                     "many bytes elsewhere)"}},
            "obs_checksum": checksum,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(lay[0].text if isinstance(lay, list) else lay.text, length, layname=layname)
    env.close()
    ppo = None
    if not args.no_ppo and layname != "mazeGenerator":
        ppo = ppo_probe(layname, dev, rank, world, dist)
    if rank == 0:
        if ppo is not None:
            line["ppo"] = ppo
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
