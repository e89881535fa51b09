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
  roofline_emit_team  the same for pmx_emit_team_obs on byte planes, the observation kernel the training loop runs
  cpu_baseline  the CPU oracle (a port of the reference's tick, oracle/pmx_oracle.c) on one host core, bounded sample
  ppo, ppo_config5  end-to-end MAPPO (rollout + GAE + 3 PPO epochs) on the workload at 512- and 16 384-sample minibatches, the
                one-rank RCCL rehearsal of the data-parallel step, and BASELINE config 5's per-GPU shard (MAPPO and IPPO)
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
ROOFLINE_MIN_LAUNCHES = 500
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


def emit_team_probe(lay, n_envs, length, dev, rank, actions, launches=ROOFLINE_MIN_LAUNCHES):
    """The observation kernel the TRAINING loop runs (pmx_emit_team_kernel on byte planes: the two learners' canonicalised
    observations + the merged critic input of every env, written into rollout-buffer slots), timed like the expansion kernel:
    start/stop events on each dispatch, >= 500 launches, a tick without observations between two launches as in
    VecMAPPOTrainer.rollout, and a ring of slots larger than the Infinity Cache so that the bytes have to reach HBM."""
    import pmx
    env = pmx.PmxVecEnv(lay, n_envs, length=length, auto_reset=True, obs_dtype="uint8", device=dev, seed=rank)
    env.reset()
    H, W = env.layout.height, env.layout.width
    per = n_envs * 3 * 8 * H * W                                    # algorithmic bytes of one launch (uint8)
    n_slots = max(4, min(64, (320 << 20) // per + 1))
    team = torch.empty((n_slots, n_envs, 2, 8, H, W), dtype=torch.uint8, device=dev)
    merged = torch.empty((n_slots, n_envs, 8, H, W), dtype=torch.uint8, device=dev)
    n_act = actions.shape[0]
    for k in range(20):
        env.step(actions[k % n_act], want_obs=False)
        env.emit_team_obs(False, team[k % n_slots], merged[k % n_slots])
    seg, seg_us, tot_ms, tot_n, done = 125, [], 0.0, 0, 0
    while done < launches:
        n = min(seg, launches - done)
        env.profile_begin(n + 8)
        for k in range(n):
            env.step(actions[(done + k) % n_act], want_obs=False)
            env.emit_team_obs(False, team[(done + k) % n_slots], merged[(done + k) % n_slots])
        pr = env.profile_end()
        seg_us.append(pr["expand_ms"] * 1e3 / max(pr["expand_launches"], 1))
        tot_ms += pr["expand_ms"]; tot_n += pr["expand_launches"]
        done += n
    avg_s = tot_ms / 1e3 / max(tot_n, 1)
    checksum = int(team[(done - 1) % n_slots].sum().item()) + int(merged[(done - 1) % n_slots].sum().item())
    env.close()
    del team, merged
    torch.cuda.empty_cache()
    return {"bound": "hbm", "kernel": "pmx_emit_team_kernel<uint8>", "achieved": per / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": per / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": per,
            "avg_launch_us": avg_s * 1e6, "launches": tot_n, "segment_avg_us_min_max": [min(seg_us), max(seg_us)],
            "slots": n_slots, "checksum": checksum,
            "note": "3 slots (learner, learner, merged) x 8 planes x H x W bytes per env-tick; the product's observation path "
                    "(VecMAPPOTrainer.rollout), one launch per tick, one wavefront per env"}


def e2e_probe(layout, dev, rank, world, dist, n_envs, horizon, minibatch, use_graph, algorithm="mappo", rehearse_dp=False, autocast=True):
    """One full MAPPO update -- rollout of `horizon` ticks of `n_envs` envs with policy inference, GAE, UPDATE_EPOCHS epochs of
    `minibatch`-sample optimizer steps -- timed end to end after a short warm-up (pacman_mappo_resnet.py:461-600).  With more
    than one rank every optimizer step all-reduces the flat fp32 gradient bucket over RCCL, the actor's slice while the
    critic's backward runs; the replayed step is then one hipGraph per gradient group with the eager collectives between them.
    The times are bracketed by barriers, the maximum over ranks is reported and the rates are whole-job.  rehearse_dp: the same
    collective path on a ONE-rank RCCL group (what a one-GPU box can validate: capture, replay, stream order; not the wire time)."""
    from pmx import trainer
    dp = dist is not None and (world > 1 or rehearse_dp)
    tr = trainer.VecMAPPOTrainer(layout, n_envs, horizon=horizon, minibatch=minibatch, device=dev, opponent="random", rank=rank,
                                 world_size=world, process_group=dist.group.WORLD if dp else None, use_graph=use_graph,
                                 algorithm=algorithm, force_collectives=dp and world == 1, use_autocast=autocast)

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
    graphs = 0 if not tr.use_graph else (len(tr.learner._graphs) if tr.learner._graphs else 1)
    L = tr.learner
    eager_groups = L._grad_groups() if dp else []
    groups = [int(L._offsets[hi] - L._offsets[lo]) * 4 for lo, hi in (L._g_groups if (dp and L._graphs) else eager_groups)] if dp else []
    tr.env.close()
    del tr
    torch.cuda.empty_cache()
    env_steps = n_envs * horizon * world
    return {"end_to_end_env_steps_per_s": env_steps / (t_roll + t_upd), "ppo_updates_per_s": 1.0 / (t_roll + t_upd),
            "algorithm": algorithm, "minibatch_per_gpu": minibatch,
            "envs_per_gpu": n_envs, "horizon": horizon, "epochs": 3, "optimizer_steps": steps, "rollout_s": t_roll,
            "gae_plus_update_s": t_upd, "rollout_env_steps_per_s": env_steps / t_roll, "optimizer_steps_per_s": steps / t_upd,
            "train_samples_per_s": steps * minibatch * world / t_upd, "hipgraph_replay": bool(graphs), "hipgraphs_per_step": graphs,
            "finite": finite, "fused_actor_tower": fused,
            "grad_allreduce": (f"RCCL all-reduce (mean) of the flat fp32 gradient bucket ({grad_bytes / 1e6:.1f} MB) per optimizer step in "
                               f"{len(groups)} slice(s) of {[round(g / 1e6, 2) for g in groups]} MB"
                               + (" (actor first, in flight during the critic's backward)" if len(groups) > 1 else
                                  " between the [forward + backward] graph and the [optimizer] graph (actor and critic halves run side by "
                                  "side on two streams inside the first)")
                               + (" -- ONE-rank rehearsal" if world == 1 else "")) if dp else "none (1 GPU)"}


NETWORK_NOTE = ("MAPPOAgent: actor tower = one fused HIP forward kernel + two backward kernels on bf16 MFMA (csrc/pmx_actor.hip; boards of 10, "
                "11 and 28 position tiles = tiny, small and the 20x20 boards); critic batch-major / channels-last: MFMA attention, fused "
                "in-projection, out-projection + LayerNorm and feed-forward + LayerNorm kernels (csrc/pmx_critic.hip, pmx_train.hip), projector "
                "conv + positional table and both heads' small ends as own kernels (pmx_actor.hip, pmx_heads.hip); the first linear of the "
                "two heads on hipBLASLt")


def _progress(msg):
    """Stage marks on stderr (rank 0): a run that takes minutes must not look hung, and a stage that IS hung must be nameable."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"bench: {msg} [{time.strftime('%H:%M:%S')}]", file=sys.stderr, flush=True)


def _safe(fn, *a, **kw):
    _progress(f"{fn.__name__} minibatch {a[7] if len(a) > 7 else '-'} " + " ".join(f"{k}={v}" for k, v in kw.items()))
    try:
        return fn(*a, **kw)
    except Exception as e:                      # a probe must not lose the bench line
        return {"error": f"{type(e).__name__}: {e}", "minibatch_per_gpu": a[7] if len(a) > 7 else None}


def ppo_probe(layname, layout, dev, rank=0, world=1, dist=None, n_envs=16384, horizon=32, large_minibatch=16384, rehearsal_dist=None,
              algorithms=("mappo",), float32_too=False):
    """The second half of BASELINE.json's metric: end-to-end MAPPO (rollout with policy inference + GAE + 3 PPO epochs) on the
    workload, once at the reference's minibatch of 512 samples per GPU (pacman_mappo_resnet.py:18; replayed from hipGraphs, where
    the step is launch-bound) and once at a large minibatch (fewer, larger optimizer steps -- a different optimisation schedule
    than the reference's, stated as such).  On one GPU the data-parallel step is rehearsed as well (one-rank RCCL group)."""
    runs, rehearsal, other = [], [], {}
    # one GPU: both steps are replayed from a hipGraph (at 16 384 samples that is worth 2.6 % on smallCapture and nothing on the 20 x 20
    # boards: the step is GPU-bound); data parallel, the large step stays eager -- its two slice all-reduces overlap the backward
    for mb, graph in ((512, True), (large_minibatch, world == 1)):
        runs.append(_safe(e2e_probe, layout, dev, rank, world, dist, n_envs, horizon, mb, graph))
    if world == 1 and rehearsal_dist is not None:
        for mb, graph in ((512, True), (large_minibatch, False)):
            rehearsal.append(_safe(e2e_probe, layout, dev, rank, world, rehearsal_dist, n_envs, horizon, mb, graph, rehearse_dp=True))
    for alg in algorithms:
        if alg != "mappo":
            other[alg] = _safe(e2e_probe, layout, dev, rank, world, dist, n_envs, horizon, large_minibatch, False, algorithm=alg)
    if world == 1 and float32_too:
        # the reference's own arithmetic (float32 planes, float32 network on the library ops: no hand-written network kernel runs), on a
        # quarter of the envs to bound the run: what the bf16 figures above are a speed-up OVER, on the same GPU
        f32 = _safe(e2e_probe, layout, dev, rank, world, dist, max(n_envs // 4, 512), horizon, large_minibatch, False, autocast=False)
        f32["note"] = "float32 planes and network (torch library ops), no autocast; a quarter of the envs of the bf16 rows"
        other["float32_end_to_end"] = f32
    ref, big = runs
    out = {"config": f"{layname}, {n_envs} envs/GPU, horizon {horizon}, 3 epochs, paired minibatches (the centralised critic runs once "
                     "per env-tick pair), bf16 autocast, byte observation planes, in-kernel randomTeam opponent",
           "end_to_end": runs,
           "end_to_end_env_steps_per_s": big.get("end_to_end_env_steps_per_s"), "end_to_end_minibatch": large_minibatch,
           "end_to_end_env_steps_per_s_mb512": ref.get("end_to_end_env_steps_per_s"),
           "optimizer_steps_per_s": ref.get("optimizer_steps_per_s"), "samples_per_gpu_per_step": 512,
           # BASELINE.json's "PPO updates/s": one update = the rollout of n_envs x horizon env-ticks + GAE + 3 epochs over it
           "ppo_updates_per_s": big.get("ppo_updates_per_s"), "ppo_updates_per_s_mb512": ref.get("ppo_updates_per_s"),
           "rollout_env_steps_per_s": big.get("rollout_env_steps_per_s"),
           "network": NETWORK_NOTE,
           "reference_cpu": "0.49 s per optimizer step (2 steps/s) and about 80 env-steps/s end to end on 8 host cores, smallCapture; 1.24 s "
                            "per step and about 25 env-steps/s on the 20x20 board (tools/time_reference.py, "
                            "profiles/r01_cpu_reference_ratio.json)"}
    if rehearsal:
        out["data_parallel_rehearsal_one_rank"] = rehearsal
    out.update(other)
    return out


def config5_probe(dev, rank, world, dist, n_envs=8192, horizon=32, minibatch=16384):
    """BASELINE config 5 per GPU: 8 192 envs, each on its own generated 20x20 maze (mazeGenerator seeds 1..), MAPPO and IPPO end to
    end (IPPO = the same network with the critic fed each agent's own observation; the reference has no IPPO, so that row has no
    parity target)."""
    import pmx
    from pmx import maze_generator
    lays = [pmx.Layout.from_text(maze_generator.generate_maze(seed)) for seed in range(rank * n_envs + 1, (rank + 1) * n_envs + 1)]
    out = {"config": f"{n_envs} envs/GPU each on its own generated 20x20 maze (seeds {rank * n_envs + 1}..), horizon {horizon}, 3 epochs, "
                     f"{minibatch}-sample minibatches per GPU, bf16 autocast, byte observation planes, in-kernel randomTeam opponent"}
    for alg in ("mappo", "ippo"):
        out[alg] = _safe(e2e_probe, lays, dev, rank, world, dist, n_envs, horizon, minibatch, False, algorithm=alg)
    return out


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
    ap.add_argument("--no-dp-rehearsal", action="store_true", help="skip the one-rank RCCL rehearsal of the data-parallel optimizer step")
    ap.add_argument("--no-emit", action="store_true", help="skip the pmx_emit_team_obs (training-loop observation kernel) pass")
    ap.add_argument("--no-config5", action="store_true", help="skip the 20x20 generated-maze MAPPO / IPPO probe of the default run")
    ap.add_argument("--float32-e2e", action="store_true",
                    help="add the float32 / library-op end-to-end figure (ppo.float32_end_to_end); not in the default run: MIOpen's "
                         "solver search for the float32 convolutions can take minutes on a fresh box")
    ap.add_argument("--fixed-sweep", action="store_true",
                    help="run EVERY pass with the sweep direction fixed (all bytes to HBM): for profiler runs, so that the trace's average "
                         "duration of the expansion kernel is the one `roofline` is quoted on")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON: whatever libraries print on file descriptor 1 meanwhile (RCCL's version banner at
    # communicator creation, MIOpen's notes) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
    # the action stream: uniform over the 5 actions from the device Philox generator, resident before timing
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = 64
    actions = torch.randint(0, 5, (n_act, n_envs, 4), generator=g, device=dev, dtype=torch.int8)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # Every timed pass below except `cache_assisted` runs with the sweep direction of the expansion kernel FIXED
    # (pmx_set_tuning "expand_alt" 0), so that every byte of the planes has to reach HBM.  The library's default alternates the
    # direction, and a loop like this one -- the same buffer re-stepped with nothing in between -- then overwrites the tail of
    # tick t while it is still in the 256 MiB Infinity Cache: 16 % faster here, invisible to a caller that consumes the planes.
    env.set_tuning("expand_alt", 0)
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

    # roofline pass: start/stop HIP events attached to each kernel dispatch (pmx_profile_begin/end), over AT LEAST 500 launches
    # whatever --steps is (a 20-launch average moved by 10 % from run to run), in segments whose spread is reported
    k_roof, seg = max(args.steps, ROOFLINE_MIN_LAUNCHES), 125
    seg_us, exp_ms, exp_n, rule_ms, rule_n = [], 0.0, 0, 0.0, 0
    done = 0
    while done < k_roof:
        n = min(seg, k_roof - done)
        env.profile_begin(n + 8)
        for k in range(n):
            env.step(actions[(done + k) % n_act])
        pr = env.profile_end()
        seg_us.append(pr["expand_ms"] * 1e3 / max(pr["expand_launches"], 1))
        exp_ms += pr["expand_ms"]; exp_n += pr["expand_launches"]; rule_ms += pr["rule_ms"]; rule_n += pr["rule_launches"]
        done += n
    prof_uni = {"expand_ms": exp_ms, "expand_launches": exp_n, "rule_ms": rule_ms, "rule_launches": rule_n}
    # the library's default path in this loop (alternating sweep): reported as `cache_assisted`, never as `value`
    k_alt = min(max(args.steps, 100), 500)
    env.set_tuning("expand_alt", 0 if args.fixed_sweep else -1)
    for k in range(20):
        env.step(actions[k % n_act])
    torch.cuda.synchronize(dev)
    t_alt0 = time.perf_counter()
    for k in range(k_alt):
        env.step(actions[k % n_act])
    torch.cuda.synchronize(dev)
    tick_alt_s = (time.perf_counter() - t_alt0) / k_alt
    env.profile_begin(k_alt + 8)
    for k in range(k_alt):
        env.step(actions[k % n_act])
    prof = env.profile_end()
    # the default sweep with a CONSUMER between ticks that reads the planes and writes a rollout slot (what a training loop
    # does first with every tick's observations): the cache holds the consumer's traffic then
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
    # the observation kernel of the TRAINING loop: pmx_emit_team_obs on byte planes (two learners' canonicalised planes + the
    # merged critic input per env-tick, written straight into the rollout buffers); the four-agent expansion above is the
    # reference-dtype parity path and is not launched by the trainer at all
    emit = None
    if not args.no_emit:
        _progress("emit_team_probe")
        emit = emit_team_probe(lay, n_envs, length, dev, rank, actions)
        try:
            if n_envs == WORKLOADS[args.workload][1]:
                emit["traffic"] = json.load(open(tpath)).get(f"{args.workload}:emit_uint8", {}).get("expand_hbm_bytes_per_launch")
        except Exception:
            pass

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
            "tick_all_bytes_to_hbm": {"us_per_tick": dt / args.steps * 1e6, "GBps": n_envs * B / (dt / args.steps) / 1e9,
                                      "frac_of_peak": n_envs * B / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                                      "note": "whole tick (rule + expansion kernels, host launch path) with the sweep direction fixed: "
                                              "the loop `value` and `ms_per_step` are taken from"},
            "roofline": {"bound": "hbm", "kernel": "pmx_expand_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_with": "sweep direction fixed (pmx_set_tuning expand_alt 0): every byte of the planes reaches HBM",
                         "traffic_scope": "PMC TCC FETCH_SIZE/WRITE_SIZE = L2 <-> fabric requests; the Infinity Cache sits behind them",
                         "algorithmic_bytes_per_launch": expand_bytes, "avg_launch_us": expand_s * 1e6,
                         "launches": prof_uni["expand_launches"], "rule_kernel_avg_us": rule_s * 1e6,
                         "segment_avg_us_min_max": [min(seg_us), max(seg_us)],
                         "frac_min_max": [expand_bytes / (max(seg_us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                          expand_bytes / (min(seg_us) * 1e-6) / 1e9 / HBM_PEAK_GBS],
                         "same_device_fill_GBps": write_ceiling,
                         "cache_assisted": {
                             "avg_launch_us": expand_alt_s * 1e6, "GBps": expand_bytes / expand_alt_s / 1e9,
                             "us_per_tick": tick_alt_s * 1e6, "env_steps_per_s": n_envs / tick_alt_s,
                             "note": "the library's default path in THIS loop (same buffer re-stepped, nothing in between): the sweep "
                                     "direction alternates and tick t+1 overwrites the tail of tick t while it is still in the 256 MiB "
                                     "Infinity Cache; algorithmic bytes / time, NOT an HBM rate, and not what `value` is quoted on"},
                         "with_consumer": None if prof_cons is None else {
                             "avg_launch_us": prof_cons["expand_ms"] * 1e3 / max(prof_cons["expand_launches"], 1),
                             "GBps": expand_bytes / (prof_cons["expand_ms"] / 1e3 / max(prof_cons["expand_launches"], 1)) / 1e9,
                             "note": "default path with a copy of the planes into a rollout slot between ticks (reads them, writes as "
                                     "many bytes elsewhere)"}},
            "obs_checksum": checksum,
        }
        if emit is not None:
            line["roofline_emit_team"] = emit
        if not args.no_cpu_baseline and world == 1:
            _progress("cpu_baseline")
            line["cpu_baseline"] = cpu_baseline(lay[0].text if isinstance(lay, list) else lay.text, length, layname=layname)
    env.close()
    ppo = cfg5 = None
    if not args.no_ppo:
        rehearsal_dist = None
        if world == 1 and dist is None and not args.no_dp_rehearsal:
            try:            # a one-rank RCCL group for the data-parallel rehearsal of the optimizer step
                import socket
                import torch.distributed as rdist
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
                rdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                         device_id=torch.device("cuda", local))
                rehearsal_dist = rdist
            except Exception as e:
                print(f"bench: no one-rank RCCL group for the rehearsal ({type(e).__name__}: {e})", file=sys.stderr)
        elif dist is not None and world == 1:
            rehearsal_dist = dist
        if layname == "mazeGenerator" or layname == "bloxCapture":
            ppo = ppo_probe(layname, lay, dev, rank, world, dist, n_envs=n_envs, rehearsal_dist=None, algorithms=("mappo", "ippo"))
        else:
            ppo = ppo_probe(layname, lay, dev, rank, world, dist, n_envs=n_envs, rehearsal_dist=rehearsal_dist,
                            float32_too=args.float32_e2e)
            if not args.no_config5:
                cfg5 = config5_probe(dev, rank, world, dist)
        if rehearsal_dist is not None and dist is None:
            rehearsal_dist.destroy_process_group()
    if rank == 0:
        if ppo is not None:
            line["ppo"] = ppo
        if cfg5 is not None:
            line["ppo_config5"] = cfg5
        os.write(real_stdout, (json.dumps(line) + "\n").encode())           # the ONE line of stdout
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
