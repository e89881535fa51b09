#!/usr/bin/env python3
"""Times the small kernels of the path on one GPU: GAE (both variants), maze distances, canonicalize/merge."""
import ctypes as C, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pmx
lib = pmx._lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


out = {}
for (T, n, mode, tag) in ((32, 32768, 0, "gae_lane_T32_n32768"), (32, 131072, 0, "gae_lane_T32_n131072"),
                          (2048, 2, 1, "gae_wave_T2048_n2_reference_regime"), (2048, 2, 0, "gae_lane_T2048_n2"),
                          (2048, 256, 1, "gae_wave_T2048_n256")):
    rew = torch.randn(T, n, device="cuda"); val = torch.randn(T, n, device="cuda")
    done = (torch.rand(T, n, device="cuda") < 0.01).float(); last = torch.randn(n, device="cuda")
    adv = torch.empty_like(rew); ret = torch.empty_like(rew)
    us = timed(lambda: lib.pmx_gae_mode(rew.data_ptr(), val.data_ptr(), done.data_ptr(), last.data_ptr(), T, n, 0.99, 0.95,
                                        adv.data_ptr(), ret.data_ptr(), mode, st()))
    out[tag] = {"us": us, "GBps_algorithmic": 20.0 * T * n / us / 1e3}
for name in ("tinyCapture", "smallCapture", "bloxCapture"):
    env = pmx.PmxVecEnv(name, 1)
    t0 = time.perf_counter()
    for _ in range(20):
        env.maze_distances()
    torch.cuda.synchronize()
    out[f"maze_distances_{name}"] = {"ms_per_call_incl_host": (time.perf_counter() - t0) / 20 * 1e3}
    env.close()
print(json.dumps(out, indent=1))
