"""Does the expansion kernel's time depend on where the planes live?  One process, several PmxVecEnv instances of the bench workload
(smallCapture 16 384, float32 planes, fixed sweep), each created after a differently sized allocation so that its 323 MB observation
buffer lands at another address; 500 timed launches each.  usage (GPU box): python tools/placement_probe.py [trials]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmx

if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1234)
    actions = torch.randint(0, 5, (64, 16384, 4), generator=g, device=dev, dtype=torch.int8)
    keep = []
    for t in range(trials):
        if t:
            keep.append(torch.empty(((37 * t) % 200 + 1) << 20, dtype=torch.uint8, device=dev))   # shifts the next allocation
        if t == trials - 2:
            keep.clear(); torch.cuda.empty_cache()
        env = pmx.PmxVecEnv(pmx.get_layout("smallCapture"), 16384, length=300, auto_reset=True, obs_dtype="float32", device=dev)
        env.reset()
        env.set_tuning("expand_alt", 0)
        for k in range(50):
            env.step(actions[k % 64])
        us = []
        for s in range(4):
            env.profile_begin(125 + 8)
            for k in range(125):
                env.step(actions[k % 64])
            pr = env.profile_end()
            us.append(round(pr["expand_ms"] * 1e3 / pr["expand_launches"], 2))
        print(json.dumps({"trial": t, "obs_ptr": hex(env.obs.data_ptr()), "ptr_mod_2MiB": env.obs.data_ptr() % (2 << 20),
                          "expand_us_segments": us, "reserved_MiB": torch.cuda.memory_reserved(dev) >> 20}), flush=True)
        env.close()
        del env
