"""Is some launch variant of the float32 expansion less sensitive to the observation buffer's mapping than the default?
One process, several fresh mappings (handle closed, allocator cache emptied), every variant timed on each mapping (125 launches,
fixed sweep).  usage (GPU box): python tools/variant_probe.py [mappings]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmx

VARIANTS = [
    ("default", {}),
    ("cap12000", {"expand_lds_pad": 12000}),
    ("nt_cap24000", {"expand_nt": 1, "expand_lds_pad": 24000}),
    ("nt_cap30000", {"expand_nt": 1, "expand_lds_pad": 30000}),
    ("nt_cap40000", {"expand_nt": 1, "expand_lds_pad": 40000}),
    ("nt_cap50000", {"expand_nt": 1, "expand_lds_pad": 50000}),
    ("nt_cap70000", {"expand_nt": 1, "expand_lds_pad": 70000}),
    ("wave_per_env_nt_cap40000", {"expand_wave_per_env": 1, "expand_nt": 1, "expand_lds_pad": 40000}),
]
KEYS = ["expand_lds_pad", "expand_wave_per_env", "expand_nt"]

if __name__ == "__main__":
    mappings = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1234)
    actions = torch.randint(0, 5, (64, 16384, 4), generator=g, device=dev, dtype=torch.int8)
    keep = []
    for m in range(mappings):
        torch.cuda.empty_cache()
        keep.append(torch.empty(((53 * m) % 90 + 3) << 20, dtype=torch.uint8, device=dev))      # the next mapping lands elsewhere
        env = pmx.PmxVecEnv(pmx.get_layout("smallCapture"), 16384, length=300, auto_reset=True, obs_dtype="float32", device=dev)
        env.reset()
        env.set_tuning("expand_alt", 0)
        row = {"mapping": m, "obs_ptr": hex(env.obs.data_ptr())}
        for name, tun in VARIANTS:
            for k in KEYS:
                env.set_tuning(k, -1)
            for k, v in tun.items():
                env.set_tuning(k, v)
            for k in range(30):
                env.step(actions[k % 64])
            env.profile_begin(125 + 8)
            for k in range(125):
                env.step(actions[k % 64])
            pr = env.profile_end()
            row[name] = round(pr["expand_ms"] * 1e3 / pr["expand_launches"], 2)
        print(json.dumps(row), flush=True)
        env.close()
        del env
