import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pmx
N = 16384
for bots in (False, True):
    env = pmx.PmxVecEnv("smallCapture", N, length=300, auto_reset=True, obs_dtype="uint8", bots=bots)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(0, 5, (N, 4), generator=g, device="cuda", dtype=torch.int8)
    variants = {"random actions": a}
    if bots:
        b = a.clone(); b[:, 0] = -3; b[:, 2] = -4
        variants["red = baselineTeam (offense, defense)"] = b
        c = a.clone(); c[:, 0] = -2; c[:, 2] = -2
        variants["red = randomTeam in-kernel"] = c
    for name, act in variants.items():
        for _ in range(20): env.step(act)
        env.profile_begin(220)
        for _ in range(200): env.step(act)
        p = env.profile_end()
        print(json.dumps({"bots_handle": bots, "actions": name, "rule_us": p["rule_ms"] / p["rule_launches"] * 1e3, "expand_us": p["expand_ms"] / p["expand_launches"] * 1e3}))
    env.close()
