#!/usr/bin/env python3
"""Fills the R3_* placeholders of DESIGN.md / README.md from profiles/r03/*.json (the bench lines of the final measurement pass)."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(name):
    return json.loads(open(os.path.join(ROOT, "profiles", "r03", name)).read().strip().splitlines()[-1])
d, blox = load("bench_default.json"), load("bench_blox4096.json")
p, c5 = d["ppo"], d["ppo_config5"]
ref, big = p["end_to_end"]
dref, dbig = p["data_parallel_rehearsal_one_rank"]
def e(x): return f"{x:.3g}".replace("e+0", "e").replace("e+", "e")
def sci(x):
    m, ex = f"{x:.2e}".split("e"); return f"{m}e{int(ex)}"
v = {
 "R3_E2E_BIG": sci(big["end_to_end_env_steps_per_s"]), "R3_SPS_BIG": f"{big['optimizer_steps_per_s']:.0f}",
 "R3_E2E_512": sci(ref["end_to_end_env_steps_per_s"]), "R3_SPS_512": f"{ref['optimizer_steps_per_s']:,.0f}".replace(",", " "),
 "R3_DP_E2E_512": sci(dref["end_to_end_env_steps_per_s"]), "R3_DP_SPS_512": f"{dref['optimizer_steps_per_s']:,.0f}".replace(",", " "),
 "R3_DP_E2E_BIG": sci(dbig["end_to_end_env_steps_per_s"]),
 "R3_CFG5_MAPPO": sci(c5["mappo"]["end_to_end_env_steps_per_s"]), "R3_CFG5_IPPO": sci(c5["ippo"]["end_to_end_env_steps_per_s"]),
 "R3_CFG5_ROLL": f"{c5['mappo']['rollout_s']:.2f}", "R3_CFG5_UPD": f"{c5['mappo']['gae_plus_update_s']:.2f}",
 "R3_BLOX_BIG": sci(blox["ppo"]["end_to_end"][1]["end_to_end_env_steps_per_s"]), "R3_BLOX_512": sci(blox["ppo"]["end_to_end"][0]["end_to_end_env_steps_per_s"]),
 "R3_BLOX_SPS": f"{blox['ppo']['end_to_end'][0]['optimizer_steps_per_s']:.0f}",
 "R3_PROJ_BIG": sci(8 * big["end_to_end_env_steps_per_s"]), "R3_PROJ_512": sci(8 * dref["end_to_end_env_steps_per_s"]),
 "R3_TICK_VALUE": sci(d["value"]), "R3_TICK_US": f"{d['ms_per_step'] * 1e3:.1f}",
 "R3_EXP_US": f"{d['roofline']['avg_launch_us']:.1f}", "R3_EXP_FRAC": f"{d['roofline']['frac']:.2f}",
 "R3_EMIT_US": f"{d['roofline_emit_team']['avg_launch_us']:.1f}", "R3_EMIT_FRAC": f"{d['roofline_emit_team']['frac']:.2f}",
}
for f in ("DESIGN.md", "README.md"):
    path = os.path.join(ROOT, f)
    s = open(path).read()
    for k in sorted(v, key=len, reverse=True):
        s = s.replace(k, v[k])
    left = sorted(set(re.findall(r"R3_[A-Z0-9_]+", s)))
    open(path, "w").write(s)
    print(f, "filled;", "left:", left)
