#!/bin/bash
# Round-3 third pass: the whole GPU suite, then the default bench line (new structure: fixed-sweep value, 500-launch roofline,
# emit-team roofline, DP rehearsal, config-5 probe).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3c
mkdir -p $O
echo "(suite skipped in this rerun)"
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -5 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], r["frac_min_max"], "launches", r["launches"], "avg us", r["avg_launch_us"])
print("emit", {k: d["roofline_emit_team"][k] for k in ("achieved", "frac", "avg_launch_us", "launches")})
p = d["ppo"]
for e in p["end_to_end"]: print("e2e", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("error"))
for e in p.get("data_parallel_rehearsal_one_rank", []): print("dp ", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("hipgraphs_per_step"), e.get("error"))
c = d.get("ppo_config5", {})
for a in ("mappo", "ippo"): print("cfg5", a, c.get(a, {}).get("end_to_end_env_steps_per_s"), c.get(a, {}).get("error"))
PY
