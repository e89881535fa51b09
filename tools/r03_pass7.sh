#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -q -x > $O/pytest_trainer.log 2>&1; echo "trainer tests rc=$?"; tail -3 $O/pytest_trainer.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-emit --no-config5 --no-unidirectional > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3h/bench_small.json").read().strip().splitlines()[-1])
p = d["ppo"]
for e in p["end_to_end"]: print("e2e", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("finite"), e.get("error"))
for e in p.get("data_parallel_rehearsal_one_rank", []): print("dp ", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("hipgraphs_per_step"), e.get("finite"), e.get("error"))
PY
for v in "" "PMX_ACTOR_SPLIT_MAX=1000000 PMX_ACTOR_SPLIT_BWD_MAX=1000000 PMX_ACTOR_SPLIT_WAVES=4"; do echo "== $v"; env $v python tools/actor_bench.py --batch 8192 16384 --layout smallCapture --iters 10 --no-library 2>/dev/null | cut -c1-190; done | tee $O/actor_split_ab.txt
