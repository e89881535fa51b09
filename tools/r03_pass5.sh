#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -x -q -k "tail or heads or projector" > $O/pytest_heads.log 2>&1; echo "heads rc=$?"; grep -E "^E  |passed|failed" $O/pytest_heads.log | head -12
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/pytest_gpu.log
timeout -k 10 300 python tools/train_bench.py --envs 16384 --horizon 32 --minibatch 16384 --updates 2 > $O/train_small.json 2>/dev/null; tail -c 330 $O/train_small.json; echo
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-emit --no-config5 --no-unidirectional > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3f/bench_small.json").read().strip().splitlines()[-1])
p = d["ppo"]
for e in p["end_to_end"]: print("e2e", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("error"))
for e in p.get("data_parallel_rehearsal_one_rank", []): print("dp ", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("hipgraphs_per_step"), e.get("error"))
PY
