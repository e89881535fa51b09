#!/bin/bash
# graph capture right behind RCCL traffic (the one-rank data-parallel rehearsal), repeated on the fastest workload: the capture must
# survive the ProcessGroupNCCL watchdog's event polling (thread-local capture mode, PPOLearner.capture)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3lines; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --workload tiny4096 --no-cpu-baseline --no-emit --steps 100 --warmup 10 > $O/bench_tiny4096.json 2> $O/bench_tiny4096.err; echo "tiny run $i rc=$?"
done
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3lines/bench_tiny4096.json").read().strip())
print(d["ppo"]["config"][:60])
for e in d["ppo"]["end_to_end"]:
    print(e.get("minibatch_per_gpu"), e.get("envs_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("error"))
for e in d["ppo"]["data_parallel_rehearsal_one_rank"]:
    print("dp", e.get("minibatch_per_gpu"), e.get("end_to_end_env_steps_per_s"), e.get("optimizer_steps_per_s"), e.get("error"))
PY
timeout -k 10 400 python -m pytest tests/test_gpu_trainer.py -q -k "graph or gather" 2>&1 | tail -1
