#!/bin/bash
# Learning check of the graph-replayed step at the reference's minibatch of 512 (gather + scalars in one launch, folded row sums, unit root,
# optimizer tail): the compressed curriculum on smallCapture, 61 updates
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3learn; mkdir -p $O
timeout -k 10 800 python tools/train.py --layout smallCapture --envs 2048 --horizon 64 --minibatch 512 --graph --updates 61 --opponent curriculum --curriculum-scale 0.1 --eval-every 20 --log $O/train_small_curriculum_mb512_graph.jsonl 2>&1 | grep --line-buffered "eval" | cut -c1-230
echo "rc=$?"
python - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/r3learn/train_small_curriculum_mb512_graph.jsonl") if l.strip()]
ups = [r for r in rows if "sec" in r]
print(len(ups), "updates", "%.0f s" % sum(r["sec"] for r in ups), "median env-steps/s %.3e" % statistics.median(r["env_steps_per_s"] for r in ups[3:]), "steps per update", ups[-1].get("steps"))
PY
