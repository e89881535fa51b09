#!/bin/bash
# bloxCapture (the layout the reference trains on) with a less compressed curriculum: thresholds at 60 / 240 updates, 320 updates
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3learn; mkdir -p $O
timeout -k 10 1000 python tools/train.py --layout bloxCapture --envs 2048 --horizon 64 --minibatch 8192 --updates 321 --opponent curriculum --curriculum-scale 0.3 --eval-every 40 --log $O/train_blox_curriculum_scale03.jsonl 2>&1 | grep --line-buffered "eval" | cut -c1-230
echo "rc=$?"
