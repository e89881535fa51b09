#!/bin/bash
# Learning check on the round's final kernels: the reference's curriculum, compressed (DESIGN 5.2's recipe), smallCapture and bloxCapture
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3learn; mkdir -p $O
timeout -k 10 500 python tools/train.py --layout smallCapture --envs 4096 --horizon 64 --minibatch 8192 --updates 141 --opponent curriculum --curriculum-scale 0.1 --eval-every 20 --log $O/train_small_curriculum.jsonl > $O/train_small.out 2>&1; echo "small rc=$?"
grep -c . $O/train_small_curriculum.jsonl; grep "eval" $O/train_small.out | tail -8 | cut -c1-250
timeout -k 10 600 python tools/train.py --layout bloxCapture --envs 2048 --horizon 64 --minibatch 8192 --updates 141 --opponent curriculum --curriculum-scale 0.1 --eval-every 35 --log $O/train_blox_curriculum.jsonl > $O/train_blox.out 2>&1; echo "blox rc=$?"
grep "eval" $O/train_blox.out | tail -6 | cut -c1-250
