#!/bin/bash
# Round-3 second pass: the new attention kernels (tests + A/B timing at 154 and 400 tokens) and the tower test that was fixed.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -x -q -k "attention or critic or end_to_end" > $O/pytest_attn.log 2>&1; echo "attn tests rc=$?"; tail -3 $O/pytest_attn.log
timeout -k 10 300 python -m pytest tests/test_gpu_actor_tower.py -x -q -k "sum_over" > $O/pytest_tower.log 2>&1; echo "tower rc=$?"; tail -2 $O/pytest_tower.log
for S in 154 400; do
  ATTN_S=$S timeout -k 10 120 python tools/attn_time.py >> $O/attn_time.txt 2>&1
  ATTN_S=$S PMX_ATTN_FWD_V1=1 PMX_ATTN_BWD_TWO_PASS=1 timeout -k 10 120 python tools/attn_time.py >> $O/attn_time.txt 2>&1
done
cat $O/attn_time.txt
timeout -k 10 400 python tools/train_bench.py --layout mazes --envs 4096 --horizon 32 --minibatch 16384 --updates 2 > $O/train_mazes4096.json 2> $O/train_mazes4096.err; echo "train mazes rc=$?"; tail -c 500 $O/train_mazes4096.json
