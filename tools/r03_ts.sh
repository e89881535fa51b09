#!/bin/bash
# two streams (actor / critic halves of the step side by side) against one stream at 16 384-sample minibatches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { python tools/train_bench.py --layout $1 --envs $2 --horizon 16 --minibatch 16384 --updates 4 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['runs'][1:]
print('  update_s', ['%.4f'%x['update_s'] for x in r], 'steps', r[-1].get('optimizer_steps'), {k:v for k,v in d.items() if k not in ('config','runs')})"; }
for lay in "smallCapture 8192" "mazes 2048"; do
  echo "== $lay two streams"; run $lay; run $lay
  echo "== $lay one stream"; PMX_NO_TWO_STREAMS=1 run $lay; PMX_NO_TWO_STREAMS=1 run $lay
done
