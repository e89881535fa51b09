#!/bin/bash
# the 16 384-sample step eager against replayed from a hipGraph: is any of its 6.3 ms the host's?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { python tools/train_bench.py --layout $1 --envs $2 --horizon 16 --minibatch 16384 --updates 4 $3 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['runs'][1:]
print('  update_s', ['%.4f'%x['update_s'] for x in r])"; }
for cfg in "smallCapture 8192" "mazes 2048"; do
  echo "== $cfg eager"; run $cfg ""
  echo "== $cfg graph"; run $cfg --graph
done
