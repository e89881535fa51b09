#!/bin/bash
# rocprofv3 kernel-trace summaries of the optimizer step: small / mazes at 16 384-sample minibatches, small at 512 (graph replay).
# usage: tools/r03_prof.sh <outdir under gpurun_out>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r3prof}
mkdir -p $O
cd /tmp
for cfg in "small smallCapture 8192" "mazes mazes 2048"; do
  set -- $cfg
  rm -rf /tmp/prof_$1
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python3 $ROOT/tools/train_bench.py --layout $2 --envs $3 --horizon 16 --minibatch 16384 --updates 3 > /tmp/prof_$1.log 2>&1; echo "prof $1 rc=$?"
  python3 $ROOT/tools/prof_summary.py $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) "" 70 > $O/kernel_stats_train_step_$1_mb16384.txt
  grep "^{" /tmp/prof_$1.log | tail -1 > $O/train_bench_of_the_profiled_run_$1.json
done
cd $ROOT
tools/step512_prof.sh 80 > $O/kernel_stats_train_step_mb512_graph.txt 2>&1 || echo "step512 prof failed"
