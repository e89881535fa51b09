#!/bin/bash
# Development aid (GPU box, repo root): tools/ab_run.sh <variant> ...  -> per-kernel averages of the actor tower for each ab/<variant>.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cp $ROOT/pacman-marl-2025_amd/libpmx_hip.so /tmp/libpmx_orig.so
for v in "$@"; do
  cp $ROOT/ab/$v.so $ROOT/pacman-marl-2025_amd/libpmx_hip.so
  echo "== $v"
  $ROOT/tools/actor_prof.sh 8192 | grep "fwd\|bwd"
done
cp /tmp/libpmx_orig.so $ROOT/pacman-marl-2025_amd/libpmx_hip.so
