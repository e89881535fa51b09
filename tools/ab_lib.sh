#!/bin/bash
# Development aid (GPU box, repo root): tools/ab_lib.sh "<command>" <variant> ...  -> runs the command once per ab/<variant>.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=$1; shift
cp $ROOT/pacman-marl-2025_amd/libpmx_hip.so /tmp/libpmx_orig.so
for v in "$@"; do
  cp $ROOT/ab/$v.so $ROOT/pacman-marl-2025_amd/libpmx_hip.so
  echo "== $v"
  bash -c "$CMD"
done
cp /tmp/libpmx_orig.so $ROOT/pacman-marl-2025_amd/libpmx_hip.so
