#!/bin/bash
# Development aid: tools/ab_build.sh <name> <source.hip> [extra hipcc flags]  ->  ab/<name>.so, a libpmx_hip.so whose <source> object
# was compiled with the extra flags (same source hash, so the loader accepts it: cp ab/<name>.so pacman-marl-2025_amd/libpmx_hip.so).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; SRC=$2; shift 2
cd $ROOT/pacman-marl-2025_amd
mkdir -p $ROOT/ab
OBJ=$ROOT/ab/$NAME.${SRC%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c csrc/$SRC -o $OBJ -Rpass-analysis=kernel-resource-usage 2>&1 | grep -c "error" || true
OBJS=""
for o in pmx_step pmx_api pmx_train pmx_actor pmx_critic pmx_heads; do
  if [ "$o.hip" == "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS build/pmx_stamp.cpp -o $ROOT/ab/$NAME.so
echo "built ab/$NAME.so"
