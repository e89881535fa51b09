#!/bin/bash
# Rebuilds libpmx_hip.so with the forward kernel's phase timers compiled in (development only; run `python -m pmx.build` style
# rebuild afterwards: python -c "from pmx import build; build.build(force=True)").
set -e
cd "$(dirname "$0")/../pacman-marl-2025_amd"
python -c "import sys; sys.path.insert(0, '..'); from pmx import build; build.build()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPMX_ACTOR_TIMING -c csrc/pmx_actor.hip -o build/pmx_actor.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared build/pmx_step.o build/pmx_api.o build/pmx_train.o build/pmx_actor.o build/pmx_critic.o build/pmx_stamp.cpp -o libpmx_hip.so
