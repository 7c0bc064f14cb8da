#!/bin/bash
# Diagnostic builds of the library (never timed or shipped): per-section shader-clock stamps for tools/stamps.py (flight)
# and tools/ball_stamps.py (walk_on_ball).  Same flags as flybody_amd/build.py plus the stamp macro.
set -e
cd "$(dirname "$0")/.."
mkdir -p flybody_amd/csrc/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
SRC="flybody_amd/csrc/fly_env.hip flybody_amd/csrc/ball_env.hip flybody_amd/csrc/nstep.hip"
/opt/rocm/bin/hipcc $FLAGS -DFFE_STAMPS -o flybody_amd/csrc/variants/libflybody_env_stamps.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DFFB_STAMPS -o flybody_amd/csrc/variants/libflybody_env_bstamps.so $SRC
ls -la flybody_amd/csrc/variants
