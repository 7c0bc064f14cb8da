#!/bin/bash
# Diagnostic builds of the library (never timed or shipped): per-section shader-clock stamps for tools/stamps.py (flight)
# and tools/ball_stamps.py (walk_on_ball), and the timing-ablation build for tools/ablate.py.  Same flags as
# flybody_amd/build.py (incl. its per-source ones) plus the diagnostic macro.
set -e
cd "$(dirname "$0")/.."
mkdir -p flybody_amd/csrc/variants
C=flybody_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
variant() {  # name, extra macro flags
  local name=$1; shift
  local T=$(mktemp -d)
  /opt/rocm/bin/hipcc $FLAGS -mllvm -disable-machine-licm "$@" -c $C/fly_env.hip -o $T/fly.o
  /opt/rocm/bin/hipcc $FLAGS -mllvm -sink-insts-to-avoid-spills=1 "$@" -c $C/ball_env.hip -o $T/ball.o
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/nstep.hip -o $T/nstep.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/libflybody_env_$name.so $T/fly.o $T/ball.o $T/nstep.o
  rm -rf $T
}
variant stamps -DFFE_STAMPS
variant bstamps -DFFB_STAMPS
variant ablation -DFFE_ABLATION
variant trace -DFFE_TRACE
variant dbgcf -DFFE_DBGCF   # contact rows + forces of the last solve of envs 0..63 (tools/dbg_forced.py, tools/dbg_flight_iters.py)
ls -la $C/variants
