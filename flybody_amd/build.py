"""Builds the HIP shared library in-tree (hipcc cross-compiles gfx950 without a GPU present)."""

from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libflybody_env.so")
SOURCES = ["fly_env.hip", "ball_env.hip", "nstep.hip"]
# Loop-invariant code motion hoists per-lane LDS addresses and literal constants out of the substep loop of the two step
# kernels and then spills them.  flight: no machine LICM -> 119 VGPRs, 0 B scratch (was 56 B; +0.5 % env-steps/s);
# walk_on_ball: sinking invariants back into the loop where that avoids a spill -> 176 B scratch (was 548 B; +1.2 %).  Measured
# each way on both kernels (profiles/r02_compiler_flag_ab.log).
PER_SOURCE_FLAGS = {
    "fly_env.hip": ["-mllvm", "-disable-machine-licm"],
    "ball_env.hip": ["-mllvm", "-sink-insts-to-avoid-spills=1"],
}
HEADERS = ["dev_model.hpp", "ball_model.hpp", "ball_env.hpp", "dev_math.hpp", "convex.hpp", "launch_order.hpp", os.path.join("..", "..", "include", "flybody_env.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: the SLP vectoriser packs scalar FMAs whose multiplier sits in an SGPR (v_readlane broadcasts) into
    # v_pk_mul + moves, which costs more than it saves in both kernels (+3 % env-steps/s without it, measured).
    # iterative-ilp scheduling: the default scheduler serialises every v_readlane -> s_nop -> v_fma pair of the unrolled dense
    # loops through one SGPR; the ILP scheduler batches the broadcasts (walk_on_ball +8.6 %, flight +0.8 %, measured)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
              "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, "_obj_" + os.path.splitext(src)[0] + ".o")
        cmd = common + PER_SOURCE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    for obj in objs:
        os.remove(obj)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
