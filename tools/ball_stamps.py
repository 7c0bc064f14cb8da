"""Per-section shader-clock shares of the walk_on_ball kernel (diagnostic build -DFFB_STAMPS; never timed or shipped).

    FLYBODY_ENV_LIB=flybody_amd/csrc/variants/libflybody_env_bstamps.so python tools/ball_stamps.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch

from flybody_amd import _capi, fly_envs

NAMES = ["kinematics sweep", "velocity sweep", "body forces + subtree sweep", "joint forces + M assembly", "factor M and M + hB", "collision: ball vs leg capsules",
         "actuation + contact rows", "smooth forces", "rows + G (block solves incl. a_s)", "newton + noslip (dense, registers)", "collision: geom frames + ball vs convex", "collision: sphere / capsule pairs",
         "collision: convex pairs", "-", "-", "-", "-", "constraint forces + final/Euler solve", "sensors",
         "integration", "prologue + store"]
B = 4096
env = fly_envs.walk_on_ball(batch_size=B)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = [(torch.rand(B, 59, device="cuda", generator=g) * 0.4 - 0.2).contiguous() for _ in range(8)]
for k in range(10):
    env.step(acts[k % 8])
torch.cuda.synchronize()
L = _capi.lib()
buf = (C.c_ulonglong * 24)()
L.ffb_debug_read_stamps(buf, 1)
for k in range(10):
    env.step(acts[k % 8])
torch.cuda.synchronize()
L.ffb_debug_read_stamps(buf, 0)
tot = sum(buf[:21])
for n, v in zip(NAMES, buf[:21]):
    print(f"{n:32s} {100.0 * v / tot:6.2f} %")
