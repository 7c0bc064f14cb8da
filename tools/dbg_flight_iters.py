"""Diagnostic (dbgcf build): per-substep solver passes, contacts and how many of them push, over a random-action rollout of 64 envs."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("FLYBODY_ENV_LIB", os.path.join(ROOT, "flybody_amd", "csrc", "variants", "libflybody_env_dbgcf.so"))
from flybody_amd import _capi
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
B = 64
env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0)
env.reset()
spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
buf = np.zeros((64, 76), np.float32)
hist = {}
for k in range(300):
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    env.step(a)
    torch.cuda.synchronize()
    _capi.lib().ffe_debug_read_cf(buf.ctypes.data_as(C.POINTER(C.c_float)))
    for i in range(B):
        key = (int(buf[i, 0]), bin(int(buf[i, 1])).count("1"), int(buf[i, 2]))
        hist[key] = hist.get(key, 0) + 1
tot = sum(hist.values())
print("(contacts, pushing at convergence, passes of the last substep's solve): share")
for key, v in sorted(hist.items(), key=lambda kv: -kv[1])[:25]:
    print(key, f"{100.0 * v / tot:.1f} %")
print("mean passes", sum(k[2] * v for k, v in hist.items()) / tot, "mean contacts", sum(k[0] * v for k, v in hist.items()) / tot, "mean pushing", sum(k[1] * v for k, v in hist.items()) / tot)
