"""Records per-env solver work of the flight step over consecutive control steps of the bench workload (fixed full-range
actions), to evaluate launch-order predictors offline.   python tools/cost_history.py out.npz [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

out = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = 8192
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0)
spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
env.reset()
for _ in range(50): env.step(a)
ints_h = np.zeros((steps, B, 8), dtype=np.int32)
qpos_h = np.zeros((steps, B, 6), dtype=np.float32)
for k in range(steps):
    ts = env.step(a)
    ints, _ = env.get_task_state()
    ints_h[k] = ints.cpu().numpy()
np.savez_compressed(out, ints=ints_h, tab_off=np.asarray(tables.tab_off) if hasattr(tables, "tab_off") else np.zeros(1))
print("saved", out, ints_h.shape, "mean iters", ints_h[..., 6].mean(), "active frac", (ints_h[..., 6] > 0).mean())
