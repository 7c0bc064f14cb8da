import sys, os
sys.path.insert(0, os.getcwd())
import torch
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
B=8192
for fl in (0, 1<<24):
    env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0, physics_flags=fl)
    env.reset()
    spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    for _ in range(50): env.step(a)
    ms = min(env.time_steps(a, 300) for _ in range(2))
    print(os.environ.get("FLYBODY_ENV_LIB","default"), "flags", fl, f"{ms:.4f} ms/step {B/ms/1e3:.3f} M/s", flush=True)
    env.close()
