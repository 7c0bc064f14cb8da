"""Timing-only ablations of the step kernel (results are wrong under the DBG flags; for profiling).
The in-kernel ablation switches exist only in the diagnostic build: bash tools/build_variants.sh first."""
import sys, os
os.environ.setdefault("FLYBODY_ENV_LIB", os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "csrc", "variants", "libflybody_env_ablation.so"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
FLAGS = {"baseline": 0, "no_fluid": 1, "no_limit": 2, "no_fluid_no_limit": 3, "skip_factor": 1 << 16, "skip_solve": 1 << 17,
         "skip_factor_solve": 3 << 16, "skip_stage1_repeat": 1 << 18, "skip_Mentries": 1 << 19, "skip_all_linalg+stage1": 7 << 16, "skip_all+ghost": (7 << 16) | (1 << 20), "skip_all+ghost+wbpg": (7 << 16) | (3 << 20), "skip_all+ghost+wbpg+obs": (7 << 16) | (7 << 20), "skip_all+ghost+wbpg+obs+nolimit+nofluid": (7 << 16) | (7 << 20) | 3}
for name, fl in FLAGS.items():
    env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0, physics_flags=fl)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    a = ((torch.rand(B, 12, device="cuda", generator=g) * 2 - 1) * 0.3).contiguous()
    for _ in range(5):
        env.step(a)
    ms = env.time_steps(a, 40)
    print(f"{name:28s} {ms:8.4f} ms/step  {B / ms / 1e3:8.3f} M env-steps/s", flush=True)
    env.close()
