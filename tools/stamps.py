"""Per-section shader-clock shares of flight_step_kernel from the -DFFE_STAMPS diagnostic build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["FLYBODY_ENV_LIB"] = os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "csrc", "variants", "libflybody_env_stamps.so")
import torch
from flybody_amd import _capi
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

NAMES = ["s1 kinematics+com+cinert", "s1 cdof+velocities", "s1 rne fwd+fluid", "s1 subtree sums+joint space", "factor: M entries", "factor: elimination",
         "stage2 glue (limits, actuation)", "triangular solves", "integration+sensors+ghost+prologue", "epilogue obs/reward", "store", "prologue: action mix, wing targets, ghost, actuator base", "sensor accumulation + actuation", "limit instantiation", "constraint block tail", "prologue: launch, state load, WBPG step", "collision (position stage)", "  of it: geom frames + bounding spheres", "  of it: separating-direction bounds + cache", "  of it: narrow phase"]
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
B = 8192
env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0)
L = _capi.lib()
L.ffe_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
env.reset()
spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
for _ in range(10): env.step(a)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 20)()
L.ffe_debug_read_stamps(buf, 1)
n = 20
for _ in range(n): env.step(a)
torch.cuda.synchronize()
L.ffe_debug_read_stamps(buf, 1)
tot = sum(buf[:17])  # (17-19 are parts of 16)
print(f"total shader clocks per wave-step: {tot / (n * B):.0f}")
for k, name in enumerate(NAMES):
    print(f"{name:40s} {buf[k] / (n * B):10.0f} clk/wave-step  {100.0 * buf[k] / tot:5.1f} %")
