"""Times env.step of either task with given physics flags (e.g. 64 = FFE_NO_CONTACT) on the bench workload: what a stage costs.
    python tools/bench_flags.py flight 0 64"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from flybody_amd import fly_envs  # noqa: E402

kind = sys.argv[1]
for fl in [int(x) for x in sys.argv[2:]]:
    B = 8192 if kind == "flight" else 4096
    env = fly_envs.flight_imitation(batch_size=B, random_state=0, physics_flags=fl) if kind == "flight" else fly_envs.walk_on_ball(batch_size=B, physics_flags=fl)
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum, device="cuda"), torch.tensor(spec.maximum, device="cuda")
    if kind != "flight":
        lo, hi = torch.full_like(lo, -0.2), torch.full_like(hi, 0.2)
    g = torch.Generator(device="cuda").manual_seed(1234)
    acts = [(lo + (hi - lo) * torch.rand(B, spec.shape[0], device="cuda", generator=g)).contiguous() for _ in range(16)]
    env.reset()
    for k in range(60):
        env.step(acts[k % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for k in range(n):
        env.step(acts[k % 16])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{kind} flags {fl}: {dt * 1e3:.4f} ms/step, {B / dt / 1e6:.3f} M env-steps/s")
    env.close()
