"""Long-run health check of both step kernels at the bench batches: finite outputs, episodes rolling over, counters sane, two
identical handles staying bit-identical (determinism).    python tools/soak.py [flight_steps] [ball_steps]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from flybody_amd import fly_envs
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
B = 8192
envs = [BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=1) for _ in range(2)]
spec = envs[0].action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
ts = [e.reset() for e in envs]
bad = 0; last = 0; first = 0; same = True; rsum = 0.0; maxlen = 0
for k in range(fs):
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    ts = [e.step(a) for e in envs]
    o = envs[0].flat_observation
    bad += int((~torch.isfinite(o)).sum()) + int((~torch.isfinite(ts[0].reward)).sum())
    last += int((ts[0].step_type == 2).sum()); first += int((ts[0].step_type == 0).sum()); rsum += float(ts[0].reward.sum())
    if k % 50 == 0:
        same &= bool(torch.equal(o, envs[1].flat_observation)) and bool(torch.equal(ts[0].reward, ts[1].reward)) and bool(torch.equal(ts[0].step_type, ts[1].step_type))
        ints, _ = envs[0].get_task_state(); maxlen = max(maxlen, int(ints[:, 2].max()))
print(json.dumps({"workload": "flight_imitation", "envs": B, "steps": fs, "non_finite_values": bad, "episode_ends": last, "episode_starts": first,
                  "mean_reward": rsum / (B * fs), "max_episode_step_seen": maxlen, "two_handles_bit_identical": same}), flush=True)
for e in envs: e.close()
for amp in (0.2, 1.0):
    B = 4096
    envs = [fly_envs.walk_on_ball(batch_size=B) for _ in range(2)]
    g = torch.Generator(device="cuda").manual_seed(0)
    ts = [e.reset() for e in envs]
    bad = 0; last = 0; same = True; rsum = 0.0
    for k in range(bs):
        a = ((torch.rand(B, 59, device="cuda", generator=g) * 2 - 1) * amp).contiguous()
        ts = [e.step(a) for e in envs]
        o = envs[0].flat_observation
        bad += int((~torch.isfinite(o)).sum()) + int((~torch.isfinite(ts[0].reward)).sum())
        last += int((ts[0].step_type == 2).sum()); rsum += float(ts[0].reward.sum())
        if k % 50 == 0:
            same &= bool(torch.equal(o, envs[1].flat_observation)) and bool(torch.equal(ts[0].reward, ts[1].reward))
    ints, _ = envs[0].get_task_state()
    print(json.dumps({"workload": "walk_on_ball", "action_amplitude": amp, "envs": B, "steps": bs, "non_finite_values": bad, "episode_ends": last,
                      "mean_reward": rsum / (B * bs), "envs_flagged_overflow_at_end": int((ints[:, 7] != 0).sum()), "two_handles_bit_identical": same}), flush=True)
    for e in envs: e.close()
