"""Diagnostic: open-loop full-range flight rollout (deep wing strikes resynchronised as in the tests); prints the env-steps whose
reward differs from the oracle's by more than 1e-4 with the oracle's contacts and deepest convex overlap."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BLOB  # noqa: E402
from flybody_amd import fly_envs  # noqa: E402
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories  # noqa: E402
from flybody_amd.tasks.trajectories import preprocess  # noqa: E402
from flybody_amd.tasks.wbpg import build_tables  # noqa: E402
from oracle import oracle as O  # noqa: E402

names = json.load(open(BLOB.replace(".ffmb", ".json")))["geom_name"]
B, STEPS, DEEP = 256, 500, float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
env = fly_envs.flight_imitation(batch_size=B, random_state=0)
rq, rv = preprocess(*flight_trajectories())
om = O.OracleModel(BLOB)
tables = build_tables(base_wing_pattern())
oenvs = [O.OracleFlightEnv(om, tables, rq, rv, ghost_accel_z=env.ghost_accel_z, seed=0, env_id=i) for i in range(B)]
lo = torch.tensor(env.action_spec().minimum, device="cuda"); hi = torch.tensor(env.action_spec().maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(77)
env.reset(); [e.reset() for e in oenvs]
alive = np.ones(B, bool); deep_prev = np.zeros(B); nbad = deep = comp = nflip = nflip_bad = nover = 0; flipgaps = []
hist = []
for k in range(STEPS):
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    ts = env.step(a)
    a_s = a.cpu().numpy().astype(np.float64)
    rew, st = ts.reward.cpu().numpy(), ts.step_type.cpu().numpy()
    ints = env.get_task_state()[0].cpu().numpy()
    resync = []
    for j in range(B):
        if not alive[j]:
            continue
        oenvs[j].data.contact_hist()
        ost, orr, od, _ = oenvs[j].step(a_s[j])
        ohist, ogap = oenvs[j].data.contact_hist()
        ghist = [(int(ints[j, 7]) >> (16 + 4 * q)) & 15 for q in range(4)]
        flip = ost == 1 and list(ohist[:4]) != ghist
        ratio = oenvs[j].data.deep_ratio()
        dr, deep_prev[j] = max(ratio, deep_prev[j]), ratio
        if ost != st[j]:
            alive[j] = False
            continue
        if dr > DEEP:
            deep += 1; resync.append(j); continue
        if (int(ints[j, 7]) >> 8) & 255:
            nover += 1; resync.append(j); continue
        comp += 1
        e = abs(float(rew[j]) - orr)
        if flip:
            nflip += 1; flipgaps.append(ogap); resync.append(j)
            if e > 1e-4: nflip_bad += 1
            continue
        hist.append((e, dr))
        if e > 1e-4 and nbad < 12:
            nbad += 1
            con = [(names[int(c[0])][:-10], names[int(c[1])][:-10], f"{c[5]:.2e}") for c in oenvs[j].data.contacts() if int(c[3]) == 0]
            print(f"step {k} env {j}: ohist {list(ohist[:4])} ghist {ghist} gap {ogap:.1e} reward err {e:.2e} deep ratio (this/prev step) {dr:.2f} gpu nct {ints[j,7] & 255} ovf {ints[j,7] >> 8} | oracle contacts now {con}")
            resync.append(j)
    if resync:
        qpos, qvel = env.get_state()
        rows = torch.tensor(resync, device=qpos.device)
        qpos[rows] = torch.tensor(np.stack([oenvs[j].data.qpos for j in resync]), dtype=qpos.dtype, device=qpos.device)
        qvel[rows] = torch.tensor(np.stack([oenvs[j].data.qvel for j in resync]), dtype=qvel.dtype, device=qvel.device)
        env.set_state(qpos, qvel)
h = np.array(hist)
print(f"contact flips {nflip} (of them reward err > 1e-4: {nflip_bad}); oracle switching gaps on flips: max {max(flipgaps, default=0):.2e} median {np.median(flipgaps) if flipgaps else 0:.2e}")
print(f"over {nover}")
print(f"compared {comp}, deep {deep}, dropped {int((~alive).sum())}; reward err > 1e-4: {(h[:,0] > 1e-4).sum()}, > 1e-5: {(h[:,0] > 1e-5).sum()}, max {h[:,0].max():.2e}")
for lo_, hi_ in ((0, 0.1), (0.1, 0.2), (0.2, 0.3), (0.3, 0.4), (0.4, 0.5), (0.5, 1.0)):
    m = (h[:, 1] >= lo_) & (h[:, 1] < hi_)
    if m.any():
        print(f"  deep ratio [{lo_}, {hi_}): {m.sum()} env-steps, max reward err {h[m, 0].max():.2e}, > 1e-5: {(h[m, 0] > 1e-5).sum()}")
