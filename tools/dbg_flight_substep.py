"""Diagnostic: one physics substep HIP vs oracle from oracle rollout states with several active contacts."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BLOB  # noqa: E402
from flybody_amd.batched_env import BatchedFlyEnv  # noqa: E402
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories  # noqa: E402
from flybody_amd.tasks.trajectories import preprocess  # noqa: E402
from flybody_amd.tasks.wbpg import build_tables  # noqa: E402
from oracle import oracle as O  # noqa: E402

names = json.load(open(BLOB.replace(".ffmb", ".json")))["geom_name"]
tables = build_tables(base_wing_pattern())
ref = preprocess(*flight_trajectories(8, 3006))
om = O.OracleModel(BLOB)
oenv = O.OracleFlightEnv(om, tables, *ref, seed=3, env_id=0)
rng = np.random.RandomState(5)
lo = np.array([-0.2, -3, -0.5, -1, -1, -1, -1, -1, -1, -0.7, -1.05, -1.0]); hi = np.array([0.2, 3, 0.3, 1, 1, 1, 1, 1, 1, 0.7, 0.7, 1.0])
oenv.reset()
d = oenv.data
states = []
for k in range(600):
    st = oenv.step(lo + (hi - lo) * (0.5 + 0.15 * rng.uniform(-1, 1, 12)))[0]
    act = [c for c in d.contacts() if int(c[3]) == 0]
    if len(act) >= 2 and not any("wing" in names[int(c[0])] for c in act) and len(states) < 24:
        states.append((d.qpos.copy(), d.qvel.copy(), d.ctrl.copy()))
print("states", len(states))
B = len(states)
env = BatchedFlyEnv(tables, *ref, batch_size=B, seed=3)
env.reset()
env.set_state(torch.tensor(np.stack([s[0] for s in states])), torch.tensor(np.stack([s[1] for s in states])))
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
env.physics_step(torch.tensor(np.stack([s[2] for s in states]).astype(np.float32), device="cuda"), NS)
q, v = [x.cpu().numpy() for x in env.get_state()]
ints, _ = [x.cpu().numpy() for x in env.get_task_state()]
dd = O.OracleData(om)
for i, s in enumerate(states):
    dd.qpos[:], dd.qvel[:] = s[0], s[1]
    dd.ctrl[:] = s[2]
    dd.step1()
    con = [(names[int(c[0])][:-10], names[int(c[1])][:-10], f"{c[5]:.2e}", int(c[3])) for c in dd.contacts()]
    J, aref, D, ty = dd.efc()
    dd.step2()
    frc = np.ctypeslib.as_array(om.L.fo_efc_force(dd.ptr), (len(ty),)).copy()
    dd.step1()
    for _ in range(NS - 1):
        dd.step2(); dd.step1()
    ev = np.abs(v[i] - dd.qvel)
    print(i, f"qvel err {ev.max():.2e} (rel {(ev / np.maximum(1, np.abs(dd.qvel))).max():.1e}) at {ev.argsort()[-3:].tolist()} | gpu nct(after) {ints[i,7] & 255} iters {ints[i,6]} | oracle rows types {ty.tolist()} forces {np.round(frc, 3).tolist()} | contacts {con}")
