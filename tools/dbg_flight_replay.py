"""Diagnostic: replay one env-step of tools/dbg_flight_contacts.py substep by substep (HIP physics_step vs oracle)."""
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

ENV, STEP = int(sys.argv[1]), int(sys.argv[2])
SEED, SCALE, BB = (int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (11, 0.3, 16)
ESEED = int(sys.argv[6]) if len(sys.argv) > 6 else 3
names = json.load(open(BLOB.replace(".ffmb", ".json")))["geom_name"]
tables = build_tables(base_wing_pattern())
ref = preprocess(*flight_trajectories(8, 3006))
B = BB
env = BatchedFlyEnv(tables, *ref, batch_size=1, seed=ESEED)
amin, amax = env.action_spec().minimum, env.action_spec().maximum
om = O.OracleModel(BLOB)
oe = O.OracleFlightEnv(om, tables, *ref, ghost_accel_z=env.ghost_accel_z, seed=ESEED, env_id=ENV)
rng = np.random.RandomState(SEED)
oe.reset()
d = oe.data
for k in range(STEP + 1):
    a = (amin + (amax - amin) * (0.5 + 0.5 * SCALE * rng.uniform(-1, 1, (B, len(amin))))).astype(np.float32)
    if k == STEP:
        q0, v0 = d.qpos.copy(), d.qvel.copy()
    oe.step(a[ENV].astype(np.float64))
ctrl = d.ctrl.copy()
dd = O.OracleData(om)
dd.qpos[:], dd.qvel[:] = q0, v0
dd.ctrl[:] = ctrl
dd.step1()
env.reset()
env.set_state(torch.tensor(q0[None]), torch.tensor(v0[None]))
for sub in range(4):
    con = [(names[int(c[0])][:-10], names[int(c[1])][:-10], f"{c[5]:.3e}", int(c[3])) for c in dd.contacts()]
    dd.step2()
    J, aref, D, ty = dd.efc()
    frc = np.ctypeslib.as_array(om.L.fo_efc_force(dd.ptr), (len(ty),)).copy()
    it = om.L.fo_solver_iter(dd.ptr)
    dd.step1()
    env.physics_step(torch.tensor(ctrl[None].astype(np.float32), device="cuda"), 1)
    q, v = [x.cpu().numpy()[0] for x in env.get_state()]
    ints = env.get_task_state()[0].cpu().numpy()[0]
    ev = np.abs(v - dd.qvel)
    print("   deep ratio", dd.deep_ratio())
    print(f"sub {sub}: qvel err {ev.max():.2e} at {ev.argsort()[-3:].tolist()} | gpu nct(after) {ints[7] & 255} iters {ints[6]} | oracle contacts(before) {con} rows {ty.tolist()} forces {np.round(frc, 2).tolist()} iters {it}")
