"""Diagnostic: substep-by-substep comparison HIP vs oracle from a dumped walk_on_ball state (gpurun_out/dbg_state_*.npz)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flybody_amd.batched_env import BatchedBallEnv  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_ball import BALL_BLOB  # noqa: E402

S = np.load(sys.argv[1])
m = O.OracleModel(BALL_BLOB)
d = O.OracleData(m)
from flybody_amd.model.blob import read_blob  # noqa: E402
aa = np.asarray(read_blob(BALL_BLOB)["act_action"])
ctrl = np.array([S["a"][k] if k >= 0 else 0.0 for k in aa]).astype(np.float32)  # ref: fruitfly.py:480-492 apply_action
d.qpos[:], d.qvel[:], d.act[:] = S["q"], S["v"], S["act"]
d.ctrl[:] = ctrl.astype(np.float64)
d.step1()
env = BatchedBallEnv(batch_size=1)
env.reset()
env.set_state(torch.tensor(S["q"][None], dtype=torch.float64, device="cuda"), torch.tensor(S["v"][None], dtype=torch.float64, device="cuda"))
env.set_act(torch.tensor(S["act"][None], dtype=torch.float64, device="cuda"))
import json
names = json.load(open(BALL_BLOB.replace(".ffmb", ".json")))["geom_name"]
for sub in range(10):
    if sub >= 8:
        # teacher-forced single substep from the oracle's state, with and without limits / contacts
        for fl in (0, 2, 64):
            e2 = BatchedBallEnv(batch_size=1, physics_flags=fl)
            e2.reset()
            e2.set_state(torch.tensor(d.qpos[None].copy(), dtype=torch.float64, device="cuda"), torch.tensor(d.qvel[None].copy(), dtype=torch.float64, device="cuda"))
            e2.set_act(torch.tensor(d.act[None].copy(), dtype=torch.float64, device="cuda"))
            e2.physics_step(torch.tensor(ctrl[None], device="cuda"), 1)
            v2 = e2.get_state()[1].cpu().numpy()[0]
            d2 = O.OracleData(m)
            m.set_flags(fl)
            d2.qpos[:], d2.qvel[:], d2.act[:] = d.qpos, d.qvel, d.act
            d2.ctrl[:] = d.ctrl
            d2.step1(); d2.step2()
            m.set_flags(0)
            ev2 = np.abs(v2 - d2.qvel)
            print(f"   teacher-forced sub {sub} flags {fl}: qvel err {ev2.max():.2e} at {ev2.argsort()[-3:].tolist()}; oracle rows {d2.nefc} ncon {d2.ncon}")
            e2.close()
        print("   oracle contacts:", [(names[int(c[0])], names[int(c[1])], f"{c[5]:.3e}", int(c[3]), f"f={c[15]:.2e}") for c in d.contacts()])
        J, aref, Dd, ty = d.efc()
        print("   efc types", ty.tolist(), "rows: dofs", [np.nonzero(J[r])[0].tolist() for r in range(len(ty))])
    d.step2()
    nc, nefc, it = d.ncon, d.nefc, m.L.fo_solver_iter(d.ptr)
    d.step1()
    env.physics_step(torch.tensor(ctrl[None], device="cuda"), 1)
    q, v = env.get_state()
    ints, _ = env.get_task_state()
    q, v, ints = q.cpu().numpy()[0], v.cpu().numpy()[0], ints.cpu().numpy()[0]
    ev = np.abs(v - d.qvel)
    k = ev.argsort()[-3:]
    print(f"sub {sub}: oracle ncon {nc} nefc {nefc} iters {it} | gpu ncon {ints[5]} nself {ints[3]} iters {ints[6]} ovf {ints[7]} | qvel err max {ev.max():.2e} (rel {ev.max()/max(1,np.abs(d.qvel).max()):.1e}) at dofs {k.tolist()} qpos err {np.abs(q-d.qpos).max():.1e}")
