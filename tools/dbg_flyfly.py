"""Diagnostic: per-state errors of the fly-fly one-substep comparison (tests/test_gpu_ball.py::test_fly_fly_contacts_one_substep)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ball as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

m, names, states = T._fly_fly_states(32, seed=3)
rs = np.random.RandomState(5)
ctrls = [rs.uniform(-0.5, 0.5, 59).astype(np.float32) for _ in states]
d = O.OracleData(m)
ref = []
for s_, c_ in zip(states, ctrls):
    d.qpos[:], d.qvel[:], d.act[:] = s_[:3]
    d.ctrl[:] = c_
    d.step1()
    con = d.contacts()
    J, aref, D, ty = d.efc()
    d.step2()
    d.step1()
    ref.append((d.qpos.copy(), d.qvel.copy(), len(con), len(ty), [(names[int(r[0])][:-10], names[int(r[1])][:-10], round(r[5], 5)) for r in con if int(r[0]) != 0]))
q, v, a, ints = T._gpu_advance(torch, [s_[:3] for s_ in states], ctrls, 1, 0)
for i, r in enumerate(ref):
    ev = np.abs(v[i] - r[1])
    print(i, f"qvel err {ev.max():.2e} rel {ev.max()/max(1,np.abs(r[1]).max()):.1e} at {ev.argsort()[-3:].tolist()} ovf {ints[i,7]} oracle ncon {r[2]} rows {r[3]} | {r[4]}")
