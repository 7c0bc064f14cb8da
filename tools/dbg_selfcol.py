"""Debug helper: per-env report of the fly-fly contact parity test (one substep from oracle-found contact states)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch

import test_gpu_ball as TB
from oracle import oracle as O

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m, names, states = TB._fly_fly_states(32, seed=3)
rs = np.random.RandomState(5)
ctrls = [rs.uniform(-0.5, 0.5, 59).astype(np.float32) for _ in states]
d = O.OracleData(m)
ref, info = [], []
for s_, c_ in zip(states, ctrls):
    m.set_flags(flags)
    d.qpos[:], d.qvel[:], d.act[:] = s_[:3]
    d.ctrl[:] = c_
    d.step1()
    con = d.contacts()
    d.step2()
    frc = d.efc_force[: d.nefc].copy()
    allc = [(names[int(r[0])][:14], names[int(r[1])][:22], round(r[5], 4)) for r in con]
    info.append((len(con), d.nefc, [(names[int(r[0])][:18], names[int(r[1])][:18], round(r[5], 5), int(r[3])) for r in con if "ball" not in names[int(r[0])]],
                 np.round(frc[np.argsort(-np.abs(frc))[:4]], 3), allc, d.efc()[3].copy()))
    d.step1()
    ref.append((d.qpos.copy(), d.qvel.copy(), d.act.copy()))
    m.set_flags(0)
q, v, a, ints = TB._gpu_advance(torch, [s_[:3] for s_ in states], ctrls, 1, flags)
for i, r in enumerate(ref):
    ev = np.abs(v[i] - r[1]).max() / max(1.0, np.abs(r[1]).max())
    j = int(np.abs(v[i] - r[1]).argmax())
    print(f"{i:2d} qvel rel err {ev:9.2e} (dof {j}: gpu {v[i][j]:.4g} ref {r[1][j]:.4g}) | oracle ncon {info[i][0]} nefc {info[i][1]} | gpu ncon {ints[i, 5]} nself {ints[i, 3]} iters {ints[i, 6]} ovf {ints[i, 7]} | {info[i][2]} f {info[i][3]}")

import json
jn = json.load(open("flybody_amd/assets/fly_ball.json"))["jnt_name"]
for i, r in enumerate(ref):
    ev = np.abs(v[i] - r[1]).max() / max(1.0, np.abs(r[1]).max())
    if ev > 1e-3:
        e = np.abs(v[i] - r[1])
        top = np.argsort(-e)[:8]
        print("env", i, "worst dofs", [(int(j), jn[j - 2] if j >= 3 else "ball", round(float(v[i][j]), 3), round(float(r[1][j]), 3)) for j in top])
        print("   contacts", info[i][4], "efc types", info[i][5].tolist())
