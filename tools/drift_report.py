"""SURVEY.md section 8(d) drift report: open-loop drift of qpos / qvel of the HIP physics step against the float64 oracle, with no
resynchronisation at all: 64 envs x 1000 control steps (4 substeps each) of `ffe_physics_step` with limits, fluid and actuation on and a
wing-beat-like control signal.  Two runs: the fly's own contacts off (pure float32 drift of a smooth system) and on (contact events
amplify the drift: a pair that makes contact a substep earlier on one side is a discontinuity).

    python tools/drift_report.py            # prints the curve; profiles/r03_drift_report.log is this output
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from flybody_amd.batched_env import BatchedFlyEnv  # noqa: E402
from flybody_amd.model.blob import read_blob  # noqa: E402
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories  # noqa: E402
from flybody_amd.tasks.trajectories import preprocess  # noqa: E402
from flybody_amd.tasks.wbpg import build_tables  # noqa: E402
from oracle import oracle as O  # noqa: E402

BLOB = os.path.join(ROOT, "flybody_amd", "assets", "fly_flight.ffmb")


def run(flags, B=64, nctrl=1000, checkpoints=(10, 100, 1000), quiet=False):
    tables = build_tables(base_wing_pattern())
    ref = preprocess(*flight_trajectories(8, 3006))
    blob = read_blob(BLOB)
    env = BatchedFlyEnv(tables, *ref, batch_size=B, seed=0, physics_flags=flags)
    nq, nv, nu = env.spec.nq, env.spec.nv, env.spec.nu
    rng = np.random.RandomState(4)
    th = np.deg2rad(47.5)
    qpos = np.tile(blob["qpos0"], (B, 1))
    qpos[:, :3] = [0.0, 0.0, 1.0]
    qpos[:, 3:7] = [np.cos(th / 2), 0, -np.sin(th / 2), 0]
    qvel = np.zeros((B, nv))
    qvel[:, 0] = 30.0
    env.set_state(torch.tensor(qpos), torch.tensor(qvel))
    om = O.OracleModel(BLOB)
    om.set_flags(flags)
    datas = []
    for i in range(B):
        d = O.OracleData(om)
        d.qpos[:], d.qvel[:] = qpos[i], qvel[i]
        d.step1()
        datas.append(d)
    # wing-beat-like controls: the six wing torques swing at 200 Hz with per-env phase and amplitude, the head / abdomen servos hold
    # small random set points (ctrl is fixed over a control step of 4 substeps, as the task applies it)
    lo, hi = np.asarray(blob["act_ctrlrange"]).reshape(-1, 2).T
    phase, amp = rng.uniform(0, 2 * np.pi, (B, 1)), rng.uniform(0.2, 0.6, (B, 1))
    base = rng.uniform(-0.3, 0.3, (B, nu)) * np.minimum(-lo, hi)
    wing = np.array([int(blob["act_trnid"][u]) in set(int(j) for j in blob["wing_jnt"]) and int(blob["act_trntype"][u]) == 0 for u in range(nu)])
    out = {}
    for k in range(1, nctrl + 1):
        t = k * 2e-4
        ctrl = base.copy()
        ctrl[:, wing] = amp * np.sin(2 * np.pi * 200.0 * t + phase + np.arange(wing.sum())[None, :] * 0.7)
        ctrl = np.clip(ctrl, lo, hi).astype(np.float32)
        env.physics_step(torch.tensor(ctrl, device="cuda"), 4)
        for i, d in enumerate(datas):
            d.ctrl[:] = ctrl[i]
            for _ in range(4):
                d.step2(); d.step1()
        if k in checkpoints:
            q, v = [x.cpu().numpy() for x in env.get_state()]
            oq, ov = np.stack([d.qpos for d in datas]), np.stack([d.qvel for d in datas])
            eq = np.abs(q - oq).max(axis=1)
            ev = (np.abs(v - ov) / np.maximum(1.0, np.abs(ov).max(axis=1, keepdims=True))).max(axis=1)
            out[k] = (float(np.median(eq)), float(eq.max()), float(np.median(ev)), float(ev.max()))
            if not quiet:
                print(f"  after {k:5d} control steps ({4 * k} substeps): |dqpos| median {out[k][0]:.2e} max {out[k][1]:.2e}   rel |dqvel| median {out[k][2]:.2e} max {out[k][3]:.2e}")
    env.close()
    return out


if __name__ == "__main__":
    print("open-loop drift, HIP float32 vs oracle float64, 64 envs, no resynchronisation (limits, fluid, actuation on)")
    print(" fly-fly contacts off (FFE_NO_CONTACT):")
    run(64)
    print(" fly-fly contacts on:")
    run(0)
