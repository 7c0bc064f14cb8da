"""Diagnostic: teacher-forced control steps of walk_on_ball, HIP vs oracle, printing per-group errors and the oracle's contacts of the
env-steps that exceed a tolerance (development aid for the convex contact path)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flybody_amd import fly_envs  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_ball import BALL_BLOB, _gpu_contact_history, _obs_groups  # noqa: E402

names = json.load(open(BALL_BLOB.replace(".ffmb", ".json")))["geom_name"]
B = 8
env = fly_envs.walk_on_ball(batch_size=B)
m = O.OracleModel(BALL_BLOB)
oenvs = [O.OracleBallEnv(m) for _ in range(B)]
env.reset()
[e.reset() for e in oenvs]
groups = _obs_groups()
rs = np.random.RandomState(0)
for t in range(30):
    q, v = env.get_state()
    ac = env.get_act()
    q, v, ac = q.cpu().numpy(), v.cpu().numpy(), ac.cpu().numpy()
    a = rs.uniform(-0.2, 0.2, (B, 59)) * (1.0 + 0.1 * t)
    ts = env.step(torch.tensor(a, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    obs = env.flat_observation.cpu().numpy()
    ghist = _gpu_contact_history(env)
    ints, _ = env.get_task_state()
    ints = ints.cpu().numpy()
    for i, e in enumerate(oenvs):
        d = e.data
        d.qpos[:], d.qvel[:], d.act[:] = q[i], v[i], ac[i]
        d.step1()
        st, r, dsc, o = e.step(a[i].astype(np.float32).astype(np.float64))
        ohist, ogap = e.contact_history()
        errs = {n: np.abs(obs[i, lo:hi] - o[lo:hi]).max() / max(1.0, np.abs(o[lo:hi]).max()) for n, (lo, hi) in groups.items()}
        if errs["appendages_pos"] > 1.5e-6 or errs["joints_pos"] > 2.5e-6:
            lo, hi = groups["appendages_pos"]
            print(f"t {t} env {i} flip {(ohist != ghist[i]).any()} ohist {ohist.tolist()} ghist {ghist[i].tolist()} gpu ncon {ints[i,5]} nself {ints[i,3]} ovf {ints[i,7]}")
            print("   errs", {k: f"{v:.1e}" for k, v in errs.items() if v > 1e-6})
            print("   appendage err per site", np.round(np.abs(obs[i, lo:hi] - o[lo:hi]).reshape(7, 3).max(1), 7))
            lo, hi = groups["joints_pos"]
            je = np.abs(obs[i, lo:hi] - o[lo:hi])
            print("   joints_pos worst idx", je.argsort()[-4:], je[je.argsort()[-4:]])
            np.savez(os.path.join(ROOT, "gpurun_out", f"dbg_state_t{t}_e{i}.npz"), q=q[i], v=v[i], act=ac[i], a=a[i], obs=obs[i], q2=env.get_state()[0].cpu().numpy()[i], v2=env.get_state()[1].cpu().numpy()[i])
            print("   oracle contacts:", [(names[int(c[0])], names[int(c[1])], f"{c[5]:.2e}", int(c[3])) for c in d.contacts() if int(c[0]) != 0])
