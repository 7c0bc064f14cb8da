"""Debug helper: open-loop parity of the HIP walk_on_ball env against the oracle, step by step.
    python tools/dbg_ball_parity.py [flags] [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from flybody_amd.batched_env import BatchedBallEnv
from oracle import oracle as O

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = 8
env = BatchedBallEnv(batch_size=B, physics_flags=flags)
m = O.OracleModel("flybody_amd/assets/fly_ball.ffmb")
m.set_flags(flags)
oenvs = [O.OracleBallEnv(m) for _ in range(B)]
env.reset()
[e.reset() for e in oenvs]
rs = np.random.RandomState(0)
for t in range(steps):
    a = rs.uniform(-0.2, 0.2, (B, 59))
    ts = env.step(torch.tensor(a, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    q, v = env.get_state()
    q, v = q.cpu().numpy(), v.cpu().numpy()
    act = env.get_act().cpu().numpy()
    rew = ts.reward.cpu().numpy()
    ints, _ = env.get_task_state()
    ints = ints.cpu().numpy()
    worst = (0, 0, 0)
    for i, e in enumerate(oenvs):
        st, r, dsc, o = e.step(a[i].astype(np.float32).astype(np.float64))
        dq = np.abs(q[i] - e.data.qpos)
        if dq.max() > worst[0]:
            worst = (dq.max(), i, int(dq.argmax()))
        ea = np.abs(act[i] - e.data.act).max()
    print(t, "worst qpos err %.2e env %d idx %d" % worst, "act err %.1e" % ea, "rew err %.2e" % np.abs(rew - [0] * B).min(),
          "iters", ints[:, 6].tolist())
