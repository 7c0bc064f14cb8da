"""Debug helper: teacher-forced substep-by-substep comparison for one env of the open-loop parity run."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from flybody_amd.batched_env import BatchedBallEnv
from oracle import oracle as O

which, upto = int(sys.argv[1]), int(sys.argv[2])
m = O.OracleModel("flybody_amd/assets/fly_ball.ffmb")
B = 8
oenvs = [O.OracleBallEnv(m) for _ in range(B)]
[e.reset() for e in oenvs]
rs = np.random.RandomState(0)
for t in range(upto):
    a = rs.uniform(-0.2, 0.2, (B, 59))
    for i, e in enumerate(oenvs):
        e.step(a[i].astype(np.float32).astype(np.float64))
a = rs.uniform(-0.2, 0.2, (B, 59))[which].astype(np.float32)
e = oenvs[which]
d = e.data
state = (d.qpos.copy(), d.qvel.copy(), d.act.copy())
# ctrl from action (raw spec): act_action mapping via the blob
from flybody_amd.model.blob import read_blob
tb = read_blob("flybody_amd/assets/fly_ball.ffmb")
ctrl = np.array([a[k] if k >= 0 else 0.0 for k in tb["act_action"]], dtype=np.float32)
env = BatchedBallEnv(batch_size=10)
env.reset()
for flags in (0,):
    od = O.OracleData(m)
    od.qpos[:], od.qvel[:], od.act[:] = state
    od.ctrl[:] = ctrl
    od.step1()
    refs = []
    for k in range(10):
        od.step2()
        c = od.contacts()
        info = (od.ncon, od.nefc, [int(x) for x in c[:, 3]], np.round(c[:, 5], 6).tolist(), np.round(c[:, 15], 4).tolist())
        od.step1()
        refs.append((od.qpos.copy(), od.qvel.copy(), info))
    # GPU: env k advances k+1 substeps
    qpos = torch.tensor(np.tile(state[0], (10, 1)), dtype=torch.float64, device="cuda")
    qvel = torch.tensor(np.tile(state[1], (10, 1)), dtype=torch.float64, device="cuda")
    act = torch.tensor(np.tile(state[2], (10, 1)), dtype=torch.float64, device="cuda")
    for k in range(10):
        env.set_state(qpos, qvel)
        env.set_act(act)
        env.physics_step(torch.tensor(np.tile(ctrl, (10, 1)), dtype=torch.float32, device="cuda"), k + 1)
        q, v = env.get_state()
        q, v = q.cpu().numpy()[0], v.cpu().numpy()[0]
        print("substeps", k + 1, "qpos err %.2e qvel err %.2e (idx %d)" % (np.abs(q - refs[k][0]).max(), np.abs(v - refs[k][1]).max(), int(np.abs(v - refs[k][1]).argmax())),
              "oracle ncon/nefc/excl/dist/fn", refs[k][2])
