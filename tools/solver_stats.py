"""How often the joint-limit active-set iteration needs more than one pass (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from flybody_amd import fly_envs
B = 8192
env = fly_envs.flight_imitation(batch_size=B, random_state=0)
spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
env.reset()
hist = np.zeros(40, dtype=np.int64); nact = []
for k in range(60):
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    ts = env.step(a)
    ints, _ = env.get_task_state()
    it = ints[:, 6].cpu().numpy(); na = ints[:, 5].cpu().numpy()
    st = ts.step_type.cpu().numpy()
    it = it[st != 0]
    hist += np.bincount(np.clip(it, 0, 39), minlength=40)
    nact.append(na.mean())
tot = hist.sum()
print("Newton passes per control step (4 substeps): distribution", {i: round(h / tot, 3) for i, h in enumerate(hist) if h})
print("mean passes per control step", (hist * np.arange(40)).sum() / tot, " mean active limits at step end", np.mean(nact))
