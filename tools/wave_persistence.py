"""Flight kernel, -DFFE_TRACE build with the launch order switched off (slot = env): how well does a step's work predict the next step's
wave lifetime?   python tools/wave_persistence.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["FLYBODY_ENV_LIB"] = os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "csrc", "variants", "libflybody_env_trace.so")
import numpy as np, torch
from flybody_amd import _capi
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

B = 8192
L = _capi.lib()
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0, physics_flags=1 << 24)   # DBG_NO_ORDER: identity launch order
spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
env.reset()
L.ffe_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
rec = []
for k in range(90):
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    ts = env.step(a)
    if k >= 30:
        torch.cuda.synchronize()
        buf = np.zeros((B, 4), dtype=np.uint64)
        assert L.ffe_debug_read_trace(buf.ctypes.data, B) == 0
        st = ts.step_type.cpu().numpy()
        life = (buf[:, 1].astype(np.int64) - buf[:, 0].astype(np.int64)) * 0.01
        ex = buf[:, 3] >> np.uint64(8); hi_ = buf[:, 2] >> np.uint64(32)
        iters = (ex & np.uint64(0xff)).astype(np.float64); hist = ((ex >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
        csum = sum(((hist >> (4 * q)) & 15) for q in range(4)).astype(np.float64)
        nb = (hi_ & np.uint64(0xff)).astype(np.float64); tcoll = ((hi_ >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64) * 0.01
        rec.append((life, iters, csum, nb, tcoll, st))
life = np.stack([r[0] for r in rec]); iters = np.stack([r[1] for r in rec]); csum = np.stack([r[2] for r in rec]); nb = np.stack([r[3] for r in rec]); tcoll = np.stack([r[4] for r in rec])
st = np.stack([r[5] for r in rec])
ok = (st[:-1] != 2) & (st[1:] != 0) & (st[1:] != 2)  # pairs of consecutive steps inside an episode
x0, x1 = life[:-1][ok], life[1:][ok]
print(f"{ok.sum()} consecutive env-step pairs; lifetime autocorrelation {np.corrcoef(x0, x1)[0,1]:.2f}")
b0, b1 = nb[:-1][ok] > 0, nb[1:][ok] > 0
print(f"P(second-pass call next step | one this step) = {b1[b0].mean():.2f}; P(next | none this step) = {b1[~b0].mean():.3f}; base rate {b1.mean():.3f}")
work = iters + 0.75 * csum + 4 * nb
for name, key in (("passes", iters), ("contacts", csum), ("second-pass calls", nb), ("collision time", tcoll), ("work = passes + 0.75 contacts + 4 calls", work), ("lifetime itself", life)):
    print(f"   corr(next lifetime, this step's {name}) = {np.corrcoef(key[:-1][ok], x1)[0,1]:.2f}")
# the top 1 % longest next-step waves: what rank did this step's work give them?
thr = np.percentile(x1, 99)
w0 = work[:-1][ok]
rank = (w0[:, None] if False else None)
order = np.argsort(-w0); pos = np.empty(len(w0)); pos[order] = np.arange(len(w0)) / len(w0)
print(f"waves in the top 1 % of next-step lifetime (> {thr:.0f} us): median launch position by this step's work {np.median(pos[x1 > thr]):.2f} (0 = first, 0.5 = random)")
pos2 = np.empty(len(w0)); o2 = np.argsort(-tcoll[:-1][ok]); pos2[o2] = np.arange(len(w0)) / len(w0)
print(f"   ... by this step's collision time: {np.median(pos2[x1 > thr]):.2f}")
