"""Flight kernel, -DFFE_TRACE build: least-squares fit of a wave's lifetime on what its step did (solver passes, contacts per substep),
and how well the launch-order key predicted it.   python tools/wave_cost_fit.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ.setdefault("FLYBODY_ENV_LIB", os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "csrc", "variants", "libflybody_env_trace.so"))
import numpy as np, torch
from flybody_amd import _capi
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

B = 8192
L = _capi.lib()
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0)
spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
env.reset()
L.ffe_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
rows = []
for k in range(60):
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    env.step(a)
    if k >= 40 and k % 4 == 0:
        torch.cuda.synchronize()
        buf = np.zeros((B, 4), dtype=np.uint64)
        assert L.ffe_debug_read_trace(buf.ctypes.data, B) == 0
        rows.append(buf)
buf = np.concatenate(rows)
t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64)
life = (t1 - t0) * 0.01  # us
ex = buf[:, 3].astype(np.uint64) >> np.uint64(8)
iters = (ex & np.uint64(0xff)).astype(np.float64); act = ((ex >> np.uint64(8)) & np.uint64(0xff)).astype(np.float64)
key = ((ex >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64); hist = ((ex >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
ncs = np.stack([(hist >> (4 * q)) & 15 for q in range(4)], 1)
ntot = ncs.sum(1).astype(np.float64); nmax = ncs.max(1).astype(np.float64)
hi = buf[:, 2].astype(np.uint64) >> np.uint64(32)
nb = (hi & np.uint64(0xff)).astype(np.float64); nfac = ((hi >> np.uint64(8)) & np.uint64(0xff)).astype(np.float64); tcoll = ((hi >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64) * 0.01
print(f"collision stage per wave-step: mean {tcoll.mean():.1f} us ({100 * tcoll.sum() / life.sum():.1f} % of wave time), p50 {np.percentile(tcoll, 50):.1f}, p99 {np.percentile(tcoll, 99):.1f}, max {tcoll.max():.1f}; second-pass calls per step: mean {nb.mean():.2f} (share of steps with any {np.mean(nb > 0):.2f}); factorisations per step mean {nfac.mean():.2f} max {nfac.max():.0f}")
top = np.argsort(-life)[:25]
print("longest waves: life us | collision us | passes | factorisations | second-pass calls | contacts per substep | active limits | key")
for i in top:
    print(f"  {life[i]:7.1f} | {tcoll[i]:6.1f} | {iters[i]:3.0f} | {nfac[i]:3.0f} | {nb[i]:2.0f} | {ncs[i].tolist()} | {act[i]:2.0f} | {key[i]:5.0f}")
X = np.stack([np.ones(len(life)), iters, nfac, ntot, nb], 1)
coef = np.linalg.lstsq(X, life - tcoll, rcond=None)[0]
print(f"outside the collision stage: {coef[0]:.1f} + {coef[1]:.2f} x passes + {coef[2]:.2f} x factorisations + {coef[3]:.2f} x contacts + {coef[4]:.2f} x second-pass calls; residual sd {np.std(life - tcoll - X @ coef):.1f}")
X = np.stack([np.ones(len(life)), ntot, nb], 1)
coef = np.linalg.lstsq(X, tcoll, rcond=None)[0]
print(f"collision stage: {coef[0]:.1f} + {coef[1]:.2f} x contacts + {coef[2]:.2f} x second-pass calls; residual sd {np.std(tcoll - X @ coef):.1f}")
# only waves of the first round (started before any wave ended) have comparable contention
first = t0 < np.percentile(t1, 1)
print(f"{len(life)} waves; lifetime us mean {life.mean():.1f} p10 {np.percentile(life,10):.1f} p50 {np.percentile(life,50):.1f} p90 {np.percentile(life,90):.1f} p99 {np.percentile(life,99):.1f} max {life.max():.1f}")
for name, sel in (("all", np.ones(len(life), bool)), ("first round", first), ("later", ~first)):
    X = np.stack([np.ones(sel.sum()), iters[sel], ntot[sel], (nmax[sel] >= 3).astype(float)], 1)
    coef, res, *_ = np.linalg.lstsq(X, life[sel], rcond=None)
    pred = X @ coef
    print(f"{name}: life = {coef[0]:.1f} + {coef[1]:.2f} x passes + {coef[2]:.2f} x contacts(sum over substeps) + {coef[3]:.1f} x [3+ contacts at once]; residual sd {np.std(life[sel] - pred):.1f} us (sd of lifetime {np.std(life[sel]):.1f})")
    print(f"   correlation of lifetime with: passes {np.corrcoef(iters[sel], life[sel])[0,1]:.2f}, contacts {np.corrcoef(ntot[sel], life[sel])[0,1]:.2f}, launch key {np.corrcoef(key[sel], life[sel])[0,1]:.2f}")
print("mean passes", iters.mean(), "mean contacts per substep", ntot.mean() / 4)
k1 = key[-B:]
print("launch-order key by workgroup index (every 512th):", k1[::512].astype(int).tolist(), "| histogram of keys:", np.bincount(k1.astype(int) // 8)[:32].tolist())
l1 = life[-B:]
print("mean lifetime by workgroup index (blocks of 1024):", [round(float(l1[i:i + 1024].mean()), 1) for i in range(0, B, 1024)])
print("second-pass calls by workgroup index (blocks of 1024):", [round(float(nb[-B:][i:i + 1024].mean()), 2) for i in range(0, B, 1024)])
