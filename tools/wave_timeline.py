"""Occupancy timeline of one launch of a step kernel from the -DFFE_TRACE diagnostic build (bash tools/build_variants.sh):
every wave's start / end time (100 MHz reference clock ticks) and hardware slot.  Prints the dispatch ramp, the wave-lifetime distribution, the number
of resident waves over time and per-CU slot use.   python tools/wave_timeline.py [flight|ball] [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["FLYBODY_ENV_LIB"] = os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "csrc", "variants", "libflybody_env_trace.so")
import numpy as np, torch
from flybody_amd import _capi, fly_envs
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

which = sys.argv[1] if len(sys.argv) > 1 else "flight"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (8192 if which == "flight" else 4096)
L = _capi.lib()
g = torch.Generator(device="cuda").manual_seed(0)
if which == "flight":
    tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))
    env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0)
    spec = env.action_spec(); lo = torch.tensor(spec.minimum, device="cuda"); hi = torch.tensor(spec.maximum, device="cuda")
    a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
    warm, read = 30, L.ffe_debug_read_trace
else:
    env = fly_envs.walk_on_ball(batch_size=B)
    a = ((torch.rand(B, 59, device="cuda", generator=g) * 2 - 1) * 0.2).contiguous()
    warm, read = 330, L.ffb_debug_read_trace
read.argtypes = [C.c_void_p, C.c_int]
env.reset()
for _ in range(warm): env.step(a)
torch.cuda.synchronize()
buf = np.zeros((B, 4), dtype=np.uint64)
assert read(buf.ctypes.data, B) == 0
if len(sys.argv) > 3:
    np.save(sys.argv[3], buf)  # raw rows (start, end, HW_ID, XCC_ID), one per workgroup in launch order
t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64); hw = buf[:, 2].astype(np.int64); xcc = buf[:, 3].astype(np.int64) & 0xf
b = t0.min(); t0 -= b; t1 -= b  # s_memrealtime: the 100 MHz reference clock, common to all XCDs (10 ns ticks)
life = t1 - t0; span = t1.max()
print(f"{which} B={B}: launch span {span} ticks (longest XCD; per XCD: {[int(t1[xcc == x].max()) for x in np.unique(xcc)]})")
print(f"wave lifetime mean {life.mean():.0f}  p10 {np.percentile(life, 10):.0f}  p50 {np.percentile(life, 50):.0f}  p90 {np.percentile(life, 90):.0f}  "
      f"max {life.max()};  sum(lifetimes) / span = {life.sum() / span:.0f} resident waves on average")
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3; slot = hw & 0xf
key = (xcc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd
nsimd = len(np.unique(key))
print(f"distinct SIMDs used: {nsimd}; distinct CUs: {len(np.unique(key >> 2))}; wave slots seen per SIMD: {sorted(set(int(x) for x in slot))}")
per_cu = np.bincount(np.unique(key >> 2, return_inverse=True)[1])
print(f"waves per CU over the launch: min {per_cu.min()} max {per_cu.max()}")
edges = np.linspace(0, span, 41)
for lo_, hi_ in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo_ + hi_)
    res = int(((t0 <= mid) & (t1 > mid)).sum())
    started = int(((t0 >= lo_) & (t0 < hi_)).sum()); ended = int(((t1 >= lo_) & (t1 < hi_)).sum())
    print(f"  t = {mid:9.0f} ticks ({mid / span:5.3f})   resident {res:5d}   started {started:5d}   ended {ended:5d}")
order = np.argsort(t0)
print("start clock of the k-th started wave:", {int(k): int(t0[order[k]]) for k in (0, 255, 1023, 2047, 4095, 6143, B - 1) if k < B})
# lifetime vs start time: do later waves run faster (less contention) or slower?
first = t0 < np.percentile(t0, 45); second = ~first
print(f"lifetime of waves started in the first round: mean {life[first].mean():.0f}; later: mean {life[second].mean():.0f}")
