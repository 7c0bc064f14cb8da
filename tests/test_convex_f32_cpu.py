"""The kernels' float32 narrow phase for the fly's convex pairs (flybody_amd/csrc/convex.hpp, compiled for the host with -DCVX_HOST by
tests/cvx_host.cpp) against the float64 oracle's general convex collider (oracle/fly_oracle.c `convex_distance`, itself pinned by
tests/test_oracle_convex.py).  Runs without a GPU: it is the same header the HIP kernels include.

 * every pair class the dispatcher `collide` knows, on random poses from separated to overlapping;
 * the pairs that really come near each other, on states of oracle rollouts of both tasks: every candidate pair that passes the
   broad phase's separating-direction bound - and that bound is checked to never cull a pair the oracle finds in contact.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from oracle.oracle import _dp
from test_oracle_convex import PAIRS, Scene

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HERE = os.path.dirname(os.path.abspath(__file__))
fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "oracle", "_build", "libcvx_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = [os.path.join(HERE, "cvx_host.cpp"), os.path.join(ROOT, "flybody_amd", "csrc", "convex.hpp")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in src):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", out, src[0]])
    L = C.CDLL(out)
    L.cvxh_collide.restype = C.c_float
    L.cvxh_separation_bound.restype = C.c_float
    return L


def _quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1 + R[i, i] - R[j, j] - R[k, k]) * 2
    q = np.zeros(4)
    q[0], q[1 + i], q[1 + j], q[1 + k] = (R[k, j] - R[j, k]) / s, 0.25 * s, (R[j, i] + R[i, j]) / s, (R[k, i] + R[i, k]) / s
    return q


def _rec(xpos, xmat, size, gtype):
    return np.ascontiguousarray(np.concatenate([xpos, _quat(xmat), size, [gtype]]), dtype=np.float32)


def _collide(L, a, b):
    n, p = np.zeros(3, np.float32), np.zeros(3, np.float32)
    d = L.cvxh_collide(a.ctypes.data_as(fp), b.ctypes.data_as(fp), n.ctypes.data_as(fp), p.ctypes.data_as(fp))
    return float(d), n.astype(np.float64), p.astype(np.float64)


# float32 bounds per class: |dist error| (cm), |normal error|, |position error| (cm) - each about 3 x the worst seen over 300 poses
BOUNDS = {
    "capsule-cylinder": (6e-8, 6e-6, 2e-7),
    "capsule-ellipsoid": (6e-8, 3e-6, 1e-7),
    "sphere-ellipsoid": (6e-8, 2e-6, 1e-7),
    "sphere-cylinder": (3e-7, 2e-6, 5e-7),
    "ellipsoid-ellipsoid": (4e-8, 6e-4, 3e-6),
    "ellipsoid-ellipsoid (thin wing)": (5e-7, 2e-3, 4e-4),  # a 0.0022 x 0.0175 x 0.114 blade flat on the thorax: the position along it is soft
    "ellipsoid-cylinder": (5e-8, 1.5e-3, 8e-6),  # (the normal of an overlap across the cylinder's rim is found by the dual Newton: softer)
}


@pytest.mark.parametrize("kind", sorted(BOUNDS))
def test_pair_class_on_random_poses(lib, kind):
    s = Scene()
    g1, g2 = (s.geom(n) for n in PAIRS[kind])
    s.rng = np.random.RandomState(11)
    off = np.array([0.1, -0.05, 0.2])  # world coordinates of the size the fly's geoms have in the tasks
    ed = en = ep = 0.0
    for _ in range(300):
        s.place(g1, g2, -0.15, 0.3)  # distance from 15 % of the thinner geom's size deep to 30 % apart
        d64, n64, p64 = s.distance(g1, g2)
        d, n, p = _collide(lib, _rec(s.d.geom_xpos[g1] + off, s.d.geom_xmat[g1], s.gs[g1], s.gt[g1]),
                           _rec(s.d.geom_xpos[g2] + off, s.d.geom_xmat[g2], s.gs[g2], s.gt[g2]))
        ed, en, ep = max(ed, abs(d - d64)), max(en, np.linalg.norm(n - n64)), max(ep, np.linalg.norm(p - off - p64))
    bd, bn, bp = BOUNDS[kind]
    assert ed < bd and en < bn and ep < bp, (kind, ed, en, ep)


def _rollout(kind):
    blob = os.path.join(ROOT, "flybody_amd", "assets", f"fly_{kind}.ffmb")
    m = O.OracleModel(blob)
    rng = np.random.RandomState(3)
    if kind == "ball":
        env = O.OracleBallEnv(m)
        draw = lambda: rng.uniform(-1.0, 1.0, 59)
    else:
        from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
        from flybody_amd.tasks.trajectories import preprocess
        from flybody_amd.tasks.wbpg import build_tables

        rq, rv = preprocess(*flight_trajectories(4, 3006))
        env = O.OracleFlightEnv(m, build_tables(base_wing_pattern()), rq, rv, seed=5, env_id=0)
        lo = np.array([-0.2, -3, -0.5, -1, -1, -1, -1, -1, -1, -0.7, -1.05, -1.0])
        hi = np.array([0.2, 3, 0.3, 1, 1, 1, 1, 1, 1, 0.7, 0.7, 1.0])
        draw = lambda: lo + (hi - lo) * rng.uniform(0, 1, 12)
    return m, env, draw


@pytest.mark.parametrize("kind", ["ball", "flight"])
def test_pairs_of_rollout_states(lib, kind):
    from flybody_amd.model.blob import read_blob

    m, env, draw = _rollout(kind)
    t = read_blob(os.path.join(ROOT, "flybody_amd", "assets", f"fly_{kind}.ffmb"))
    gt, gs, gm = np.asarray(t["geom_type"]), np.asarray(t["geom_size"]).reshape(-1, 3), np.asarray(t["geom_margin"])
    g1s, g2s = np.asarray(t["cand_g1"]), np.asarray(t["cand_g2"])
    conv = [(int(a), int(b)) for a, b in zip(g1s, g2s) if (gt[a] >= 4 or gt[b] >= 4) and gt[a] >= 2]
    cand = set(conv)
    env.reset()
    d = env.data
    ng = len(gt)
    n, p = np.zeros(3), np.zeros(3)
    ntested = ntouch = 0
    worst = np.zeros(3)
    for step in range(60):
        env.step(draw())
        for row in d.contacts():  # every convex contact the oracle made is a candidate pair of the kernels' static list
            a, b = int(row[0]), int(row[1])
            if gt[a] >= 2 and (gt[a] >= 4 or gt[b] >= 4) and not (kind == "ball" and a == 0):
                assert (a, b) in cand, (a, b)
        xp = np.ctypeslib.as_array(m.L.fo_geom_xpos(d.ptr), (3 * ng,)).reshape(ng, 3).copy()
        xm = np.ctypeslib.as_array(m.L.fo_geom_xmat(d.ptr), (9 * ng,)).reshape(ng, 3, 3).copy()
        for a, b in conv:
            if np.linalg.norm(xp[a] - xp[b]) > 0.12:
                continue
            ra, rb = _rec(xp[a], xm[a], gs[a], gt[a]), _rec(xp[b], xm[b], gs[b], gt[b])
            bound = lib.cvxh_separation_bound(ra.ctypes.data_as(fp), rb.ctypes.data_as(fp))
            margin = max(gm[a], gm[b])
            d64 = m.L.fo_convex_distance(m.ptr, d.ptr, a, b, _dp(n), _dp(p))
            assert bound <= d64 + 1e-6, (a, b, bound, d64)  # a lower bound: it never culls a touching pair
            if bound > margin + 1e-3:
                continue
            dd, nn, pp = _collide(lib, ra, rb)
            ntested += 1
            ntouch += d64 <= margin
            if d64 > margin + 5e-4:  # far from a contact, only the decision matters
                assert dd > margin, (a, b, d64, dd)
                continue
            deep = d64 < -0.5 * min(gs[a][gs[a] > 0].min(), gs[b][gs[b] > 0].min())
            if deep:
                continue  # (a wing driven through the abdomen by full-range random actions: several stationary directions)
            thin = min(gs[a][gs[a] > 0].min(), gs[b][gs[b] > 0].min()) < 0.0035  # wing blades: soft position along the blade
            e = np.array([abs(dd - d64), np.linalg.norm(nn - n), np.linalg.norm(pp - p)])
            assert e[0] < 3e-7 and e[1] < (2e-3 if thin else 2e-4) and e[2] < (4e-4 if thin else 2e-5), (a, b, d64, dd, e)
            worst = np.maximum(worst, e)
    assert ntested > 200 and ntouch > 30, (ntested, ntouch)
    print(f"{kind}: {ntested} pair evaluations ({ntouch} within margin), worst |dist| {worst[0]:.1e} |n| {worst[1]:.1e} |pos| {worst[2]:.1e}")
