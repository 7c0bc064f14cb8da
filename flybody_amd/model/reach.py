"""Static candidate list of collision pairs for the device kernels.

MuJoCo tests every geom pair that survives its filters (same weld body, parent - child, <exclude>, contype / conaffinity:
`fruitfly.xml:16-25,733-760`, `tasks/walk_on_ball.py:33-40`) with a bounding-sphere test each step.  Most of those ~2 400 pairs can
never come near each other: their relative pose only depends on the hinges between the two bodies, and those have limits.
`candidate_pairs` keeps a pair when, over a sample of joint configurations that covers the limits (with a margin for the soft
limits' overshoot), the bounding spheres come within `slack` of each other; the kernels run MuJoCo's bounding-sphere test on
that list only.  The float64 CPU checker of the tests keeps the full list; tests/test_model_compiler.py checks that no contact it
ever finds in long rollouts is missing from the candidates.
"""
from __future__ import annotations

import numpy as np

from . import pyref
from . import quat as Q

PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER = 0, 2, 3, 4, 5


def bounding_radius(gtype: int, size) -> float:
    if gtype == SPHERE:
        return float(size[0])
    if gtype == CAPSULE:
        return float(size[0] + size[1])
    if gtype == CYLINDER:
        return float(np.hypot(size[0], size[1]))
    if gtype == ELLIPSOID:
        return float(max(size))
    return 0.0


def filtered_pairs(m):
    """(g1, g2) in mj_collision's order (body pairs, then geoms), geom1 = the lower type code."""
    nb, ng = m.nbody, len(m.geom_bodyid)
    weld, par = m.body_weldid, m.body_parentid
    excl = {(int(a), int(b)) for a, b in np.asarray(m.exclude_pairs).reshape(-1, 2)}
    by_body = [[] for _ in range(nb)]
    for g in range(ng):
        by_body[m.geom_bodyid[g]].append(g)
    out = []
    for b1 in range(nb):
        for b2 in range(b1 + 1, nb):
            w1, w2 = weld[b1], weld[b2]
            if w1 == w2:
                continue
            wp1, wp2 = weld[par[w1]], weld[par[w2]]
            if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                continue
            if (b1, b2) in excl or (b2, b1) in excl:
                continue
            for ga in by_body[b1]:
                for gb in by_body[b2]:
                    if not ((m.geom_contype[ga] & m.geom_conaffinity[gb]) or (m.geom_contype[gb] & m.geom_conaffinity[ga])):
                        continue
                    g1, g2 = (ga, gb) if m.geom_type[ga] <= m.geom_type[gb] else (gb, ga)
                    out.append((g1, g2))
    return out


def bounding_capsule(gtype: int, size):
    """(axis index, half length, radius) of a capsule around the geom, in the geom frame: the geom itself for a sphere / capsule,
    axis segment + a ball of the cylinder's corner distance... kept simple and safe: a cylinder's axis with its radius padded to reach
    the rim from the segment end, an ellipsoid's longest axis shortened by its middle semi-axis, that semi-axis (+5 %) as radius."""
    if gtype == SPHERE:
        return 2, 0.0, float(size[0])
    if gtype == CAPSULE:
        return 2, float(size[1]), float(size[0])
    if gtype == CYLINDER:
        return 2, float(size[1]), float(size[0])  # (segment of the full half length) + ball(R) contains segment + disc(R)
    if gtype == ELLIPSOID:
        k = int(np.argmax(size))
        mid = float(np.sort(size)[1])
        return k, float(size[k] - mid), 1.05 * mid
    return 2, 0.0, 0.0


def _segment_distance(p1, a1, l1, p2, a2, l2):
    """Distances between N pairs of segments p + x a, |x| <= l (vectorised closest points of two segments)."""
    dif = p1 - p2
    mb = -np.einsum("ij,ij->i", a1, a2)
    u = -np.einsum("ij,ij->i", a1, dif)
    v = np.einsum("ij,ij->i", a2, dif)
    det = 1.0 - mb * mb
    par = np.abs(det) < 1e-9
    det = np.where(par, 1.0, det)
    x1 = np.clip((u - mb * v) / det, -l1, l1)
    x1 = np.where(par, 0.0, x1)
    x2 = np.clip(v - mb * x1, -l2, l2)
    x1 = np.clip(u - mb * x2, -l1, l1)
    x2 = np.clip(v - mb * x1, -l2, l2)
    d = (p1 + x1[:, None] * a1) - (p2 + x2[:, None] * a2)
    return np.linalg.norm(d, axis=1)


def candidate_pairs(m, nsample: int = 1500, overshoot: float = 0.25, slack: float = 0.006, seed: int = 0):
    """Pairs of `filtered_pairs` whose bounding capsules come within `slack` (cm) of each other somewhere in the joint box
    [lo - overshoot, hi + overshoot] (radians; unlimited hinges: +-pi).  Planes keep all their pairs."""
    pairs = filtered_pairs(m)
    ng = len(m.geom_bodyid)
    cap = [bounding_capsule(int(m.geom_type[g]), np.asarray(m.geom_size[g])) for g in range(ng)]
    cax = np.array([c[0] for c in cap])
    chalf = np.array([c[1] for c in cap])
    crad = np.array([c[2] for c in cap])
    margin = np.asarray(m.geom_margin)
    rng = np.random.RandomState(seed)
    hinge = [j for j in range(m.njnt) if m.jnt_type[j] == pyref.JNT_HINGE]
    lo = np.array([m.jnt_range[j][0] - overshoot if m.jnt_limited[j] else -np.pi for j in hinge])
    hi = np.array([m.jnt_range[j][1] + overshoot if m.jnt_limited[j] else np.pi for j in hinge])
    qadr = np.array([m.jnt_qposadr[j] for j in hinge])
    p1 = np.array([p[0] for p in pairs])
    p2 = np.array([p[1] for p in pairs])
    best = np.full(len(pairs), np.inf)
    reach = crad[p1] + crad[p2] + np.maximum(margin[p1], margin[p2])
    gmat0 = np.array([Q.to_mat(np.asarray(m.geom_quat[g])) for g in range(ng)])
    for s in range(nsample):
        q = np.array(m.qpos0, dtype=np.float64).copy()
        u = rng.uniform(0, 1, len(hinge))
        if s % 3 == 1:
            u = np.round(u)  # corners of the box
        elif s % 3 == 2:
            u = np.clip(rng.normal(0.5, 0.35, len(hinge)), 0, 1)
        if s == 0:
            q[qadr] = np.clip(q[qadr], lo, hi)
        else:
            q[qadr] = lo + u * (hi - lo)
        k = pyref.kinematics(m, q)
        bmat = np.array([Q.to_mat(k["xquat"][b]) for b in range(m.nbody)])
        gb = np.asarray(m.geom_bodyid)
        gpos = k["xpos"][gb] + np.einsum("gij,gj->gi", bmat[gb], np.asarray(m.geom_pos))
        gax = np.einsum("gij,gj->gi", bmat[gb], gmat0[np.arange(ng), :, cax])
        dist = _segment_distance(gpos[p1], gax[p1], chalf[p1], gpos[p2], gax[p2], chalf[p2]) - reach
        best = np.minimum(best, dist)
    keep = (best < slack) | (np.asarray(m.geom_type)[p1] == PLANE)
    return [pairs[i] for i in range(len(pairs)) if keep[i]], pairs
