"""CPU checks of the oracle's general convex collider (mj: mjc_Convex restated as the minimum-translation problem it solves,
oracle/fly_oracle.c `convex_distance`) and of mjc_SphereCylinder.

MuJoCo itself is absent (parity unpinned, see the oracle header), so the collider is pinned by what defines its answer:
 * the signed-distance functions it is built on against finite differences and against the support functions;
 * a primal-dual certificate: for every unit n, -o(n) <= dist (o = overlap of the two geoms along n), so a returned (dist, n)
   with dist + o(n) = 0 is optimal - checked for every pair of geom types the fly has, separated and overlapping;
 * the direction found against a dense sampling of directions (no sampled direction overlaps less);
 * pairs with a closed form: sphere - capsule and capsule - capsule through the general routine equal the analytic colliders the
   oracle already uses for them, and sphere - cylinder through it equals the restated mjc_SphereCylinder.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle.oracle import _dp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
BLOB = os.path.join(ROOT, "flybody_amd", "assets", "fly_ball.ffmb")
NAMES = json.load(open(BLOB.replace(".ffmb", ".json")))["geom_name"]
SPHERE, CAPSULE, ELLIPSOID, CYLINDER = 2, 3, 4, 5


class Scene:
    """The ball model's geoms, posed at will by writing their world frames (the collider reads nothing else)."""

    def __init__(self):
        self.m = O.OracleModel(BLOB)
        self.d = O.OracleData(self.m)
        self.d.forward()
        m = self.m
        self.gt = np.ctypeslib.as_array(m.L.fo_geom_type(m.ptr), shape=(m.ngeom,))
        self.gs = np.ctypeslib.as_array(m.L.fo_geom_size(m.ptr), shape=(3 * m.ngeom,)).reshape(-1, 3)
        self.rng = np.random.RandomState(7)

    def geom(self, name):
        return NAMES.index(name)

    def sdf(self, g, x, hess=True):
        x = np.ascontiguousarray(x, dtype=np.float64)
        gr, H = np.zeros(3), np.zeros((3, 3))
        f = self.m.L.fo_geom_sdf(self.m.ptr, self.d.ptr, g, _dp(x), _dp(gr), _dp(H) if hess else None)
        return f, gr, H

    def support(self, g, n):
        out, n = np.zeros(3), np.ascontiguousarray(n, dtype=np.float64)
        self.m.L.fo_geom_support(self.m.ptr, self.d.ptr, g, _dp(n), _dp(out))
        return out

    def overlap(self, g1, g2, n):
        n = n / np.linalg.norm(n)
        return float(n @ (self.support(g1, n) - self.support(g2, -n)))

    def distance(self, g1, g2):
        n, p = np.zeros(3), np.zeros(3)
        dist = self.m.L.fo_convex_distance(self.m.ptr, self.d.ptr, g1, g2, _dp(n), _dp(p))
        return dist, n, p

    def small(self, g):
        return float(self.gs[g].min() if self.gt[g] == ELLIPSOID else self.gs[g][0])

    def randrot(self):
        q = self.rng.normal(size=4)
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])

    def place(self, g1, g2, lo, hi):
        """Random orientations and approach direction; geom2 then moved along the contact normal so that the distance becomes (to
        first order) a draw from [lo, hi] x the smaller geom's smallest size."""
        d = self.d
        sm = min(self.small(g1), self.small(g2))
        d.geom_xmat[g1], d.geom_xmat[g2] = self.randrot(), self.randrot()
        d.geom_xpos[g1] = 0
        d.geom_xpos[g2] = 0
        u = self.rng.normal(size=3)
        u /= np.linalg.norm(u)
        t = u @ (self.support(g1, u) - self.support(g2, -u))
        d.geom_xpos[g2] = u * (t + 0.5 * sm)
        # (the overlap along u only bounds the distance: measure it and close in along the true normal)
        dist, n, _ = self.distance(g1, g2)
        d.geom_xpos[g2] += n * (self.rng.uniform(lo, hi) * sm - dist)
        return sm


@pytest.fixture(scope="module")
def scene():
    return Scene()


PAIRS = {  # one representative per pair of types (geom1 has the lower type code, as mj_collision orders them)
    "capsule-cylinder": ("femur_T3_left_collision", "abdomen_3_collision"),
    "capsule-ellipsoid": ("haustellum_collision", "head_collision"),
    "sphere-ellipsoid": ("abdomen_7_collision", "thorax_collision"),
    "ellipsoid-ellipsoid": ("labrum_left_lower_collision", "labrum_right_lower_collision"),
    "ellipsoid-ellipsoid (thin wing)": ("thorax_collision", "wing_left_brown_collision"),
    "ellipsoid-cylinder": ("coxa_T3_left_collision", "abdomen_2_collision"),
    "cylinder-cylinder": ("abdomen_collision", "abdomen_4_collision"),
    "sphere-cylinder": ("ball_geom", "abdomen_6_collision"),
}


def test_every_fly_pair_is_collided_now(scene):
    m = scene.m
    assert m.ngeom == 71 and m.npair == 2358 and m.npair_unsupported == 0


def test_signed_distance_gradient_and_hessian(scene):
    s, rng = scene, scene.rng
    for name in ("abdomen_7_collision", "femur_T3_left_collision", "head_collision", "wing_left_brown_collision", "abdomen_3_collision"):
        g = s.geom(name)
        c, R, size, t = s.d.geom_xpos[g].copy(), s.d.geom_xmat[g].copy(), s.gs[g], s.gt[g]
        ext = np.array([size[0], size[0], size[0] + size[1]]) if t in (SPHERE, CAPSULE) else (np.array([size[0], size[0], size[1]]) if t == CYLINDER else size)
        checked = 0
        for _ in range(300):
            x = c + R @ (rng.uniform(-1.5, 1.5, 3) * ext)
            f, gr, H = s.sdf(g, x)
            assert abs(np.linalg.norm(gr) - 1) < 1e-12
            # the nearest surface point has the support value of its own normal (outside points)
            if f > 0:
                p = x - f * gr
                assert abs(gr @ (s.support(g, gr) - p)) < 1e-9 * ext.max()
            eps = 1e-6 * ext.min()
            gfd, Hfd = np.zeros(3), np.zeros((3, 3))
            for k in range(3):
                e = np.zeros(3)
                e[k] = eps
                fp, gp, _ = s.sdf(g, x + e, False)
                fm, gm, _ = s.sdf(g, x - e, False)
                gfd[k], Hfd[:, k] = (fp - fm) / (2 * eps), (gp - gm) / (2 * eps)
            if np.abs(Hfd - Hfd.T).max() * ext.min() > 1e-3:
                continue  # the difference stencil straddles a seam between two smooth pieces (cylinder rim / cap, capsule end)
            assert np.abs(gfd - gr).max() < 1e-5, name
            assert np.abs(Hfd - H).max() * ext.min() < 2e-3 * max(1.0, np.abs(H).max() * ext.min()), name
            checked += 1
        assert checked > 200


@pytest.mark.parametrize("kind", sorted(PAIRS))
def test_distance_is_certified_by_its_own_direction(scene, kind):
    s = scene
    g1, g2 = (s.geom(n) for n in PAIRS[kind])
    assert s.gt[g1] <= s.gt[g2]
    nsep = npen = 0
    for _ in range(150):
        sm = s.place(g1, g2, -0.3, 0.5)
        dist, n, pos = s.distance(g1, g2)
        assert abs(np.linalg.norm(n) - 1) < 1e-12
        # -o(n) <= true distance <= the value found: equality certifies it
        assert abs(dist + s.overlap(g1, g2, n)) < 1e-6 * sm, (kind, dist)
        # the contact position is the midpoint of two witness points that face each other along the normal, each on its surface
        f1, f2 = s.sdf(g1, pos - 0.5 * dist * n, False)[0], s.sdf(g2, pos + 0.5 * dist * n, False)[0]
        assert abs(f1) < 1e-7 * sm and abs(f2) < 1e-7 * sm, (kind, dist, f1, f2)
        nsep += dist > 0
        npen += dist < 0
    assert nsep > 20 and npen > 20


def test_no_sampled_direction_overlaps_less(scene):
    s = scene
    n_dir = 1500
    i = np.arange(n_dir) + 0.5
    ph, th = np.arccos(1 - 2 * i / n_dir), np.pi * (1 + 5 ** 0.5) * i
    dirs = np.stack([np.cos(th) * np.sin(ph), np.sin(th) * np.sin(ph), np.cos(ph)], 1)
    for kind in ("capsule-cylinder", "ellipsoid-cylinder", "ellipsoid-ellipsoid", "capsule-ellipsoid"):
        g1, g2 = (s.geom(n) for n in PAIRS[kind])
        for _ in range(8):
            sm = s.place(g1, g2, -0.4, -0.02)
            dist, n, _ = s.distance(g1, g2)
            assert dist < 0
            lowest = min(s.overlap(g1, g2, q) for q in dirs)
            assert -dist <= lowest + 1e-9 * sm, kind


def test_general_routine_reproduces_the_analytic_colliders(scene):
    """sphere - capsule / capsule - capsule (closed forms the oracle uses for those pairs) and mjc_SphereCylinder."""
    s, rng = scene, scene.rng
    sp, ca, cb, cy = s.geom("abdomen_7_collision"), s.geom("femur_T3_left_collision"), s.geom("tibia_T1_right_collision"), s.geom("abdomen_5_collision")
    for _ in range(60):
        s.place(sp, ca, -0.3, 0.5)
        dist, n, pos = s.distance(sp, ca)
        c, a, half = s.d.geom_xpos[ca], s.d.geom_xmat[ca][:, 2], s.gs[ca][1]
        t = np.clip(a @ (s.d.geom_xpos[sp] - c), -half, half)
        v = c + t * a - s.d.geom_xpos[sp]
        want = np.linalg.norm(v) - s.gs[sp][0] - s.gs[ca][0]
        assert abs(dist - want) < 1e-12 and np.allclose(n, v / np.linalg.norm(v), atol=1e-9)
        assert np.allclose(pos, s.d.geom_xpos[sp] + n * (s.gs[sp][0] + 0.5 * want), atol=1e-10)
    for _ in range(60):
        s.place(ca, cb, -0.3, 0.5)
        dist, n, _ = s.distance(ca, cb)
        # closest points of two segments by dense parametrisation refined around the minimum
        p1, a1, l1, p2, a2, l2 = s.d.geom_xpos[ca], s.d.geom_xmat[ca][:, 2], s.gs[ca][1], s.d.geom_xpos[cb], s.d.geom_xmat[cb][:, 2], s.gs[cb][1]
        lo1, hi1, lo2, hi2 = -l1, l1, -l2, l2
        for _ in range(6):
            t1, t2 = np.linspace(lo1, hi1, 41), np.linspace(lo2, hi2, 41)
            D = np.linalg.norm((p1[None] + t1[:, None] * a1[None])[:, None, :] - (p2[None] + t2[:, None] * a2[None])[None, :, :], axis=2)
            i1, i2 = np.unravel_index(D.argmin(), D.shape)
            w1, w2 = (hi1 - lo1) / 40, (hi2 - lo2) / 40
            lo1, hi1, lo2, hi2 = max(-l1, t1[i1] - w1), min(l1, t1[i1] + w1), max(-l2, t2[i2] - w2), min(l2, t2[i2] + w2)
        assert abs(dist - (D.min() - s.gs[ca][0] - s.gs[cb][0])) < 1e-7 * s.gs[ca][0]
    # mjc_SphereCylinder: the sphere centre against the side wall, a cap or the rim
    hits = set()
    for _ in range(200):
        sm = s.place(sp, cy, -0.3, 0.5)
        dist, n, pos = s.distance(sp, cy)
        c, a, R, H, r = s.d.geom_xpos[cy], s.d.geom_xmat[cy][:, 2], s.gs[cy][0], s.gs[cy][1], s.gs[sp][0]
        vec = s.d.geom_xpos[sp] - c
        x = a @ vec
        pp = vec - a * x
        side, cap = abs(x) < H, pp @ pp < R * R
        if side and cap:
            continue  # centre inside the cylinder: deeper than a contact gets
        if side:
            tgt, rr, which = c + a * x, R, "side"
        elif cap:
            tgt, rr, which = None, 0.0, "cap"
        else:
            tgt, rr, which = c + a * np.sign(x) * H + pp * R / np.linalg.norm(pp), 0.0, "rim"
        if which == "cap":
            want, wn = abs(x) - H - r, -np.sign(x) * a
        else:
            v = tgt - s.d.geom_xpos[sp]
            want, wn = np.linalg.norm(v) - r - rr, v / np.linalg.norm(v)
        hits.add(which)
        assert abs(dist - want) < 1e-9 * sm and np.allclose(n, wn, atol=1e-7), which
        assert np.allclose(pos, s.d.geom_xpos[sp] + wn * (r + 0.5 * want), atol=1e-9)
    assert hits == {"side", "cap", "rim"}


def test_contacts_of_the_standing_fly(scene):
    """walk_on_ball at rest, then driven: the contacts the general collider adds carry MuJoCo's parameters - condim 1, no friction
    row, the labrum pair inside its margin but outside `margin - gap` (detected, no force)."""
    m = O.OracleModel(BLOB)
    env = O.OracleBallEnv(m)
    env.reset()
    d = env.data
    rows = [r for r in d.contacts() if "ball" not in NAMES[int(r[0])]]
    assert [(NAMES[int(r[0])], NAMES[int(r[1])]) for r in rows] == [("labrum_left_lower_collision", "labrum_right_lower_collision")]
    assert rows[0][2] == 1 and rows[0][3] == 1 and 0 < rows[0][5] < 5e-4 and rows[0][15] == 0.0
    rng = np.random.RandomState(0)
    seen = set()
    for _ in range(120):
        env.step(rng.uniform(-0.2, 0.2, 59))
        for r in d.contacts():
            a, b = NAMES[int(r[0])], NAMES[int(r[1])]
            if "ball" in a:
                continue
            seen.add((a, b))
            assert r[2] == 1 and r[5] <= r[14] + (5e-4 if "labrum" in a or "claw" in a or "claw" in b else 0.0)
            if r[3] == 0:
                assert r[5] < r[14] and r[15] >= 0.0  # active: inside margin - gap, pushes only
    assert ("haustellum_collision", "head_collision") in seen and any("abdomen" in b for _, b in seen)
