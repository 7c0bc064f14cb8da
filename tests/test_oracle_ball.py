"""Known-answer and consistency checks for the walk_on_ball half of the float64 oracle (contacts, elliptic
cones, noslip, adhesion, activation filters, touch / force sensors).  MuJoCo is not available to this build
(SURVEY.md section 8c), so the physics stays "parity unpinned"; these tests pin the restatement to the
published definitions instead: analytic derivatives of the constraint cost, KKT conditions of the solve,
static force balance, and closed-form filter / adhesion answers."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O

BALL_BLOB = os.path.join(ROOT, "flybody_amd", "assets", "fly_ball.ffmb")
META = json.load(open(os.path.join(ROOT, "flybody_amd", "assets", "fly_ball.json")))


def _blob():
    from flybody_amd.model.blob import read_blob

    return read_blob(BALL_BLOB)


BLOB_T = _blob()


def _geom_mat(d, g):
    return d.geom_xmat[g]


@pytest.fixture()
def ball():
    m = O.OracleModel(BALL_BLOB)
    return m, O.OracleBallEnv(m)


def test_model_dimensions_match_survey_appendix_a(ball):
    m, env = ball
    # SURVEY.md App. A "Ball" column: nq 106 / nv 105, nu = na = 59, obs 289, 102 hinge limits, 70 fly geoms + sphere
    assert (m.nq, m.nv, m.nu, m.na, env.OBS, env.naction, m.ngeom) == (106, 105, 59, 59, 289, 59, 71)
    assert (m.ntouch, m.nforce) == (6, 6)
    assert sum(n for _, n in env.LAYOUT) == 289
    names = META["action_names"]
    assert names[0].startswith("adhere_claw") and names[6] == "head_abduct" and names[9] == "abdomen_abduct"
    assert names[11] == "coxa_abduct_T1_left" and len(names) == 59  # `fruitfly.py:25` class order


def test_reset_state_and_first_observation(ball):
    m, env = ball
    st, r, dsc, obs = env.reset()
    o = env.split(obs)
    assert (st, r, dsc) == (0, 0.0, 1.0)
    # thorax is welded to the world: accelerometer reads +g along world z in the (axis-aligned) thorax frame;
    # the reset buffer is zero padded, so the first mean is value / 10
    assert np.allclose(o["accelerometer"], [0, 0, 98.1]) and np.allclose(o["world_zaxis"], [0, 0, 1])
    assert not o["gyro"].any() and not o["velocimeter"].any() and not o["actuator_activation"].any()
    env.set_pad_first_obs(True)
    o2 = env.split(env.reset()[3])
    assert np.allclose(o2["accelerometer"], [0, 0, 981.0]) and np.allclose(o2["touch"], 10 * o["touch"])
    # wings folded to their spring reference (`fruitfly.py:335-340`)
    d = env.data
    for i, n in enumerate(META["jnt_name"]):
        if "wing" in n:
            assert d.qpos[4 + i - 1] == m.qpos_spring[4 + i - 1] != 0.0


def test_contact_geometry_ball_vs_claw(ball):
    m, env = ball
    env.reset()
    d = env.data
    c = np.array([row for row in d.contacts() if META["geom_name"][int(row[0])] == "ball_geom" or META["geom_name"][int(row[1])] == "ball_geom"])
    assert len(c) >= 4  # (the fly's own pairs - the labrum halves sit inside their margin at rest - are covered by test_oracle_convex.py)
    ball_c, R = d.geom_xpos[0], 0.454
    for row in c:
        assert int(row[0]) == 0  # sphere (lower type code) is geom1, so the normal points from the ball to the leg
        pos, nrm, dist = row[6:9], row[9:12], row[5]
        assert abs(np.linalg.norm(nrm) - 1) < 1e-12
        # contact point sits half-way inside the overlap along the normal from the ball centre
        assert np.allclose(pos, ball_c + nrm * (R + 0.5 * dist), atol=1e-12)
        assert row[13] == 1.0 if "claw" in META["geom_name"][int(row[1])] else row[13] == 0.5  # max(geom frictions)


def test_constraint_cost_derivatives_all_cone_zones(ball):
    m, env = ball
    env.reset()
    d = env.data
    J, aref, D, ty = d.efc()
    n = len(aref)
    assert (ty == 2).sum() >= 12
    rng = np.random.RandomState(0)
    zones = set()
    for trial in range(40):
        jar = rng.randn(n) * rng.choice([1e-1, 1.0, 10.0])
        if trial % 3 == 0:
            jar[ty == 2] *= np.tile([-3, 0.3, 0.3], (ty == 2).sum() // 3)  # push towards the bottom zone
        cost, force, _ = d.constraint_eval(jar)
        g = np.zeros(n)
        for i in range(n):
            e = np.zeros(n)
            e[i] = 1e-6 * max(1.0, abs(jar[i]))
            g[i] = (d.constraint_eval(jar + e)[0] - d.constraint_eval(jar - e)[0]) / (2 * e[i])
        assert np.allclose(force, -g, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(g).max())), trial
        for i in np.where(ty == 2)[0][::3]:
            f = force[i : i + 3]
            zones.add("zero" if not f.any() else ("quad" if np.allclose(f, -D[i : i + 3] * jar[i : i + 3]) else "cone"))
    assert zones == {"zero", "quad", "cone"}
    # Hessian of the cost in qacc space = J' d2s J: compare with finite differences of J' force
    jar = rng.randn(n)
    _, f0, H = d.constraint_eval(jar, hessian=True)
    v = rng.randn(m.nv)
    eps = 1e-6
    fp = d.constraint_eval(jar + eps * (J @ v))[1]
    fm = d.constraint_eval(jar - eps * (J @ v))[1]
    Hv_fd = -(J.T @ (fp - fm)) / (2 * eps)
    assert np.allclose(H @ v, Hv_fd, rtol=1e-4, atol=1e-6 * np.abs(Hv_fd).max())
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > -1e-8 * np.abs(H).max()


def test_solver_satisfies_kkt_and_friction_cones(ball):
    m, env = ball
    env.reset()
    rs = np.random.RandomState(1)
    for _ in range(30):
        env.step(rs.uniform(-0.5, 0.5, 59))
    d = env.data
    for flags in (O.FO_NO_NOSLIP, 0):
        m.set_flags(flags)
        d.forward()
        J, aref, D, ty = d.efc()
        n = len(aref)
        force = d.efc_force[:n].copy()
        M = d.dense_M()
        if flags:  # pure Newton: stationarity M a - f_smooth - J' f(a) = 0 and f = -ds/djar at the solution
            grad = M @ d.qacc - (M @ d.qacc_smooth) - J.T @ force
            assert np.abs(grad).max() < 1e-9 * max(1.0, np.abs(M @ d.qacc).max())
            _, f_check, _ = d.constraint_eval(J @ d.qacc - aref)
            assert np.allclose(force, f_check, rtol=1e-12, atol=1e-14)
        else:  # after noslip: qacc = qacc_smooth + M^-1 J' f, tangential force inside the *true* friction cone
            assert np.allclose(M @ (d.qacc - d.qacc_smooth), J.T @ force, rtol=1e-9, atol=1e-12)
        for row in d.contacts():
            a = int(row[4])
            if a < 0 or int(row[2]) != 3:
                continue
            fn, ft = force[a], np.linalg.norm(force[a + 1 : a + 3])
            mu = row[12] if flags else row[13]
            assert fn >= 0 and ft <= mu * fn * (1 + 1e-9) + 1e-12
        assert (force[ty != 2] >= 0).all()
    m.set_flags(0)


def test_noslip_reduces_tangential_slip_acceleration(ball):
    m, env = ball
    env.reset()
    rs = np.random.RandomState(2)
    for _ in range(20):
        env.step(rs.uniform(-0.5, 0.5, 59))
    d = env.data
    res = {}
    for flags in (O.FO_NO_NOSLIP, 0):
        m.set_flags(flags)
        d.forward()
        J, aref, D, ty = d.efc()
        r = J @ d.qacc - aref
        tang = np.array([k for i in np.where(ty == 2)[0][::3] for k in (i + 1, i + 2)])
        res[flags] = np.abs(r[tang]).sum()
    m.set_flags(0)
    assert res[0] < res[O.FO_NO_NOSLIP]


def test_qcqp2_matches_brute_force():
    L = O.lib()
    rng = np.random.RandomState(0)
    for _ in range(50):
        B = rng.randn(2, 2)
        A = B @ B.T + 0.1 * np.eye(2)
        b = rng.randn(2) * 3
        dd = rng.uniform(0.3, 1.5, 2)
        r = rng.uniform(0.1, 2.0)
        res = np.zeros(2)
        L.fo_debug_qcqp2(O._dp(res), O._dp(np.ascontiguousarray(A)), O._dp(b), O._dp(dd), float(r))
        th = np.linspace(0, 2 * np.pi, 20001)
        cand = [np.linalg.solve(A, -b)] if np.sum((np.linalg.solve(A, -b) / dd) ** 2) <= r * r else []
        ring = np.stack([r * dd[0] * np.cos(th), r * dd[1] * np.sin(th)], 1)
        f = lambda x: 0.5 * np.einsum("...i,ij,...j", x, A, x) + x @ b
        best = min([f(c) for c in cand] + [f(ring).min()])
        assert f(res) <= best + 1e-6 * (1 + abs(best)) and np.sum((res / dd) ** 2) <= r * r * (1 + 1e-8)


def test_ray_capsule():
    L = O.lib()

    def ray(o, v, rad, half):
        v = np.asarray(v, float) / np.linalg.norm(v)
        return L.fo_debug_ray_capsule(O._dp(np.asarray(o, float)), O._dp(v), rad, half)

    assert ray([0, 0, 0], [1, 0, 0], 0.1, 0.5) == 0  # origin inside
    assert abs(ray([1, 0, 0.2], [-1, 0, 0], 0.1, 0.5) - 0.9) < 1e-12  # side hit
    assert abs(ray([0, 0, 2], [0, 0, -1], 0.1, 0.5) - 1.4) < 1e-12  # cap hit
    assert ray([1, 0, 0], [1, 0, 0], 0.1, 0.5) == -1  # pointing away
    assert ray([1, 0, 2], [-1, 0, 0], 0.1, 0.5) == -1  # passes above the cap


def test_activation_filter_is_explicit_euler_first_order_lag(ball):
    m, env = ball
    env.reset()
    a = np.zeros(59)
    a[11:] = 0.1  # leg position targets
    a[:6] = 1.0  # adhesion
    for k in range(1, 4):
        env.step(a)
        n = 10 * k
        d = env.data
        # act_{n} = ctrl (1 - (1 - h/tau)^n): tau 0.01 for the joint servos, 0.007 for adhesion (`fly_envs.py:147-148`)
        assert np.allclose(d.act[20], min(0.1, 10) * (1 - (1 - 2e-4 / 0.01) ** n), rtol=1e-12)
        assert np.allclose(d.act[53:], 1 - (1 - 2e-4 / 0.007) ** n, rtol=1e-12)
    # ctrl is clamped to ctrlrange before the filter (adhesion range [0, 1])
    env.reset()
    a[:6] = -0.2
    env.step(a)
    assert not env.data.act[53:].any()


def _settle(env, action, steps=400):
    for _ in range(steps):
        env.step(action)


def test_static_balance_touch_force_and_adhesion(ball):
    m, env = ball
    env.reset()
    _settle(env, np.zeros(59))
    d = env.data
    assert np.abs(d.qvel).max() < 0.05
    d.forward()  # instantiate the constraint rows for the *current* contacts (a step ends on the position stage)
    c = d.contacts()
    blob = __import__("flybody_amd.model.blob", fromlist=["read_blob"]).read_blob(BALL_BLOB)
    geom_body = blob["geom_bodyid"]
    # touch sensor i == normal force of the contacts on claw i
    for t, site in enumerate(blob["touch_site"]):
        body = blob["sites_bodyid"][site]
        fn = sum(row[15] for row in c if geom_body[int(row[1])] == body and row[4] >= 0)
        assert abs(d.sens_touch[t] - fn) <= 1e-12 + 1e-9 * fn
    # at rest the ball feels no net torque: sum r x f over its contacts vanishes
    n = d.nefc
    J, aref, D, ty = d.efc()
    tau_ball = (J.T @ d.efc_force[:n])[:3]
    assert np.abs(tau_ball).max() < 2e-3 * np.abs(d.efc_force[:n]).max() * 0.454
    # adhesion: full command on every claw raises the summed normal load by ~ gain * act per claw in contact
    base = {int(r[1]): r[15] for r in c if r[4] >= 0 and "claw" in META["geom_name"][int(r[1])]}
    a = np.zeros(59)
    a[:6] = 1.0
    _settle(env, a)
    env.data.forward()
    after = {int(r[1]): r[15] for r in env.data.contacts() if r[4] >= 0 and "claw" in META["geom_name"][int(r[1])]}
    common = set(base) & set(after)
    assert len(common) >= 4
    gain = sum(after[g] - base[g] for g in common) / len(common)
    assert 0.5 < gain < 1.3, gain  # 0.985 per claw (`fruitfly.xml:25`), shared with the neighbouring tarsal contacts


def test_force_sensor_static_identity(ball):
    """At rest cfrc_int of a tarsus = -(weight of its subtree) - (contact forces on the subtree)."""
    m, env = ball
    env.reset()
    _settle(env, np.zeros(59), 600)
    d = env.data
    d.forward()
    from flybody_amd.model.blob import read_blob
    from flybody_amd.model import quat as Q

    blob = read_blob(BALL_BLOB)
    parent, mass = blob["body_parentid"], blob["body_mass"]
    c = d.contacts()
    for f, site in enumerate(blob["force_site"]):
        b = blob["sites_bodyid"][site]
        sub = [i for i in range(len(parent)) if _is_desc(parent, i, b)]
        w = sum(mass[i] for i in sub) * np.array([0, 0, -981.0])
        fc = np.zeros(3)
        for row in c:
            if row[4] < 0:
                continue
            a = int(row[4])
            frame0 = row[9:12]
            # rebuild the tangents the same way the oracle does (mju_makeFrame) to express the full contact force
            t1 = np.array([0, 1.0, 0]) if abs(frame0[1]) < 0.5 else np.array([0, 0, 1.0])
            t1 = t1 - frame0 * (frame0 @ t1)
            t1 /= np.linalg.norm(t1)
            t2 = np.cross(frame0, t1)
            fg = frame0 * d.efc_force[a] + (t1 * d.efc_force[a + 1] + t2 * d.efc_force[a + 2] if int(row[2]) == 3 else 0)
            if blob["geom_bodyid"][int(row[1])] in sub:
                fc += fg
            if blob["geom_bodyid"][int(row[0])] in sub:
                fc -= fg
        Rs = Q.to_mat(Q.mul(d.xquat[b], blob["sites_quat"][site]))
        world = Rs @ d.sens_force[f]
        assert np.allclose(world, -w - fc, atol=2e-4), (f, world, -w - fc)


def _is_desc(parent, i, root):
    while i > 0:
        if i == root:
            return True
        i = parent[i]
    return False


def test_ball_spins_freely_without_contacts_and_fluid(ball):
    m, env = ball
    env.reset()
    d = env.data
    m.set_flags(O.FO_NO_CONTACT | O.FO_NO_FLUID)
    d.qvel[:3] = [1.0, -5.0, 0.5]
    for _ in range(500):
        d.step()
    assert np.allclose(d.qvel[:3], [1.0, -5.0, 0.5], rtol=1e-12) and abs(np.linalg.norm(d.qpos[:4]) - 1) < 1e-12
    # rotation angle = |w| t about the (body == world at start) axis
    ang = 2 * np.arccos(abs(d.qpos[0]))
    assert abs(ang - np.linalg.norm([1.0, -5.0, 0.5]) * 500 * 2e-4) < 1e-9
    # with the fluid on, the viscous torque -pi d^3 beta w decays the spin exponentially
    m.set_flags(O.FO_NO_CONTACT)
    w0 = d.qvel[:3].copy()
    for _ in range(500):
        d.step()
    assert 0 < np.linalg.norm(d.qvel[:3]) < np.linalg.norm(w0)
    m.set_flags(0)


def test_episode_protocol_time_limit_and_reward(ball):
    m, env = ball
    env = O.OracleBallEnv(m, time_limit=0.01)  # 50 float64 additions of 2e-4 reach 0.01: 5 control steps
    assert env.reset()[0] == 0
    d = env.data
    for k in range(5):
        st, r, dsc, obs = env.step(np.zeros(59))
        w = d.qvel[:3]
        expect = np.prod(np.maximum(0, 1 - np.abs(w - np.array([0, -5.0, 0])) / 6))  # `walk_on_ball.py:61-73`
        assert abs(r - expect) < 1e-15 and np.array_equal(env.split(obs)["ball_qvel"], w)
        assert st == (2 if k == 4 else 1) and dsc == 1.0
    assert env.step(np.zeros(59))[0] == 0  # auto-reset on the call after LAST


def test_config1_random_actions_stay_finite_and_deterministic(ball):
    """BASELINE.json configs[0]: walk_on_ball, 1 env, random raw actions U(-0.2, 0.2) (`task_utils.py:13-24`)."""
    m, env = ball
    outs = []
    for rep in range(2):
        env.reset()
        rs = np.random.RandomState(0)
        tot = 0.0
        for _ in range(60):
            st, r, dsc, obs = env.step(rs.uniform(-0.2, 0.2, 59))
            assert np.isfinite(obs).all() and st == 1
            tot += r
        outs.append((tot, obs.copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("time_limit", [0.01, 0.006, 0.0346, 0.05])
def test_time_limit_is_tested_on_accumulated_physics_time(ball, time_limit):
    """composer.Environment ends an episode on `physics.time() >= time_limit` with MuJoCo's time a float64 running sum of
    the physics timestep.  The oracle tests its own accumulated `time`; the host side of the product computes the step count
    by replaying the additions (`flybody_amd/batched_env.py:time_limit_control_steps`): both must agree, and for the
    reference's 2.0 s / 0.6 s limits they give 1001 / 3001 control steps, not 1000 / 3000."""
    from flybody_amd.batched_env import time_limit_control_steps

    m, _ = ball
    env = O.OracleBallEnv(m, time_limit=time_limit)
    env.reset()
    n = 0
    while True:
        n += 1
        if env.step(np.zeros(59))[0] == 2:
            break
    assert n == time_limit_control_steps(time_limit, m.timestep, 10)
    assert time_limit_control_steps(2.0, 2e-4, 10) == 1001 and time_limit_control_steps(0.6, 5e-5, 4) == 3001


def test_convex_separation_brackets_the_analytic_capsule_distance(ball):
    """The measurement routine behind tools/self_collision_stats.py (Gilbert's iteration on the Minkowski difference; used for
    the ellipsoid / cylinder pairs MuJoCo hands to its general convex collider) is checked where a closed form exists: for
    sphere / capsule pairs its lower and upper bounds must bracket the analytic distance of `fo_collision`'s narrow phase."""
    import ctypes as C

    m, env = ball
    L = m.L
    d = O.OracleData(m)
    rng = np.random.RandomState(0)
    names = META["geom_name"]
    prim = [g for g, n in enumerate(names) if "ball" not in n and int(BLOB_T["geom_type"][g]) in (2, 3)]
    checked = 0
    for trial in range(6):
        q = BLOB_T["qpos0"].copy()
        hinge = [j for j in range(len(BLOB_T["jnt_type"])) if BLOB_T["jnt_type"][j] == 3]
        for j in hinge:
            lo, hi = BLOB_T["jnt_range"][j]
            q[BLOB_T["jnt_qposadr"][j]] = np.clip(rng.uniform(-0.5, 0.5) * (hi - lo) / 2, lo, hi)
        d.qpos[:] = q
        d.forward()
        gx = d.geom_xpos
        for _ in range(60):
            g1, g2 = rng.choice(prim, 2, replace=False)
            up = C.c_double()
            lb = L.fo_convex_separation(m.ptr, d.ptr, int(g1), int(g2), C.byref(up))
            # analytic: closest points of the two segments (capsule axes; a sphere is a zero-length one)
            def seg(g):
                R = _geom_mat(d, g)
                half = BLOB_T["geom_size"][g][1] if int(BLOB_T["geom_type"][g]) == 3 else 0.0
                return gx[g] - R[:, 2] * half, gx[g] + R[:, 2] * half, BLOB_T["geom_size"][g][0]
            a0, a1, ra = seg(g1)
            b0, b1, rb = seg(g2)
            ts = np.linspace(0, 1, 401)
            pa = a0[None] + ts[:, None] * (a1 - a0)[None]
            pb = b0[None] + ts[:, None] * (b1 - b0)[None]
            dist = np.sqrt(((pa[:, None, :] - pb[None, :, :]) ** 2).sum(-1)).min() - ra - rb   # dense sampling: <= 1e-4 above the true minimum
            if dist > 1e-3:
                assert lb <= dist + 1e-9 and up.value >= dist - 2e-4 and up.value - lb < 2e-4, (names[g1], names[g2], lb, up.value, dist)
                checked += 1
            else:
                assert lb < 2e-3
    assert checked > 200
