"""CPU checks of the walk_imitation restatement (oracle/fly_oracle.c) and of the host-side walking task code
(flybody_amd/tasks/walking.py): the reference's own quaternion helpers through goldens generated from the imported
reference (tools/gen_golden.py), plane contacts through known answers, the episode protocol through
walk_imitation.py's rules.  The walking dataset is not in the reference repository: snippets here are synthetic and the choice
of tracked joints / sites is ours (parity unpinned on that choice)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O
from flybody_amd.tasks import rewards as R
from flybody_amd.tasks import walking as W

GOLD = os.path.join(ROOT, "tests", "golden", "walker_features.npz")
BLOB = os.path.join(ROOT, "flybody_amd", "assets", "fly_walk.ffmb")


@pytest.fixture(scope="module")
def view():
    return W.WalkModelView()


@pytest.fixture(scope="module")
def refs(view):
    return W.WalkRefSet(W.synthetic_snippets(view, n=3, length=100))


def make_env(view, refs, **kw):
    m = O.OracleModel(BLOB)
    return O.OracleWalkEnv(m, refs, view.mocap_jnt, view.mocap_site, (view.retract_qadr, view.retract_val), **kw)


def test_quaternion_helpers_match_the_reference_goldens():
    g = np.load(GOLD)
    for i in range(len(g["vec"])):
        want = g["z2vec"][i]
        assert np.allclose(W.quat_z2vec(g["vec"][i]), want, atol=1e-14), i
        assert np.allclose(O.test_quat(6, np.hstack((g["vec"][i], 0.0))), want, atol=1e-14), i
        want = g["joint_orientation"][i]
        assert np.allclose(W.joint_orientation_quat(g["xaxis"][i], g["ang"][i]), want, atol=1e-14)
        assert np.allclose(O.test_quat(7, np.hstack((g["xaxis"][i], 0.0)), np.array([g["ang"][i], 0, 0, 0])), want, atol=1e-14)
    # get_egocentric_vec = rotate(site - root, conj(root_quat)) (quaternions.py:137-159)
    from flybody_amd.model import quat as Q
    ego = np.array([Q.rot(s - g["root_pos"], Q.conj(g["root_quat"])) for s in g["sites"]])
    assert np.allclose(ego, g["egocentric"], atol=1e-14)
    # the sequence get_walker_features applies to the joint axes (rewards.py:45-52)
    rinv = Q.conj(g["root_quat"]) / np.dot(g["root_quat"], g["root_quat"])
    jq = np.array([W.joint_orientation_quat(Q.rot(a, rinv), q) for a, q in zip(g["xaxis"], g["ang"])])
    assert np.allclose(jq, g["joint_quat_local"], atol=1e-13)


def test_walk_model_dimensions(view):
    m = O.OracleModel(BLOB)
    assert (m.nq, m.nv, m.nu, m.na) == (109, 108, 59, 59)
    # the fly's own sphere / capsule pairs (1086) and its pairs with an ellipsoid or cylinder on one side (1202: the general convex
    # collider) + the floor against every primitive geom; the floor against an ellipsoid / cylinder is not restated
    assert m.npair == 1086 + 1202 + 48 and m.npair_unsupported == 22
    assert len(view.mocap_jnt) == 66 and len(view.mocap_site) == 6


def test_oracle_features_equal_the_numpy_restatement(view, refs):
    e = make_env(view, refs)
    e.force_next(1)
    e.reset()
    rng = np.random.RandomState(0)
    for _ in range(5):
        e.step(rng.uniform(-0.3, 0.3, e.naction))
    got = e.features()
    want = W.walker_features(view, e.data.qpos.copy(), e.data.qvel.copy())
    for k in want:
        assert np.allclose(got[k], want[k], atol=1e-12), k


def test_reset_places_the_walker_on_the_reference_and_earns_the_position_factors(view, refs):
    e = make_env(view, refs)
    e.force_next(2)
    st, r, d, obs = e.reset()
    assert (st, r, d) == (0, 0.0, 1.0) and e.traj_idx == 2 and obs.shape == (741,)
    sn = refs.snippet(2)
    assert np.allclose(e.data.qpos[:7], sn["qpos"][0, :7]) and np.allclose(e.data.qpos[view.mocap_qadr], sn["qpos"][0, 7:])
    assert np.allclose(e.data.qpos[view.retract_qadr], view.retract_val) and not e.data.qvel.any()
    f = e.reward_factors(0)
    # on the reference pose: com, end-effector and orientation factors are exact; the velocity factor sees the zero start
    assert abs(f[0] - 20.0) < 1e-9 and abs(f[2] - 1.0) < 1e-9 and abs(f[3] - 1.0) < 1e-9
    want = R.reward_factors_deep_mimic(e.features(), R.get_reference_features(sn, 0), weights=(20, 1, 1, 1))
    assert np.allclose(f, want, rtol=1e-12)
    o = e.split(obs)
    assert np.allclose(o["ref_displacement"][:3], 0.0, atol=1e-12) and np.allclose(o["ref_root_quat"][:4], [1, 0, 0, 0], atol=1e-12)
    # preview rows: reference root positions of rows 0..64 in the walker's frame (base.py:237-261)
    k = 17
    dv = sn["qpos"][k, :3] - e.data.qpos[:3]
    xmat = np.array(e.data.xquat[1])
    from flybody_amd.model import quat as Q
    assert np.allclose(o["ref_displacement"][3 * k : 3 * k + 3], Q.rot(dv, Q.conj(xmat)), atol=1e-12)


def test_reward_follows_rewards_py_during_a_rollout(view, refs):
    e = make_env(view, refs)
    e.force_next(0)
    e.reset()
    sn = refs.snippet(0)
    rng = np.random.RandomState(1)
    for k in range(1, 8):
        st, r, d, _ = e.step(rng.uniform(-0.5, 0.5, e.naction))
        want = np.prod(R.reward_factors_deep_mimic(e.features(), R.get_reference_features(sn, k), weights=(20, 1, 1, 1)))
        assert st == 1 and d == 1.0 and abs(r - want) <= 1e-12 * max(1.0, want)


def test_standing_fly_is_carried_by_the_floor(view, refs):
    """Known answer for the plane contacts: at rest the contact forces along the floor normal add up to the fly's weight."""
    e = make_env(view, refs, terminal_com_dist=float("inf"))
    e.force_next(0)
    e.reset()
    hold = np.zeros(e.naction)
    for _ in range(60):
        e.step(hold)
    d = e.data
    d.forward()  # (a step ends on the position stage of the next one: solve the constraints at this state to read the forces)
    assert d.ncon >= 3
    cons = d.contacts()  # rows: geom1, geom2, dim, exclude, efc_adr, dist, pos[3], normal[3], mu, friction, includemargin, normal force
    cons = [c for c in cons if c[0] == 0]  # the contacts with the floor plane (the fly's own pairs are internal forces)
    assert len(cons) >= 3
    assert all(np.allclose(c[9:12], [0, 0, 1]) for c in cons)  # normal = the plane's z axis, from the floor to the leg
    fz = sum(c[15] for c in cons)  # first row of a contact = its normal force
    from flybody_amd.model.blob import read_blob
    t = read_blob(BLOB)
    weight = float(np.sum(t["body_mass"])) * 981.0
    # adhesion is off (zero action) and the fly is quasi-static after 0.12 s: normal forces carry the weight
    assert abs(fz - weight) < 0.05 * weight, (fz, weight)
    assert 0.10 < d.qpos[2] < 0.14  # still standing at about its spawn height
    for c in cons:
        assert c[5] > -2e-3  # soft-contact penetration stays below 20 um


def test_episode_protocol(view, refs):
    e = make_env(view, refs, terminal_com_dist=float("inf"))
    e.force_next(1)
    e.reset()
    length = refs.off[2] - refs.off[1]
    episode_steps = length - 64 - 1  # walk_imitation.py:99-100
    k, st = 0, 1
    while st != 2:
        st, r, d, _ = e.step(np.zeros(e.naction))
        k += 1
        assert k <= episode_steps
    assert k == episode_steps and d == 1.0  # end of snippet: LAST with discount 1 (walk_imitation.py:179-187)
    st, r, d, _ = e.step(np.zeros(e.naction))
    assert (st, r, d) == (0, 0.0, 1.0)  # auto-reset on the next call
    # straying from the reference root by more than terminal_com_dist is fatal (discount 0)
    e2 = make_env(view, refs, terminal_com_dist=0.02)
    e2.force_next(0)
    e2.reset()
    for k in range(40):
        st, r, d, _ = e2.step(np.zeros(e2.naction))
        if st == 2:
            break
    assert st == 2 and d == 0.0  # the snippet's root moves at 1 cm/s, the standing fly stays behind
    # NaN actions are scrubbed (walk_imitation.py:143-144)
    e3 = make_env(view, refs)
    e3.reset()
    a = np.zeros(e3.naction); a[5] = np.nan
    st, r, d, obs = e3.step(a)
    assert np.isfinite(obs).all() and np.isfinite(r)


def test_walking_dataset_converter_and_npz_round_trip(tmp_path, view):
    """tools/convert_hdf5_to_npz.py --walking (h5py is absent: its layout logic is driven with an in-memory stand-in): lengths from
    `trajectory_lengths`, root x / y re-based, short snippets dropped; the npz round-trips into the set the oracle takes."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_hdf5_to_npz as conv

    rng = np.random.RandomState(3)
    J, S = len(view.mocap_jnt), len(view.mocap_site)
    lens, store = [80, 40, 120], []
    for L in lens:
        T = L + 7  # stored arrays are longer than the valid length
        store.append({"root_qpos": rng.randn(T, 7), "qpos": rng.randn(T, J), "root_qvel": rng.randn(T, 6), "qvel": rng.randn(T, J),
                      "root2site": rng.randn(T, S, 3), "joint_quat": rng.randn(T, J, 4)})
    dst = str(tmp_path / "walk.npz")
    n, lo, hi = conv.convert_walking(lambda i: store[i], 3, lens, view.meta["mocap_joints"], view.meta["mocap_sites"], dst)
    assert (n, lo, hi) == (2, 80, 120)  # the 40-row snippet cannot hold an episode (needs 66 rows)
    refs, jn, sn, dt = W.load_npz(dst)
    assert refs.ntraj == 2 and list(refs.off) == [0, 80, 200] and jn == view.meta["mocap_joints"] and sn == view.meta["mocap_sites"] and dt == 2e-3
    a = refs.snippet(1)
    assert np.allclose(a["qpos"][:, 2:7], store[2]["root_qpos"][:120, 2:]) and np.allclose(a["qpos"][:, 7:], store[2]["qpos"][:120])
    assert np.allclose(a["qpos"][0, :2], 0.0) and np.allclose(a["qpos"][5, :2], store[2]["root_qpos"][5, :2] - store[2]["root_qpos"][0, :2])
    assert np.allclose(a["joint_quat"], store[2]["joint_quat"][:120]) and np.allclose(a["root2site"], store[2]["root2site"][:120])
    # inference mode: a root-only trajectory, constant reward (walk_imitation.py:148-151)
    T = 90
    q = np.zeros((T, 7)); q[:, 2] = view.m.qpos0[2]; q[:, 3] = 1.0; q[:, 0] = 0.002 * np.arange(T)
    snip = W.inference_snippet(q, np.zeros((T, 6)))
    m = O.OracleModel(BLOB)
    e = O.OracleWalkEnv(m, W.WalkRefSet([snip]), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), (view.retract_qadr, view.retract_val),
                        inference_mode=True)
    e.reset()
    st, r, d, obs = e.step(np.zeros(e.naction))
    assert st == 1 and r == 1.0 and d == 1.0
