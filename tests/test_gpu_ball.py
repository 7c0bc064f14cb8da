"""GPU parity tests of the walk_on_ball HIP path (through the C ABI) against the float64 oracle on identical inputs.
Run with `-m gpu` on an MI355X.  Float32 tolerances are stated per test; the oracle's own pinning is in
tests/test_oracle_ball.py (the physics stays "parity unpinned" against MuJoCo itself)."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BALL_BLOB = os.path.join(ROOT, "flybody_amd", "assets", "fly_ball.ffmb")
# teacher-forced single control step, env-steps without a contact flip: max error per observation group, relative to the
# group's magnitude (set to <= 3x the measured maxima, see test_env_protocol_and_observation_parity)
# measured on MI355X (profiles/r02_gpu_tests.log): reward 2.1e-5, ball_qvel 1.5e-4, joints_vel 1.5e-4, force 5.5e-5, touch 3.5e-5,
# joints_pos 8.3e-7, appendages_pos 4.4e-7, actuator_activation 6.5e-8, the rest exact
TOL = {"reward": 6e-5, "accelerometer": 1e-6, "actuator_activation": 2e-7, "appendages_pos": 1.5e-6, "ball_qvel": 4.5e-4, "force": 1.7e-4,
       "gyro": 1e-6, "joints_pos": 2.5e-6, "joints_vel": 4.5e-4, "touch": 1.1e-4, "velocimeter": 1e-6, "world_zaxis": 1e-6}


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _oracle_states(n, amp, seed, settle=3):
    """States (qpos, qvel, act) sampled along an oracle rollout under random actions: realistic contact sets."""
    from oracle import oracle as O

    m = O.OracleModel(BALL_BLOB)
    env = O.OracleBallEnv(m)
    env.reset()
    rs = np.random.RandomState(seed)
    out = []
    for k in range(n):
        for _ in range(settle):
            env.step(rs.uniform(-amp, amp, 59))
        d = env.data
        out.append((d.qpos.copy(), d.qvel.copy(), d.act.copy()))
    return m, out


def _oracle_advance(m, state, ctrl, nsteps, flags=0):
    from oracle import oracle as O

    d = O.OracleData(m)
    m.set_flags(flags)
    d.qpos[:], d.qvel[:], d.act[:] = state
    d.ctrl[:] = ctrl
    d.step1()
    info = []
    for _ in range(nsteps):
        d.step2()
        info.append((d.ncon_matter, d.nefc, m.L.fo_solver_iter(d.ptr)))
        d.step1()
    m.set_flags(0)
    return d.qpos.copy(), d.qvel.copy(), d.act.copy(), info


def _gpu_advance(torch, states, ctrls, nsteps, flags=0):
    from flybody_amd.batched_env import BatchedBallEnv

    B = len(states)
    env = BatchedBallEnv(batch_size=B, physics_flags=flags)
    env.reset()
    qpos = torch.tensor(np.stack([s[0] for s in states]), dtype=torch.float64, device="cuda")
    qvel = torch.tensor(np.stack([s[1] for s in states]), dtype=torch.float64, device="cuda")
    act = torch.tensor(np.stack([s[2] for s in states]), dtype=torch.float64, device="cuda")
    env.set_state(qpos, qvel)
    env.set_act(act)
    env.physics_step(torch.tensor(np.stack(ctrls), dtype=torch.float32, device="cuda"), nsteps)
    q, v = env.get_state()
    a = env.get_act()
    ints, _ = env.get_task_state()
    torch.cuda.synchronize()
    out = q.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy(), ints.cpu().numpy()
    env.close()
    return out


def _report(tag, q, v, a, ref):
    eq = max(np.abs(q[i] - r[0]).max() for i, r in enumerate(ref))
    ev = max(np.abs(v[i] - r[1]).max() / max(1.0, np.abs(r[1]).max()) for i, r in enumerate(ref))
    ea = max(np.abs(a[i] - r[2]).max() for i, r in enumerate(ref))
    print(f"{tag}: qpos {eq:.3e} qvel(rel) {ev:.3e} act {ea:.3e}")
    return eq, ev, ea


@pytest.mark.parametrize("flags,name", [(64 | 2, "smooth"), (64, "limits"), (128 | 256, "contacts"), (128, "adhesion"), (0, "full")])
def test_one_substep_teacher_forced(torch_mod, flags, name):
    """One physics substep from oracle-sampled states: smooth dynamics only -> + joint limits -> + ball contacts
    (elliptic Newton) -> + adhesion -> + noslip."""
    m, states = _oracle_states(24, 0.6, seed=5)
    rs = np.random.RandomState(11)
    ctrls = [rs.uniform(-0.5, 0.5, 59).astype(np.float32) for _ in states]
    ref = [_oracle_advance(m, s, c.astype(np.float64), 1, flags) for s, c in zip(states, ctrls)]
    q, v, a, ints = _gpu_advance(torch_mod, states, ctrls, 1, flags)
    eq, ev, ea = _report(name, q, v, a, ref)
    if flags == 0:
        print("contacts oracle", [r[3][0][0] for r in ref], "gpu", ints[:, 5].tolist(), "iters oracle", [r[3][0][2] for r in ref], "gpu", ints[:, 6].tolist())
    assert ea < 1e-6
    assert eq < 2e-6, name   # qpos moves by h * qvel: 2e-4 * O(10)
    assert ev < 1.5e-4, name   # one substep of contact forces at float32 (measured 5e-5 with noslip, 3e-6 without)


def test_one_substep_at_saturated_actions(torch_mod):
    """The same comparison in the regime a random policy produces (every actuator driven over its whole range: several joints on
    their limits, legs pressed into the ball, the fly's own geoms in contact).  No state may exceed the kernel's contact / row /
    column capacities (16 contacts, 48 rows, 16 rows per block of M): all 32 are compared."""
    m, states = _oracle_states(32, 1.0, seed=9, settle=6)
    rs = np.random.RandomState(21)
    ctrls = [rs.uniform(-1.0, 1.0, 59).astype(np.float32) for _ in states]
    ref = [_oracle_advance(m, s, c.astype(np.float64), 1) for s, c in zip(states, ctrls)]
    q, v, a, ints = _gpu_advance(torch_mod, states, ctrls, 1)
    assert np.isfinite(q).all() and np.isfinite(v).all()
    rows = [r[3][0][1] for r in ref]
    print(f"saturated: overflow flags {ints[:, 7].tolist()}; oracle constraint rows min/mean/max {min(rows)}/{np.mean(rows):.1f}/{max(rows)}, "
          f"contacts max {max(r[3][0][0] for r in ref)}")
    assert (ints[:, 7] == 0).all()
    eq = max(np.abs(q[i] - ref[i][0]).max() for i in range(len(states)))
    ev = max(np.abs(v[i] - ref[i][1]).max() / max(1.0, np.abs(ref[i][1]).max()) for i in range(len(states)))
    print(f"saturated: qpos {eq:.3e} qvel(rel) {ev:.3e}")
    assert eq < 1e-6 and ev < 1.5e-4  # measured 2.2e-7 / 4.6e-5


def test_forced_capacity_overflow_stays_finite(torch_mod):
    """poses no policy reaches - every hinge drawn uniformly over its range, legs, wings and abdomen through each other - put far
    more geom pairs in contact than the tile's 16 slots.  The kernel flags those envs (task-state int 7, bit 0), keeps the deepest
    contacts (ball contacts, sphere / capsule pairs and convex pairs alike) and must stay finite through a whole control step."""
    from flybody_amd.model.blob import read_blob

    t = read_blob(BALL_BLOB)
    rng = np.random.RandomState(17)
    lo, hi = np.asarray(t["jnt_range"])[:, 0], np.asarray(t["jnt_range"])[:, 1]
    adr = np.asarray(t["jnt_qposadr"])
    hinge = [j for j in range(len(adr)) if int(t["jnt_type"][j]) == 3]
    m, base = _oracle_states(1, 0.2, seed=3)
    states = []
    for _ in range(48):
        q = base[0][0].copy()
        for j in hinge:
            if hi[j] > lo[j]:
                q[int(adr[j])] = rng.uniform(lo[j], hi[j])
        states.append((q, np.zeros_like(base[0][1]), base[0][2].copy()))
    ctrls = [rng.uniform(-1.0, 1.0, 59).astype(np.float32) for _ in states]
    q, v, a, ints = _gpu_advance(torch_mod, states, ctrls, 10)
    flagged = int((ints[:, 7] & 1).sum())
    print(f"forced overflow: {flagged} of {len(states)} envs flagged 'more than 16 contacts', contacts at the end of the step max {int(ints[:, 5].max())}")
    assert flagged >= 4, ints[:, 7].tolist()
    assert np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(a).all()
    assert np.abs(v).max() < 1e6


def test_ten_substeps_open_loop(torch_mod):
    m, states = _oracle_states(16, 0.4, seed=7)
    rs = np.random.RandomState(3)
    ctrls = [rs.uniform(-0.3, 0.3, 59).astype(np.float32) for _ in states]
    ref = [_oracle_advance(m, s, c.astype(np.float64), 10) for s, c in zip(states, ctrls)]
    q, v, a, _ = _gpu_advance(torch_mod, states, ctrls, 10)
    eq, ev, ea = _report("10 substeps", q, v, a, ref)
    assert eq < 2e-5 and ev < 1e-2 and ea < 1e-6


def _obs_groups():
    from oracle.oracle import OracleBallEnv

    out, o = {}, 0
    for name, n in OracleBallEnv.LAYOUT:
        out[name] = (o, o + n)
        o += n
    return out


def _gpu_contact_history(env, n=10):
    """Per substep of the last control step: contacts inside their includemargin (ffe_get_task_state columns 0-1)."""
    ints, _ = env.get_task_state()
    ints = ints.cpu().numpy().astype(np.int64)
    word = (ints[:, 0] & 0xffffffff) | ((ints[:, 1] & 0xffffffff) << 32)
    return np.stack([(word >> (4 * s)) & 15 for s in range(n)], axis=1)


def _gpu_detected_history(env, n=10):
    """Per substep of the last control step: contacts detected inside their margin, active or not (ffe_get_task_state real 0)."""
    _, reals = env.get_task_state()
    word = reals.cpu().numpy()[:, 0].astype(np.uint64)
    return np.stack([(word >> np.uint64(5 * s)) & np.uint64(31) for s in range(n)], axis=1).astype(np.int64)


def test_env_protocol_and_observation_parity(torch_mod):
    """reset + 30 control steps with identical random raw actions: FIRST / MID protocol, reward, every observation group.

    Contact make/break is a discontinuity of the time stepping (a claw whose distance crosses its includemargin one substep
    earlier gets one more substep of contact force), so an open-loop comparison of a contact-rich rollout amplifies float32
    rounding by orders of magnitude within a few control steps.  The oracle is therefore teacher-forced: before every
    control step it is put on the HIP env's state, and one full control step (10 substeps + sensors + reward) is compared.

    Every compared env-step is classified by its per-substep contact history (contacts inside their includemargin, from
    both sides): where the histories agree the step is held to float32 accuracy, with no allowance; where they differ (a
    *contact flip*) the oracle must show that some candidate pair sat within 2e-6 cm of its switching distance during that
    control step - otherwise the difference is a bug, not rounding - and the flips are counted and bounded."""
    from flybody_amd import fly_envs
    from oracle import oracle as O

    torch = torch_mod
    B = 8
    env = fly_envs.walk_on_ball(batch_size=B)
    m = O.OracleModel(BALL_BLOB)
    oenvs = [O.OracleBallEnv(m) for _ in range(B)]
    ts = env.reset()
    torch.cuda.synchronize()
    groups = _obs_groups()
    ref0 = [e.reset() for e in oenvs]
    assert (ts.step_type.cpu().numpy() == 0).all() and ref0[0][0] == 0
    obs = env.flat_observation.cpu().numpy()
    worst = {}
    for name, (lo, hi) in groups.items():
        err = max(np.abs(obs[i, lo:hi] - ref0[i][3][lo:hi]).max() / max(1.0, np.abs(ref0[i][3][lo:hi]).max()) for i in range(B))
        worst[name] = err
    print("reset obs errors", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-5
    rs = np.random.RandomState(0)
    werr = {k: [] for k in groups}   # non-flip env-steps
    rerr, flip_rerr, flip_gaps = [], [], []
    for t in range(30):
        q, v = env.get_state()
        ac = env.get_act()
        q, v, ac = q.cpu().numpy(), v.cpu().numpy(), ac.cpu().numpy()
        a = rs.uniform(-0.2, 0.2, (B, 59)) * (1.0 + 0.1 * t)
        ts = env.step(torch.tensor(a, dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
        obs = env.flat_observation.cpu().numpy()
        rew = ts.reward.cpu().numpy()
        ghist, gdet = _gpu_contact_history(env), _gpu_detected_history(env)
        for i, e in enumerate(oenvs):
            d = e.data
            d.qpos[:], d.qvel[:], d.act[:] = q[i], v[i], ac[i]
            d.step1()
            st, r, dsc, o = e.step(a[i].astype(np.float32).astype(np.float64))
            assert st == int(ts.step_type[i]) and dsc == float(ts.discount[i])
            ohist, ogap = e.contact_history()
            # (a detection inside the margin counts as well: an adhesion actuator pulls on every detected contact of its claw)
            if (ohist != ghist[i]).any() or (e.detected_history() != gdet[i]).any():
                flip_rerr.append(abs(r - rew[i])); flip_gaps.append(float(ogap.min()))
                continue
            rerr.append(abs(r - rew[i]))
            for name, (lo, hi) in groups.items():
                werr[name].append(np.abs(obs[i, lo:hi] - o[lo:hi]).max() / max(1.0, np.abs(o[lo:hi]).max()))
    mx = {k: float(np.max(v)) for k, v in werr.items()}
    n_all = len(rerr) + len(flip_rerr)
    print(f"30 teacher-forced control steps x 8 envs: {len(flip_rerr)} contact flips in {n_all} env-steps "
          f"(reward err on flips max {max(flip_rerr, default=0.0):.2e}, closest switching distance {max(flip_gaps, default=0.0):.2e} cm)")
    print("   non-flip steps: max obs errors", {k: f"{v:.2e}" for k, v in mx.items()}, "reward max %.2e" % np.max(rerr))
    assert len(flip_rerr) <= 0.05 * n_all
    assert all(g < 2e-6 for g in flip_gaps), flip_gaps      # a flip needs a pair at its switching distance (float32 rounding of ~0.5 cm positions)
    assert max(flip_rerr, default=0.0) < 2e-2               # one substep of one claw's contact force
    # no contact flip: float32 accuracy, max over all such env-steps (bounds = 3x the values measured on MI355X, profiles/r02_gpu_tests.log)
    assert np.max(rerr) < TOL["reward"]
    for name in groups:
        assert mx[name] < TOL[name], (name, mx[name])
    env.close()


def test_open_loop_rollout_statistics(torch_mod):
    """Open loop, 40 control steps: trajectories decorrelate at contact events (see above) but the ensemble must agree."""
    from flybody_amd import fly_envs
    from oracle import oracle as O

    torch = torch_mod
    B = 16
    env = fly_envs.walk_on_ball(batch_size=B)
    m = O.OracleModel(BALL_BLOB)
    oenvs = [O.OracleBallEnv(m) for _ in range(B)]
    env.reset()
    [e.reset() for e in oenvs]
    rs = np.random.RandomState(4)
    gr, orr, dq = [], [], []
    for t in range(40):
        a = rs.uniform(-0.2, 0.2, (B, 59))
        ts = env.step(torch.tensor(a, dtype=torch.float32, device="cuda"))
        gr.append(ts.reward.cpu().numpy().copy())
        orr.append([e.step(a[i].astype(np.float32).astype(np.float64))[1] for i, e in enumerate(oenvs)])
    q, _ = env.get_state()
    q = q.cpu().numpy()
    dq = max(np.abs(q[i, 4:] - e.data.qpos[4:]).max() for i, e in enumerate(oenvs))
    gr, orr = np.array(gr), np.array(orr)
    print("open loop: mean reward gpu %.5f oracle %.5f, max |dreward| %.2e, max joint angle difference %.2e rad" % (gr.mean(), orr.mean(), np.abs(gr - orr).max(), dq))
    assert abs(gr.mean() - orr.mean()) < 2e-3 and np.abs(gr - orr).max() < 0.05 and dq < 0.02
    env.close()


def test_time_limit_autoreset_and_determinism(torch_mod):
    from flybody_amd.batched_env import BatchedBallEnv

    torch = torch_mod
    env = BatchedBallEnv(batch_size=4, time_limit=0.01)  # 5 control steps: 50 float64 additions of 2e-4 reach 0.01
    assert env.spec.nsub == 10
    env.reset()
    a = torch.zeros(4, 59, dtype=torch.float32, device="cuda")
    types = []
    for _ in range(7):
        ts = env.step(a)
        types.append(ts.step_type.cpu().numpy().copy())
    types = np.array(types)
    assert (types[:4] == 1).all() and (types[4] == 2).all() and (types[5] == 0).all() and (types[6] == 1).all()
    o1 = env.flat_observation.cpu().numpy().copy()
    assert np.isfinite(o1).all() and np.abs(o1[0] - o1[3]).max() == 0.0  # identical envs stay bit-identical
    env.close()


def test_config3_batch_rollout_properties(torch_mod):
    """BASELINE configs[2] shape (walk_on_ball, contacts, batch 4096): finite, bounded, contact counts within capacity."""
    from flybody_amd import fly_envs

    torch = torch_mod
    B = 4096
    env = fly_envs.walk_on_ball(batch_size=B)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(20):
        a = (torch.rand(B, 59, device="cuda", generator=g) * 0.4 - 0.2).contiguous()
        ts = env.step(a)
    torch.cuda.synchronize()
    obs = env.flat_observation
    assert torch.isfinite(obs).all() and (ts.step_type == 1).all()
    ints, _ = env.get_task_state()
    assert int(ints[:, 5].max()) <= 16 and int(ints[:, 5].min()) >= 1
    r = ts.reward
    assert float(r.min()) >= 0.0 and float(r.max()) <= 1.0
    env.close()


def test_config1_single_env_1000_random_steps(torch_mod):
    """BASELINE configs[0]: walk_on_ball, 1 env, 1000 random-action control steps (raw U(-0.2, 0.2), `task_utils.py:13-24`).
    The episode runs to the 2.0 s time limit, which composer.Environment tests on MuJoCo's accumulated float64 time: 10 000
    additions of 2e-4 give 1.9999999999998, so the reference's episode has 1001 control steps - MID for 1000 steps, LAST
    (discount 1) on step 1001, FIRST on the next call."""
    from flybody_amd import fly_envs

    torch = torch_mod
    env = fly_envs.walk_on_ball(batch_size=1)
    ts = env.reset()
    assert int(ts.step_type[0]) == 0
    g = torch.Generator(device="cuda").manual_seed(0)
    total = 0.0
    for k in range(1001):
        ts = env.step((torch.rand(1, 59, device="cuda", generator=g) * 0.4 - 0.2).contiguous())
        st = int(ts.step_type[0])
        assert st == (2 if k == 1000 else 1), (k, st)
        total += float(ts.reward[0])
    assert float(ts.discount[0]) == 1.0 and torch.isfinite(env.flat_observation).all()
    assert 0.0 < total / 1000 < 1.0
    ts = env.step(torch.zeros(1, 59, device="cuda"))
    assert int(ts.step_type[0]) == 0 and float(ts.reward[0]) == 0.0
    ints, _ = env.get_task_state()
    assert int(ints[0, 7]) == 0  # no contact overflow
    env.close()


def test_full_episode_soak_batch_1024(torch_mod):
    """A whole 2.0 s episode (1001 control steps = 10 010 contact-solving substeps) of 1024 envs under random actions, across
    the time-limit reset: every observation finite, rewards in [0, 1], all envs reach LAST together at step 1001 and FIRST on
    the next call, contact counts within capacity, and the capacity-overflow flag (column 7 of the task state) never raised."""
    from flybody_amd import fly_envs

    torch = torch_mod
    B = 1024
    env = fly_envs.walk_on_ball(batch_size=B)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = [(torch.rand(B, 59, device="cuda", generator=g) * 0.4 - 0.2).contiguous() for _ in range(32)]
    rsum = torch.zeros(B, device="cuda")
    finite = torch.ones((), dtype=torch.bool, device="cuda")
    overflow = torch.zeros(B, dtype=torch.int32, device="cuda")
    maxcon = torch.zeros(B, dtype=torch.int32, device="cuda")
    for k in range(1001):
        ts = env.step(acts[k % 32])
        rsum += ts.reward
        finite &= torch.isfinite(env.flat_observation).all() & torch.isfinite(ts.reward).all()
        if k % 50 == 49 or k == 1000:
            ints, _ = env.get_task_state()
            overflow |= ints[:, 7]
            maxcon = torch.maximum(maxcon, ints[:, 5])
    assert bool(finite)
    assert (ts.step_type == 2).all() and (ts.discount == 1.0).all()
    assert float(rsum.min()) >= 0.0 and float(rsum.max()) <= 1001.0 and 0.0 < float(rsum.mean()) / 1001 < 1.0
    assert int(overflow.max()) == 0, f"{int((overflow != 0).sum())} envs overflowed the contact / row capacity"
    assert 1 <= int(maxcon.max()) <= 16
    ts = env.step(acts[0])
    assert (ts.step_type == 0).all()
    print(f"soak: mean reward {float(rsum.mean()) / 1000:.4f}, max contacts seen {int(maxcon.max())}")
    env.close()


def test_actor_loop_with_torch_policy(torch_mod):
    """The caller of the hot path: a batched on-device policy driving walk_on_ball through
    `BatchedActorLoop`; observations and actions never leave the GPU, episode statistics carry the reference's log keys."""
    from flybody_amd import fly_envs
    from flybody_amd.actor_loop import BatchedActorLoop

    torch = torch_mod
    B = 256
    env = fly_envs.walk_on_ball(batch_size=B)
    g = torch.Generator(device="cuda").manual_seed(0)
    W = 0.02 * torch.randn(289, 59, device="cuda", generator=g)
    lo, hi = env.raw_action_bounds()
    lo_t, hi_t = torch.tensor(lo, device="cuda"), torch.tensor(hi, device="cuda")

    def policy(obs):
        return torch.maximum(torch.minimum(torch.tanh(obs @ W) * 0.2, hi_t), lo_t)

    loop = BatchedActorLoop(env, policy)
    stats = loop.run(1002)  # one full 2.0 s episode (1001 control steps) of every env plus the first step of the next
    assert stats["episodes"] == B and abs(stats["episode_length"] - 1001.0) < 1e-6
    assert 0.0 <= stats["episode_return"] <= 1001.0 and stats["steps_per_second"] > 0
    env.close()


def test_canonical_action_wrapper_folded_in(torch_mod):
    """walk_on_ball with canonical_actions=True == the raw env fed canonical2real(a) (tasks/task_utils.py:53-76; acme
    CanonicalSpecWrapper as applied at train_dmpo_ray.py:128-129); the spec then spans [-1, 1]."""
    from flybody_amd.batched_env import BatchedBallEnv

    torch = torch_mod
    B = 16
    raw = BatchedBallEnv(batch_size=B)
    can = BatchedBallEnv(batch_size=B, canonical_actions=True, clip_actions=True)
    spec = can.action_spec()
    assert (spec.minimum == -1).all() and (spec.maximum == 1).all() and spec.shape == (59,)
    lo, hi = raw.raw_action_bounds()
    assert (hi > lo).all()
    raw.reset(); can.reset()
    rng = np.random.RandomState(0)
    for _ in range(5):
        a = rng.uniform(-1.3, 1.3, (B, 59)).astype(np.float32)              # some entries need the clip
        real = (lo + np.float32(0.5) * (np.clip(a, -1, 1) + np.float32(1)) * (hi - lo)).astype(np.float32)   # canonical2real in float32
        t0 = raw.step(torch.tensor(real, device="cuda"))
        o0 = raw.flat_observation.clone()
        t1 = can.step(torch.tensor(a, device="cuda"))
        # the kernel contracts the map into FMAs, so the controls agree to an ulp, not bit for bit
        lo_, hi_ = _obs_groups()["actuator_activation"]
        assert torch.allclose(o0[:, lo_:hi_], can.flat_observation[:, lo_:hi_], rtol=0, atol=1e-6)
        lo_, hi_ = _obs_groups()["joints_pos"]
        assert torch.allclose(o0[:, lo_:hi_], can.flat_observation[:, lo_:hi_], rtol=0, atol=1e-5)
        assert torch.allclose(t0.reward, t1.reward, atol=1e-4) and torch.equal(t0.step_type, t1.step_type)
    raw.close(); can.close()


def test_reset_envs_restarts_a_subset_only(torch_mod):
    from flybody_amd.batched_env import BatchedBallEnv

    torch = torch_mod
    env = BatchedBallEnv(batch_size=6)
    first = env.reset()
    obs_first = env.flat_observation.clone()
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(4):
        ts = env.step((torch.rand(6, 59, device="cuda", generator=g) * 0.4 - 0.2).contiguous())
    before = env.flat_observation.clone()
    mask = torch.tensor([0, 1, 0, 0, 1, 0], dtype=torch.bool, device="cuda")
    ts = env.reset_envs(mask)
    assert torch.equal(env.flat_observation[~mask], before[~mask]) and torch.equal(env.flat_observation[mask], obs_first[mask])
    assert ts.step_type.tolist() == [1, 0, 1, 1, 0, 1]
    ints, _ = env.get_task_state()
    assert ints[:, 2].tolist() == [4, 0, 4, 4, 0, 4]
    env.close()


# ---------------------------------------------------------------------------------------------- fly-fly contacts
import functools


@functools.lru_cache(maxsize=4)
def _fly_fly_states(n, seed=0, max_depth=0.003):
    """Random joint poses (inside the joint ranges) in which the fly's own geoms touch: legs against legs, mouth parts against
    front legs, abdomen tip against hind tarsi, claws (margin + gap, adhesion) - sphere / capsule pairs - and pairs with an
    ellipsoid or a cylinder on one side (femur / tibia / coxa on an abdomen segment, legs on the thorax or head, ...: the general
    convex collider).  Found with the oracle; returns [(qpos, qvel, act, names of the touching pairs)] with shallow penetrations only.
    (The labrum halves and the haustellum on the head touch in every pose; a pose qualifies by a pair beyond those.)"""
    import json

    from flybody_amd.model.blob import read_blob
    from oracle import oracle as O

    t = read_blob(BALL_BLOB)
    meta = json.load(open(BALL_BLOB.replace(".ffmb", ".json")))
    names, jname = meta["geom_name"], meta["jnt_name"]
    gtype = np.asarray(t["geom_type"])
    m = O.OracleModel(BALL_BLOB)
    d = O.OracleData(m)
    rng = np.random.RandomState(seed)
    hinge = [j for j in range(len(t["jnt_type"])) if t["jnt_type"][j] == 3]
    cand = {"claw": [], "mouth": [], "convex": [], "other": []}
    lo_, hi_ = t["jnt_range"][hinge].T
    qa = t["jnt_qposadr"][hinge]
    always = {("labrum_left_lower_collision", "labrum_right_lower_collision"), ("haustellum_collision", "head_collision")}

    def pose(dq, amp):
        q = t["qpos0"].copy()
        q[qa] = np.clip(q[qa] + amp * dq * (hi_ - lo_) / 2, lo_, hi_)
        d.qpos[:] = q; d.qvel[:] = 0; d.act[:] = 0; d.ctrl[:] = 0
        d.forward()
        c = d.contacts()
        sc = [r for r in c if "ball" not in names[int(r[0])] and "ball" not in names[int(r[1])]]
        return q, c, sc, [r for r in sc if (names[int(r[0])], names[int(r[1])]) not in always]

    for _ in range(4000):
        if all(len(cand[k]) >= n // 4 for k in ("claw", "mouth", "convex")) and len(cand["other"]) >= n:
            break
        # a random direction in joint space, scaled up until the first fly-fly pair touches (bisection), then a little further:
        # a shallow contact, as a simulation would meet it
        dq = rng.uniform(-1, 1, len(hinge))
        # move two legs plus (sometimes) the head / mouth parts and the abdomen; the other legs keep standing on the ball
        moving = list(rng.choice(["T1_left", "T1_right", "T2_left", "T2_right", "T3_left", "T3_right"], 2, replace=False))
        moving += [x for x in ("head", "rostrum", "haustellum", "labrum", "antenna") if rng.rand() < 0.5] + (["abdomen"] if rng.rand() < 0.5 else [])
        dq *= np.array([any(k in jname[j] for k in moving) for j in hinge], dtype=float)
        a0, a1 = 0.0, 1.0
        if not pose(dq, a1)[3]:
            continue
        for _ in range(14):
            am = 0.5 * (a0 + a1)
            if pose(dq, am)[3]:
                a1 = am
            else:
                a0 = am
        q, c, sc, new = pose(dq, a1 + rng.uniform(0.002, 0.02))
        if not new or len(c) > 14 or d.nefc > 40 or any(r[5] < -max_depth for r in sc) or any(r[5] < -0.02 for r in c):
            continue
        pairs = [(names[int(r[0])], names[int(r[1])]) for r in sc]
        newp = [(names[int(r[0])], names[int(r[1])]) for r in new]
        kind = "claw" if any("claw" in x or "claw" in y for x, y in newp) else (
            "convex" if any(gtype[int(r[0])] >= 4 or gtype[int(r[1])] >= 4 for r in new) else (
                "mouth" if any(x.split("_")[0] in ("rostrum", "haustellum", "antenna") for x, y in newp) else "other"))
        cand[kind].append((q.copy(), rng.randn(m.nv) * 2.0, rng.uniform(-0.3, 0.3, m.na), pairs))
    out = cand["claw"][: n // 4] + cand["mouth"][: n // 4] + cand["convex"][: n // 4]   # the rarer kinds first, the rest legs against legs / abdomen tip
    out += cand["other"][: n - len(out)]
    assert len(out) == n, {k: len(v) for k, v in cand.items()}
    return m, names, out


@pytest.mark.parametrize("flags,name", [(128 | 256, "contacts"), (0, "full")])
def test_fly_fly_contacts_one_substep(torch_mod, flags, name):
    """a17: the fly's own pairs - sphere / capsule and general convex (condim 1, fruitfly.xml:16-25; excludes fruitfly.xml:733-760 +
    walk_on_ball.py:33-40) - collide on the GPU as in the oracle.  States with legs crossing, mouth parts on the front legs,
    the abdomen tip on the hind tarsi and claws (margin / gap / adhesion sharing) are put into both through set_state; after
    one physics substep the contact counts agree and qvel matches at the tolerance of test_one_substep_teacher_forced."""
    from oracle import oracle as O

    m, names, states = _fly_fly_states(32, seed=3)
    rs = np.random.RandomState(5)
    ctrls = [rs.uniform(-0.5, 0.5, 59).astype(np.float32) for _ in states]
    ref, nself = [], []
    d = O.OracleData(m)
    for s_, c_ in zip(states, ctrls):
        m.set_flags(flags)
        d.qpos[:], d.qvel[:], d.act[:] = s_[:3]
        d.ctrl[:] = c_
        d.step1()
        con = d.contacts()
        nself.append((len(con), sum(1 for r in con if "ball" not in names[int(r[0])] and "ball" not in names[int(r[1])])))
        d.step2()
        d.step1()
        ref.append((d.qpos.copy(), d.qvel.copy(), d.act.copy()))
        m.set_flags(0)
    # GPU: the contact set used by the substep is the one of the initial position stage, so count it with 0 substeps first
    q, v, a, ints = _gpu_advance(torch_mod, [s_[:3] for s_ in states], ctrls, 1, flags)
    eq, ev, ea = _report("fly-fly " + name, q, v, a, ref)
    kinds = sorted({(x.rsplit("_collision", 1)[0].split("_")[0], y.rsplit("_collision", 1)[0].split("_")[0]) for s_ in states for x, y in s_[3]})
    print("pair kinds covered:", kinds, " self-contact counts (oracle)", [n for _, n in nself], "overflow", ints[:, 7].tolist())
    assert max(n for _, n in nself) >= 2 and any("claw" in x or "claw" in y for s_ in states for x, y in s_[3])
    assert (ints[:, 7] == 0).all()
    assert ea < 1e-6 and eq < 2e-6, name
    assert ev < 3e-4, name   # (measured 2.4e-4 without noslip / adhesion, 1.0e-4 with: a claw on a wing blade - the contact position along a 0.003 cm thin ellipsoid is soft)


def _abdomen_on_ball_states(n, seed=0):
    """Poses in which the abdomen is bent down onto the ball (the ball's sphere against the abdomen's cylinders, mjc_SphereCylinder;
    condim 3 with friction, like the leg contacts): the abdomen's hinges are driven towards the ball until a segment touches, the legs
    are moved a little at random."""
    import json

    from flybody_amd.model.blob import read_blob
    from oracle import oracle as O

    t = read_blob(BALL_BLOB)
    meta = json.load(open(BALL_BLOB.replace(".ffmb", ".json")))
    names, jname = meta["geom_name"], meta["jnt_name"]
    m = O.OracleModel(BALL_BLOB)
    d = O.OracleData(m)
    rng = np.random.RandomState(seed)
    hinge = [j for j in range(len(t["jnt_type"])) if t["jnt_type"][j] == 3]
    lo_, hi_ = t["jnt_range"][hinge].T
    qa = t["jnt_qposadr"][hinge]
    abd = np.array(["abdomen" in jname[j] and "abduct" not in jname[j] for j in hinge])
    tip = np.array([jname[j] == "abdomen_7" for j in hinge])
    out = []

    def on_ball():
        return [r for r in d.contacts() if names[int(r[0])] == "ball_geom" and "abdomen" in names[int(r[1])] and "abdomen_7" not in names[int(r[1])]]

    def pose(dirn, amp):
        q = t["qpos0"].copy()
        q[qa] = np.clip(q[qa] + amp * dirn * (hi_ - lo_) / 2, lo_, hi_)
        d.qpos[:] = q; d.qvel[:] = 0; d.act[:] = 0; d.ctrl[:] = 0
        d.forward()
        return q

    for _ in range(600):
        if len(out) >= n:
            break
        # the segments bend down (negative flexion) by random shares, the tip's own hinge up, so that a cylinder touches before the
        # tip's sphere is pressed in; the legs move a little at random
        dirn = rng.uniform(-0.02, 0.02, len(hinge)) * (~abd) - rng.uniform(0.3, 1.0, len(hinge)) * abd * (~tip) + rng.uniform(0.3, 1.0, len(hinge)) * tip
        pose(dirn, 1.0)
        if not on_ball():
            continue
        a0, a1 = 0.0, 1.0
        for _ in range(16):
            am = 0.5 * (a0 + a1)
            pose(dirn, am)
            if on_ball():
                a1 = am
            else:
                a0 = am
        q = pose(dirn, a1 + rng.uniform(0.001, 0.01))
        c = d.contacts()
        hit = on_ball()
        if not hit or len(c) > 16 or d.nefc > 46 or any(r[5] < -0.006 for r in c):  # (within the kernel's capacities: 16 contacts, 48 rows)
            continue
        out.append((q.copy(), rng.randn(m.nv) * 2.0, rng.uniform(-0.3, 0.3, m.na), [names[int(r[1])] for r in hit]))
    assert len(out) == n, len(out)
    return m, names, out


def test_abdomen_on_the_ball_one_substep(torch_mod):
    """a17: the ball against the abdomen's cylinders (and the abdomen tip's sphere).  The poses put an abdomen segment on the ball;
    after one substep the contact counts equal the oracle's and qvel matches at the tolerance of the leg contacts."""
    from oracle import oracle as O

    m, names, states = _abdomen_on_ball_states(16, seed=1)
    rs = np.random.RandomState(8)
    ctrls = [rs.uniform(-0.5, 0.5, 59).astype(np.float32) for _ in states]
    ref, ncon, on_ball = [], [], []
    d = O.OracleData(m)
    for s_, c_ in zip(states, ctrls):
        d.qpos[:], d.qvel[:], d.act[:] = s_[:3]
        d.ctrl[:] = c_
        d.step1()
        d.step2()
        d.step1()
        ncon.append(d.ncon_matter)  # contacts of the position stage after the substep (those that take part in something): what the kernel reports after its launch
        on_ball.append(sum(1 for r in d.contacts() if names[int(r[0])] == "ball_geom" and "abdomen" in names[int(r[1])]))
        ref.append((d.qpos.copy(), d.qvel.copy(), d.act.copy()))
    q, v, a, ints = _gpu_advance(torch_mod, [s_[:3] for s_ in states], ctrls, 1)
    eq, ev, ea = _report("abdomen on the ball", q, v, a, ref)
    segs = sorted({x for s_ in states for x in s_[3]})
    print("segments on the ball:", segs, " contacts oracle", ncon, "gpu", ints[:, 5].tolist(), "of them abdomen on ball", on_ball, "overflow", ints[:, 7].tolist())
    assert any("abdomen_" in x and x != "abdomen_7_collision" for x in segs) and min(on_ball) >= 1
    assert ints[:, 5].tolist() == ncon and (ints[:, 7] == 0).all()
    assert ea < 1e-6 and eq < 2e-6 and ev < 1.5e-4


def test_fly_fly_contact_counts_and_sensors(torch_mod):
    """Same states through the task layer: one control step (10 substeps, sensors, reward) teacher-forced from each state.
    The number of contacts the GPU found at the state (ffe_get_task_state) equals the oracle's, and where the per-substep
    contact history agrees, the touch / force sensors, joint velocities and the reward match at the tolerances of
    test_env_protocol_and_observation_parity."""
    from flybody_amd import fly_envs
    from oracle import oracle as O

    torch = torch_mod
    m, names, states = _fly_fly_states(16, seed=7, max_depth=0.0015)
    B = len(states)
    env = fly_envs.walk_on_ball(batch_size=B)
    env.reset()
    env.set_state(torch.tensor(np.stack([s_[0] for s_ in states])), torch.tensor(np.stack([s_[1] for s_ in states])))
    env.set_act(torch.tensor(np.stack([s_[2] for s_ in states])))
    # contacts at the state itself: a zero-length physics step is not available, so read them after the control step's
    # first stage through the history instead (column 0-1), and the totals of the final state through columns 3 and 5
    rs = np.random.RandomState(2)
    a = rs.uniform(-0.2, 0.2, (B, 59)).astype(np.float32)
    ts = env.step(torch.tensor(a, device="cuda"))
    obs = env.flat_observation.cpu().numpy()
    rew = ts.reward.cpu().numpy()
    ghist = _gpu_contact_history(env)
    ints, _ = env.get_task_state()
    ints = ints.cpu().numpy()
    groups = _obs_groups()
    oenv = O.OracleBallEnv(m)
    oenv.reset()
    n_ok, n_flip = 0, 0
    worst = {k: 0.0 for k in groups}
    worst_r = 0.0
    for i, s_ in enumerate(states):
        d = oenv.data
        d.qpos[:], d.qvel[:], d.act[:] = s_[:3]
        d.step1()
        st, r, dsc, o = oenv.step(a[i].astype(np.float64))
        ohist, ogap = oenv.contact_history()
        assert ohist[0] == ghist[i, 0], (i, ohist, ghist[i])      # the contact set of the given state itself
        if (ohist != ghist[i]).any():
            n_flip += 1
            assert ogap.min() < 2e-6
            continue
        n_ok += 1
        # (the kernel keeps the contacts that take part in something: active, or with an adhesion actuator - a claw - on either body)
        c = [row for row in d.contacts() if int(row[3]) == 0 or "claw" in names[int(row[0])] or "claw" in names[int(row[1])]]
        assert len(c) == d.ncon_matter == ints[i, 5] and sum(1 for row in c if "ball" not in names[int(row[0])] and "ball" not in names[int(row[1])]) == ints[i, 3]
        worst_r = max(worst_r, abs(r - rew[i]))
        for name, (lo, hi) in groups.items():
            worst[name] = max(worst[name], np.abs(obs[i, lo:hi] - o[lo:hi]).max() / max(1.0, np.abs(o[lo:hi]).max()))
    print(f"fly-fly states, one control step: {n_ok} compared, {n_flip} contact flips; reward {worst_r:.2e}", {k: f"{v:.2e}" for k, v in worst.items()})
    assert n_ok >= B // 2
    assert worst_r < TOL["reward"]
    for name in groups:
        assert worst[name] < 3 * TOL[name], (name, worst[name])
    env.close()
