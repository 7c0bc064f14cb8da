"""Device n-step transition writer (`ffe_nstep_*`) against a numpy restatement of acme's NStepTransitionAdder semantics
(as used at agents/ray_distributed_dmpo.py:514-521; acme is not in the reference tree: parity unpinned, stated in
include/flybody_env.h) on scripted reward / discount / step_type sequences with LAST -> FIRST boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference(obs, act, rew, disc, st, n, gamma):
    """One env.  obs[t], rew[t], disc[t], st[t] = timestep t (t = 0 is FIRST); act[t] = action applied to reach timestep t.
    Returns the transitions in the order the adder writes them."""
    out, hist = [], []   # hist entries: (o_s, a_s, r_{s+1}, d_{s+1})
    last = None
    for t in range(len(st)):
        if st[t] == 0:
            hist, last = [], obs[t]
            continue
        hist.append((last, act[t], np.float32(rew[t]), np.float32(disc[t])))
        hist = hist[-n:]
        # acme's NStepTransitionAdder._write runs on every add() and does not wait for n entries: during an episode's first n - 1
        # steps it writes the short transitions (o_0 -> o_1), (o_0 -> o_2), ...; _write_last then flushes the tails
        starts = [0]
        if st[t] == 2:
            starts += list(range(1, len(hist)))
        for s in starts:
            ret, td = hist[s][2], hist[s][3]
            for i in range(s + 1, len(hist)):
                td = np.float32(td * np.float32(gamma))
                ret = np.float32(ret + np.float32(hist[i][2] * td))
                td = np.float32(td * hist[i][3])
            out.append((hist[s][0], hist[s][1], ret, td, obs[t]))
        last = obs[t]
    return out


@pytest.mark.parametrize("n_step", [1, 5, 50])
def test_nstep_writer_matches_restatement(n_step):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flybody_amd.actor_loop import NStepTransitionWriter
    from flybody_amd.dm_types import TimeStep

    B, O, A, T, gamma = 7, 11, 3, 160, 0.95
    rng = np.random.RandomState(n_step)
    obs = rng.randn(T, B, O).astype(np.float32)
    act = rng.randn(T, B, A).astype(np.float32)
    rew = rng.rand(T, B).astype(np.float32)
    disc = (rng.rand(T, B) > 0.1).astype(np.float32)
    st = np.ones((T, B), np.int32)
    st[0] = 0
    for b in range(B):       # episodes of different lengths per env, LAST followed by FIRST
        t = 0
        while True:
            t += rng.randint(2, 70)
            if t + 1 >= T:
                break
            st[t, b], st[t + 1, b] = 2, 0
            t += 1
    w = NStepTransitionWriter(B, O, A, n_step=n_step, discount=gamma, capacity=4096)
    dev = lambda x: torch.tensor(x, device="cuda")
    for t in range(T):
        ts = TimeStep(dev(st[t]), dev(rew[t]), dev(disc[t]), None)
        w.observe(dev(act[t]), ts, dev(obs[t]))
    o, a, r, d, o2 = [x.cpu().numpy() for x in w.transitions()]
    ref = [tr for b in range(B) for tr in _reference(obs[:, b], act[:, b], rew[:, b], disc[:, b], st[:, b], n_step, gamma)]
    assert len(ref) == w.num_written() == len(r) > 0
    # envs write concurrently, so slots interleave: compare as multisets keyed by (obs row, next-obs row), which are unique here
    key = lambda oo, nn: (oo.tobytes(), nn.tobytes())
    got = {key(o[i], o2[i]): (a[i], r[i], d[i]) for i in range(len(r))}
    assert len(got) == len(ref)
    for oo, aa, rr, dd, nn in ref:
        ga, gr, gd = got[key(oo, nn)]
        assert np.array_equal(ga, aa) and gr == rr and gd == dd     # same float32 operation order: bit for bit
    w.close()


def test_actor_loop_feeds_the_writer(torch_mod=None):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flybody_amd import fly_envs
    from flybody_amd.actor_loop import BatchedActorLoop, NStepTransitionWriter

    B = 64
    env = fly_envs.flight_imitation(batch_size=B, random_state=0)
    w = NStepTransitionWriter(B, env.spec.obs_dim, env.spec.action_dim, n_step=50, discount=0.99, capacity=B * 400)
    lo, hi = (torch.tensor(x, device="cuda") for x in env.raw_action_bounds())
    g = torch.Generator(device="cuda").manual_seed(0)
    loop = BatchedActorLoop(env, lambda obs: (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)), adder=w)
    stats = loop.run(300)
    n = w.num_written()
    # every MID / LAST step of an episode becomes the start of exactly one transition once its window closes
    o, a, r, d, o2 = w.transitions()
    assert stats["episodes"] > 0 and n > B * 100 and torch.isfinite(r).all() and (d >= 0).all() and (d <= 1).all()
    assert float(r.max()) <= 50.0 and float(r.min()) >= 0.0   # 50 rewards in [0, 1]
    w.close(); env.close()


def test_fused_timestep_pack_matches_torch():
    """flybody_amd.distributed.TimestepGather's fused pack (ffe_pack_timestep) == the four torch slice assignments."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flybody_amd.distributed import TimestepGather

    B, O = 1000, 104
    g = torch.Generator(device="cuda").manual_seed(0)
    obs = torch.rand(B, O, device="cuda", generator=g)
    rew, disc = torch.rand(B, device="cuda", generator=g), (torch.rand(B, device="cuda", generator=g) > 0.1).float()
    st = torch.randint(0, 3, (B,), device="cuda", generator=g, dtype=torch.int32)
    tg = TimestepGather(B, O, torch.device("cuda", 0), world=1, rank=0)
    tg(obs, rew, disc, st)
    torch.cuda.synchronize()
    uo, ur, ud, us = TimestepGather.unpack(tg.pack, O)
    assert torch.equal(uo, obs) and torch.equal(ur, rew) and torch.equal(ud, disc) and torch.equal(us, st)
