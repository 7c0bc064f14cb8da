"""Asynchronous env groups (flybody_amd/groups.py): the same envs as G handles on G streams reproduce one handle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_groups_reproduce_the_single_handle():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flybody_amd import fly_envs
    from flybody_amd.groups import EnvGroups

    B = 512
    one = fly_envs.flight_imitation(batch_size=B, random_state=0)
    grp = EnvGroups(fly_envs.flight_imitation, B, groups=2, random_state=0)
    spec = one.action_spec()
    lo, hi = torch.tensor(spec.minimum, device="cuda"), torch.tensor(spec.maximum, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    one.reset(); grp.reset()
    last = 0
    for k in range(160):
        a = (lo + (hi - lo) * torch.rand(B, 12, device="cuda", generator=g)).contiguous()
        torch.cuda.synchronize()  # (the actions are complete before the groups read them on their own streams)
        ts = one.step(a)
        tg = grp.step(a)
        grp.synchronize(); torch.cuda.synchronize()
        obs = one.flat_observation
        for i, (t2, e) in enumerate(zip(tg, grp.envs)):
            r = grp.rows(i)
            assert torch.equal(t2.reward, ts.reward[r]) and torch.equal(t2.step_type, ts.step_type[r]) and torch.equal(t2.discount, ts.discount[r])
            assert torch.equal(e.flat_observation, obs[r])
        last += int((ts.step_type == 2).sum())
    assert last > 100  # episodes rolled over (resets, new trajectories) inside the comparison
    one.close(); grp.close()


def test_grouped_actor_loop_and_ball_groups():
    """walk_on_ball: two groups of 128 envs against one handle of 256 (bit-identical: the task draws nothing at random), and the grouped
    actor loop's bookkeeping (episodes, transitions) against the single-handle loop's over the same number of steps."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flybody_amd import fly_envs
    from flybody_amd.actor_loop import BatchedActorLoop, GroupedActorLoop, NStepTransitionWriter
    from flybody_amd.groups import EnvGroups

    B = 256
    one = fly_envs.walk_on_ball(batch_size=B)
    grp = EnvGroups(fly_envs.walk_on_ball, B, groups=2)
    g = torch.Generator(device="cuda").manual_seed(2)
    one.reset(); grp.reset()
    for k in range(12):
        a = ((torch.rand(B, 59, device="cuda", generator=g) * 2 - 1) * 0.3).contiguous()
        torch.cuda.synchronize()
        ts = one.step(a); tg = grp.step(a)
        grp.synchronize(); torch.cuda.synchronize()
        for i, e in enumerate(grp.envs):
            assert torch.equal(tg[i].reward, ts.reward[grp.rows(i)]) and torch.equal(e.flat_observation, one.flat_observation[grp.rows(i)])
    one.close(); grp.close()

    # flight: the policy is a fixed function of the observation, so both loops see the same episodes
    Bf = 256
    pol = lambda o: torch.tanh(o[:, :12].contiguous() * 0.01)
    env = fly_envs.flight_imitation(batch_size=Bf, random_state=0, canonical_actions=True, clip_actions=True)
    ad = NStepTransitionWriter(Bf, env.spec.obs_dim, env.spec.action_dim, n_step=5, discount=0.99, capacity=1 << 16)
    r1 = BatchedActorLoop(env, pol, ad).run(150)
    n1 = ad.num_written()
    ad.close(); env.close()
    groups = EnvGroups(fly_envs.flight_imitation, Bf, groups=2, random_state=0, canonical_actions=True, clip_actions=True)
    ads = [NStepTransitionWriter(Bf // 2, e.spec.obs_dim, e.spec.action_dim, n_step=5, discount=0.99, capacity=1 << 16) for e in groups.envs]
    r2 = GroupedActorLoop(groups, pol, ads).run(150)
    n2 = sum(a.num_written() for a in ads)
    for a in ads:
        a.close()
    groups.close()
    assert r1["episodes"] == r2["episodes"] and n1 == n2 and abs(r1["episode_return"] - r2["episode_return"]) < 1e-9 * max(1.0, abs(r1["episode_return"]))
