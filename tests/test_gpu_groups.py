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
