"""Asynchronous env groups on one GPU.

A launch of a whole batch ends with a drain: the batch is a whole number of rounds of resident waves, the launch ends with its slowest
wave, and in its last third most wave slots idle.  The reference never waits like that: its actors are separate
processes, each stepping its own env whenever its own policy call returns (`train_dmpo_ray.py:432-452`,
`agents/ray_distributed_dmpo.py:399-404`).  `EnvGroups` is that architecture on one device: the same envs as G handles of B / G envs, each
with its own HIP stream; a group's step is ordered only against that group's previous step, so one group's drain overlaps another
group's start.  Measured (tools/bench_pipelined.py): flight 8 192 envs, 10.9 -> 14.6 M env-steps/s with 2 groups; walk_on_ball 4 096 envs,
1.33 -> 1.45 M.

The consumer has to be group-aware to keep the overlap: everything it does with group g's timestep (policy, adder) goes on
`groups.streams[g]` - `with groups.on(g): ...` - and nothing waits for all groups at once.  Results do not depend on the grouping: envs are
independent and their random streams are keyed by the global env id (`env_id_base`), so G groups reproduce one handle bit for bit
(tests/test_gpu_groups.py).
"""
from __future__ import annotations

import contextlib


class EnvGroups:
    def __init__(self, factory, batch_size: int, groups: int = 2, **kwargs):
        """`factory`: `fly_envs.flight_imitation` / `fly_envs.walk_on_ball` (or anything with the same keywords); `batch_size`: envs in
        total; `kwargs` go to every group's factory call (flight: `env_id_base` is set per group on top of the one given)."""
        import torch

        if groups < 1 or batch_size % groups:
            raise ValueError("batch_size must be a multiple of groups")
        self.batch_size, self.n = batch_size, groups
        per = batch_size // groups
        base = int(kwargs.pop("env_id_base", 0))
        self.envs = []
        for g in range(groups):
            kw = dict(kwargs)
            if "flight" in getattr(factory, "__name__", ""):
                kw["env_id_base"] = base + g * per
            self.envs.append(factory(batch_size=per, **kw))
        dev = self.envs[0].device
        self.streams = [torch.cuda.Stream(dev) for _ in range(groups)]
        self.per_group = per
        self._torch = torch

    def on(self, g: int):
        """Context: torch's current stream = group g's stream."""
        return self._torch.cuda.stream(self.streams[g])

    def rows(self, g: int) -> slice:
        return slice(g * self.per_group, (g + 1) * self.per_group)

    def reset(self):
        out = []
        for g, e in enumerate(self.envs):
            with self.on(g):
                out.append(e.reset())
        return out

    def step(self, actions):
        """`actions`: one [batch_size, A] tensor (group g takes rows `rows(g)`; it must be complete before this call is made - it is read on
        the groups' streams without a dependency on the caller's) or a list of per-group [B / G, A] tensors produced on the groups' own streams.
        Returns the groups' TimeSteps; group g's is ordered on `streams[g]`."""
        per_group = isinstance(actions, (list, tuple))
        out = []
        for g, e in enumerate(self.envs):
            with self.on(g):
                out.append(e.step(actions[g] if per_group else actions[self.rows(g)]))
        return out

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for e in self.envs:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
