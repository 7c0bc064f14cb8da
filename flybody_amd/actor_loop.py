"""Batched actor loop: the on-device counterpart of the reference's per-process acme loop
(`agents/ray_distributed_dmpo.py:401-440` `EnvironmentLoop.run_episode`, `agents/actors.py:59-101`).

The reference steps one env per OS process and calls its policy with a batch of 1.  Here observations never leave the
GPU: the policy is any callable `flat_obs[B, O] -> action[B, A]` (e.g. a torch module), and per-env episode statistics
are accumulated with the same keys the reference logs (`episode_length`, `episode_return`, `steps_per_second`).
"""

from __future__ import annotations

import time


class BatchedActorLoop:
    def __init__(self, env, policy):
        import torch

        self._t, self.env, self.policy = torch, env, policy
        B, dev = env.batch_size, env.device
        self._ret = torch.zeros(B, device=dev)
        self._len = torch.zeros(B, dtype=torch.int64, device=dev)
        self.finished_returns, self.finished_lengths = [], []

    @property
    def episodes(self) -> int:
        return sum(len(x) for x in self.finished_returns)

    def run(self, num_steps: int) -> dict:
        """Steps every env `num_steps` times (episodes roll over through the env's auto-reset)."""
        t = self._t
        ts = self.env.reset()
        self._ret.zero_(); self._len.zero_()
        start = time.perf_counter()
        for _ in range(num_steps):
            with t.no_grad():
                action = self.policy(self.env.flat_observation)
            ts = self.env.step(action.contiguous())
            mid_or_last = ts.step_type != 0
            self._ret += t.where(mid_or_last, ts.reward, t.zeros_like(ts.reward))
            self._len += mid_or_last.to(t.int64)
            done = ts.step_type == 2
            if bool(done.any()):
                self.finished_returns.append(self._ret[done].clone()); self.finished_lengths.append(self._len[done].clone())
                self._ret[done] = 0; self._len[done] = 0
        t.cuda.synchronize(self.env.device)
        wall = time.perf_counter() - start
        rets = t.cat(self.finished_returns) if self.finished_returns else t.zeros(0, device=self.env.device)
        lens = t.cat(self.finished_lengths) if self.finished_lengths else t.zeros(0, dtype=t.int64, device=self.env.device)
        return {"episodes": int(rets.numel()), "episode_return": float(rets.mean()) if rets.numel() else float("nan"),
                "episode_length": float(lens.float().mean()) if lens.numel() else float("nan"),
                "steps_per_second": num_steps * self.env.batch_size / wall}
