"""Multi-GPU plumbing for the env-sharded layout.

Envs are independent (the reference runs them as separate OS processes, `train_dmpo_ray.py:436-452`), so rank r of
W simply owns global envs [r*B, (r+1)*B); the only collective on the path is the per-step gather of what a central
learner consumes - (observation, reward, discount, step_type) - to rank 0.  Backend `nccl` is RCCL on ROCm; the
same code runs on `gloo` for the CPU tests.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard(rank: int, world: int, envs_per_rank: int) -> tuple[int, int]:
    """(env_id_base, batch) of this rank: contiguous blocks, env i lives on rank i // envs_per_rank."""
    return rank * envs_per_rank, envs_per_rank


class TimestepGather:
    """Packs a batched TimeStep into one [B, obs_dim + 3] float32 buffer and gathers it to rank 0 in ONE collective
    (reward, discount and step_type ride in the last three columns), so a step costs a single RCCL call."""

    def __init__(self, batch: int, obs_dim: int, device, world: int, rank: int, force_collective: bool = False):
        """`force_collective`: issue the collective on a one-rank group as well (the RCCL rehearsal of tests/test_gpu_rccl.py:
        communicator init, the gather on RCCL's stream, the async work handle - everything but the wire)."""
        self.world, self.rank, self.obs_dim = world, rank, obs_dim
        self.collective = world > 1 or force_collective
        self.pack = torch.empty(batch, obs_dim + 3, dtype=torch.float32, device=device)
        # gloo gathers host tensors only: a device-resident env on a gloo group (CPU-side learner, or the one-GPU rehearsal of the
        # sharded path in tests/test_gpu_sharded.py) stages the packed buffer through pinned host memory; RCCL takes it as it is
        self.stage = None
        if self.collective and self.pack.is_cuda and dist.get_backend() == "gloo":
            self.stage = torch.empty(batch, obs_dim + 3, dtype=torch.float32, pin_memory=True)
        like = self.stage if self.stage is not None else self.pack
        self.out = [torch.empty_like(like) for _ in range(world)] if (self.collective and rank == 0) else None

    def __call__(self, obs, reward, discount, step_type, async_op: bool = False):
        """Packs and gathers.  With `async_op` the collective runs on RCCL's own stream and the returned work handle
        must be waited on before this object is used again (double-buffer two of them to overlap the gather of step k
        with the physics of step k+1)."""
        p = self.pack
        if p.is_cuda and obs.is_contiguous() and obs.dtype == torch.float32:  # one fused launch (ffe_pack_timestep) instead of four
            from . import _capi

            with torch.cuda.device(p.device):
                rc = _capi.lib().ffe_pack_timestep(obs.data_ptr(), reward.data_ptr(), discount.data_ptr(), step_type.data_ptr(), p.data_ptr(),
                                                   p.shape[0], self.obs_dim, torch.cuda.current_stream(p.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("ffe_pack_timestep failed")
        else:
            p[:, : self.obs_dim] = obs
            p[:, -3] = reward
            p[:, -2] = discount
            p[:, -1] = step_type.to(torch.float32)
        if self.collective:
            if self.stage is not None:
                self.stage.copy_(p)  # (synchronises the env's stream)
                p = self.stage
            work = dist.gather(p, self.out, dst=0, async_op=async_op)
            if async_op:
                return work
        return self.out if self.rank == 0 else None

    @staticmethod
    def unpack(buf, obs_dim: int):
        """Inverse of the packing for one rank's buffer."""
        return buf[:, :obs_dim], buf[:, -3], buf[:, -2], buf[:, -1].to(torch.int32)


class ActionScatter:
    """The other direction of the sharded step (the reference's actor applies `actor.select_action` next to
    its env, `agents/ray_distributed_dmpo.py:401-404` - with the policy on rank 0 the actions travel instead): rank 0 holds the
    actions of all W * B envs, rank r receives rows [r * B, (r + 1) * B) in ONE collective (8 192 x 12 floats = 0.39 MB per rank
    and step for flight).  Asynchronous like the gather: scatter step k + 1's actions while step k is simulated."""

    def __init__(self, batch: int, act_dim: int, device, world: int, rank: int, force_collective: bool = False):
        self.world, self.rank, self.batch, self.act_dim = world, rank, batch, act_dim
        self.collective = world > 1 or force_collective
        self.local = torch.empty(batch, act_dim, dtype=torch.float32, device=device)
        self.stage = None
        if self.collective and self.local.is_cuda and dist.get_backend() == "gloo":
            self.stage = torch.empty(batch, act_dim, dtype=torch.float32, pin_memory=True)  # (gloo moves host tensors only)

    def __call__(self, all_actions=None, async_op: bool = False):
        """`all_actions`: float32 [W * B, A] on rank 0 (ignored elsewhere).  Returns this rank's [B, A] block - or, with `async_op`,
        the work handle to wait on before reading `self.local`."""
        if not self.collective:
            self.local.copy_(all_actions)
            return self.local
        recv = self.stage if self.stage is not None else self.local
        chunks = None
        if self.rank == 0:
            assert tuple(all_actions.shape) == (self.world * self.batch, self.act_dim) and all_actions.dtype == torch.float32
            src = all_actions.cpu() if self.stage is not None else all_actions
            chunks = [c.contiguous() for c in src.chunk(self.world, dim=0)]
        work = dist.scatter(recv, chunks, src=0, async_op=async_op)
        if async_op:
            return _StagedWork(work, self) if self.stage is not None else work
        if self.stage is not None:
            self.local.copy_(self.stage)
        return self.local


class _StagedWork:
    """Work handle of a scatter that lands in pinned host memory: `wait` also moves it onto the device."""

    def __init__(self, work, owner):
        self.work, self.owner = work, owner

    def wait(self):
        self.work.wait()
        self.owner.local.copy_(self.owner.stage)
