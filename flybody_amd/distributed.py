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

    def __init__(self, batch: int, obs_dim: int, device, world: int, rank: int):
        self.world, self.rank, self.obs_dim = world, rank, obs_dim
        self.pack = torch.empty(batch, obs_dim + 3, dtype=torch.float32, device=device)
        # gloo gathers host tensors only: a device-resident env on a gloo group (CPU-side learner, or the one-GPU rehearsal of the
        # sharded path in tests/test_gpu_sharded.py) stages the packed buffer through pinned host memory; RCCL takes it as it is
        self.stage = None
        if world > 1 and self.pack.is_cuda and dist.get_backend() == "gloo":
            self.stage = torch.empty(batch, obs_dim + 3, dtype=torch.float32, pin_memory=True)
        like = self.stage if self.stage is not None else self.pack
        self.out = [torch.empty_like(like) for _ in range(world)] if (world > 1 and rank == 0) else None

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
        if self.world > 1:
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
