"""Batched actor loop: the on-device counterpart of the reference's per-process acme loop
(`agents/ray_distributed_dmpo.py:401-440` `EnvironmentLoop.run_episode`, `agents/actors.py:59-101`).

The reference steps one env per OS process and calls its policy with a batch of 1.  Here observations never leave the
GPU: the policy is any callable `flat_obs[B, O] -> action[B, A]` (e.g. a torch module), and per-env episode statistics
are accumulated with the same keys the reference logs (`episode_length`, `episode_return`, `steps_per_second`).
"""

from __future__ import annotations

import ctypes as C
import time

from . import _capi


class NStepTransitionWriter:
    """Device-resident batched n-step transition adder (`ffe_nstep_*`, flybody_amd/csrc/nstep.hip): what the reference's
    actors do one env at a time through `acme.adders.reverb.NStepTransitionAdder(n_step=50, discount=...)`
    (`agents/ray_distributed_dmpo.py:514-521`, `agents/actors.py:91-101`), for B envs per call, into a replay ring in HBM.

    `observe(action, timestep)` takes the action that was applied and the `TimeStep` the env returned for it (FIRST rows start
    an episode; their action is ignored).  `transitions()` returns views (obs, action, n_step_return, discount, next_obs) of the
    slots written so far; the learner applies one more factor of `discount` to the bootstrap value, as with acme."""

    def __init__(self, batch_size: int, obs_dim: int, act_dim: int, *, n_step: int = 50, discount: float = 0.99, capacity: int = 1 << 20,
                 device: int = 0):
        import torch

        self._t, self._L = torch, _capi.lib()
        self.batch_size, self.obs_dim, self.act_dim, self.n_step, self.discount, self.capacity = batch_size, obs_dim, act_dim, n_step, discount, capacity
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        if self._L.ffe_nstep_create(batch_size, obs_dim, act_dim, n_step, float(discount), capacity, device, C.byref(h)) != 0:
            raise RuntimeError("ffe_nstep_create: " + self._L.ffe_nstep_last_error(None).decode())
        self._h = h
        ptr = [C.c_void_p() for _ in range(6)]
        assert self._L.ffe_nstep_buffers(self._h, *[C.byref(p) for p in ptr]) == 0
        self._ptr = [p.value for p in ptr]

    def observe(self, action, timestep, flat_observation):
        t = self._t
        st = timestep.step_type
        # the C ABI takes raw device pointers: everything it will read is checked here (a float64 reward, a strided view or a host
        # tensor would otherwise be read as garbage, or fault)
        def ok(x, dtype, shape):
            return x.is_cuda and x.device == self.device and x.dtype == dtype and x.is_contiguous() and tuple(x.shape) == shape

        B = self.batch_size
        assert ok(action, t.float32, (B, self.act_dim)), "action: float32 [B, A] contiguous on the writer's device"
        assert ok(flat_observation, t.float32, (B, self.obs_dim)), "flat_observation: float32 [B, O] contiguous on the writer's device"
        assert ok(st, t.int32, (B,)) and ok(timestep.reward, t.float32, (B,)) and ok(timestep.discount, t.float32, (B,)), \
            "step_type int32 [B], reward / discount float32 [B], contiguous on the writer's device"
        stream = C.c_void_p(t.cuda.current_stream(self.device).cuda_stream)
        rc = self._L.ffe_nstep_observe(self._h, action.data_ptr(), st.data_ptr(), timestep.reward.data_ptr(), timestep.discount.data_ptr(),
                                       flat_observation.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError("ffe_nstep_observe: " + self._L.ffe_nstep_last_error(self._h).decode())

    def num_written(self) -> int:
        """Transitions written since creation (synchronises)."""
        import numpy as np

        t = self._t
        t.cuda.synchronize(self.device)
        out = np.zeros(1, dtype=np.uint64)
        # 8 bytes device -> host through a torch view of the counter
        cnt = self._view(self._ptr[5], (1,), t.int64)
        return int(cnt.cpu()[0])

    def _view(self, ptr, shape, dtype):
        """torch tensor over library-owned device memory (no copy), via the CUDA array interface."""
        t = self._t
        itemsize = {t.float32: 4, t.int64: 8}[dtype]
        typestr = {t.float32: "<f4", t.int64: "<i8"}[dtype]

        class _Mem:
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}

        del itemsize
        return t.as_tensor(_Mem(), device=self.device)

    def transitions(self):
        """(obs [N,O], action [N,A], n_step_return [N], discount [N], next_obs [N,O]) views of the N = min(written, capacity) filled slots.
        Slots are claimed before their rows are stored: a reader must be stream-ordered after the last `observe` (this method
        synchronises the device through `num_written`); after the ring has wrapped the slots are in no particular age order."""
        t, n = self._t, min(self.num_written(), self.capacity)
        o = self._view(self._ptr[0], (self.capacity, self.obs_dim), t.float32)[:n]
        a = self._view(self._ptr[1], (self.capacity, self.act_dim), t.float32)[:n]
        r = self._view(self._ptr[2], (self.capacity,), t.float32)[:n]
        d = self._view(self._ptr[3], (self.capacity,), t.float32)[:n]
        o2 = self._view(self._ptr[4], (self.capacity, self.obs_dim), t.float32)[:n]
        return o, a, r, d, o2

    def close(self):
        if getattr(self, "_h", None):
            self._L.ffe_nstep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedActorLoop:
    def __init__(self, env, policy, adder: NStepTransitionWriter | None = None):
        """`adder`: optional `NStepTransitionWriter`; fed as the reference's actor feeds its adder (`observe_first` on FIRST,
        `observe(action, next_timestep)` otherwise)."""
        import torch

        self._t, self.env, self.policy, self.adder = torch, env, policy, adder
        B, dev = env.batch_size, env.device
        self._ret = torch.zeros(B, device=dev)
        self._len = torch.zeros(B, dtype=torch.int64, device=dev)
        # episode statistics accumulate on the device: nothing in the loop reads a value back, so launches stay queued ahead
        self._tot = torch.zeros(2, dtype=torch.int64, device=dev)  # finished episodes, sum of their lengths
        self._sum_ret = torch.zeros(1, dtype=torch.float64, device=dev)
        self._L = _capi.lib()

    @property
    def episodes(self) -> int:
        return int(self._tot[0].item())

    def _iteration(self):
        t = self._t
        with t.no_grad():
            action = self.policy(self.env.flat_observation)
        action = action.contiguous()
        ts = self.env.step(action)
        if self.adder is not None:
            self.adder.observe(action, ts, self.env.flat_observation)
        # per-env return / length and the totals of finished episodes: one fused launch (ffe_episode_stats)
        with t.cuda.device(self.env.device):  # (ffe_episode_stats launches on the current device: make it the env's)
            rc = self._L.ffe_episode_stats(ts.step_type.data_ptr(), ts.reward.data_ptr(), self._ret.data_ptr(), self._len.data_ptr(), self._tot.data_ptr(),
                                           self._sum_ret.data_ptr(), self.env.batch_size, C.c_void_p(t.cuda.current_stream(self.env.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("ffe_episode_stats failed")

    def run(self, num_steps: int, graph: bool = False) -> dict:
        """Steps every env `num_steps` times (episodes roll over through the env's auto-reset).  With `graph` one iteration
        (policy, env step, adder, statistics) is captured once into a HIP graph and replayed, which takes the host out of the
        loop (measured neutral with the reference-shaped policy at B = 8192, where the GPU is the bound: tools/bench_actor_loop.py;
        it matters for small batches).  Needs a policy whose work is all on the
        env's device and free of host synchronisation; the env must not be double-buffered."""
        t = self._t
        self._begin()
        if graph:
            side = t.cuda.Stream(self.env.device)
            side.wait_stream(t.cuda.current_stream(self.env.device))
            with t.cuda.stream(side):
                for _ in range(3):
                    self._iteration()
            t.cuda.current_stream(self.env.device).wait_stream(side)
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g):
                self._iteration()
            t.cuda.synchronize(self.env.device)
            start = time.perf_counter()
            for _ in range(num_steps):
                g.replay()
            t.cuda.synchronize(self.env.device)
            wall = time.perf_counter() - start
            n = int(self._tot[0].item())
            return {"episodes": n, "episode_return": float(self._sum_ret.item()) / n if n else float("nan"),
                    "episode_length": float(self._tot[1].item()) / n if n else float("nan"),
                    "steps_per_second": num_steps * self.env.batch_size / wall, "capacity_flagged_envs": self._flagged()}
        start = time.perf_counter()
        for _ in range(num_steps):
            self._iteration()
        t.cuda.synchronize(self.env.device)
        wall = time.perf_counter() - start
        n = int(self._tot[0].item())
        return {"episodes": n, "episode_return": float(self._sum_ret.item()) / n if n else float("nan"),
                "episode_length": float(self._tot[1].item()) / n if n else float("nan"),
                "steps_per_second": num_steps * self.env.batch_size / wall, "capacity_flagged_envs": self._flagged()}

    def _begin(self):
        """reset + `observe_first`, statistics zeroed (on torch's current stream)"""
        t = self._t
        ts = self.env.reset()
        self._ret.zero_(); self._len.zero_(); self._tot.zero_(); self._sum_ret.zero_()
        if self.adder is not None:
            self.adder.observe(t.zeros(self.env.batch_size, self.adder.act_dim, device=self.env.device), ts, self.env.flat_observation)

    def _flagged(self) -> int:
        """Envs whose step met more simultaneous contacts / constraint rows than the kernel carries (`ffe_get_task_state` int 7: the
        deepest contacts are kept, the env is flagged - walk_on_ball: sticky over the episode; flight: the last control step)."""
        if not hasattr(self.env, "get_task_state"):
            return 0
        w = self.env.get_task_state()[0][:, 7]
        is_flight = hasattr(self.env, "ghost_accel_z")
        return int((((w >> 8) & 255) != 0).sum()) if is_flight else int((w != 0).sum())



class GroupedActorLoop:
    """The actor loop over asynchronous env groups (`flybody_amd.groups.EnvGroups`): one `BatchedActorLoop` per group, everything a group
    does - policy, env step, adder, statistics - on that group's stream, the host enqueueing the groups round-robin.  No group waits for
    another, as the reference's actor processes do not (`train_dmpo_ray.py:432-452`); one group's launch drains while the next group's
    fills the device.  The policy is shared (its weights are read-only here)."""

    def __init__(self, groups, policy, adders=None):
        self.groups = groups
        self.loops = [BatchedActorLoop(e, policy, adders[g] if adders is not None else None) for g, e in enumerate(groups.envs)]

    def run(self, num_steps: int, graph: bool = False) -> dict:
        """`graph`: every group's iteration is captured once into a HIP graph on the group's stream and replayed (two dozen launches per
        group and step otherwise: with several groups the host becomes the bound before the device does)."""
        import torch

        for g, lp in enumerate(self.loops):
            with self.groups.on(g):
                lp._begin()
        self.groups.synchronize()
        graphs = None
        if graph:
            graphs = []
            for g, lp in enumerate(self.loops):
                with self.groups.on(g):
                    for _ in range(3):
                        lp._iteration()
                self.groups.streams[g].synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=self.groups.streams[g]):
                    lp._iteration()
                graphs.append(gr)
            self.groups.synchronize()
        start = time.perf_counter()
        for _ in range(num_steps):
            for g, lp in enumerate(self.loops):
                with self.groups.on(g):
                    if graphs is not None:
                        graphs[g].replay()
                    else:
                        lp._iteration()
        self.groups.synchronize()
        wall = time.perf_counter() - start
        n = sum(int(lp._tot[0].item()) for lp in self.loops)
        ret = sum(float(lp._sum_ret.item()) for lp in self.loops)
        length = sum(int(lp._tot[1].item()) for lp in self.loops)
        return {"episodes": n, "episode_return": ret / n if n else float("nan"), "episode_length": length / n if n else float("nan"),
                "steps_per_second": num_steps * self.groups.batch_size / wall, "capacity_flagged_envs": sum(lp._flagged() for lp in self.loops)}
