"""`BatchedFlyEnv`: the dm_env-style surface of the reference's fly environments with a leading batch
dimension, backed by the HIP library through the C ABI (include/flybody_env.h).

What it mirrors: `composer.Environment.reset/step/action_spec/observation_spec/
reward_spec/discount_spec/control_timestep` as consumed by `acme.EnvironmentLoop`
(`agents/ray_distributed_dmpo.py:314-315,399-404`).  Tensors are torch-ROCm tensors on the env's device;
PyTorch is used only for device memory and streams.
"""

from __future__ import annotations

import collections
import ctypes as C
import os

import numpy as np

from . import _capi
from .dm_types import Array, BoundedArray, StepType, TimeStep  # noqa: F401  (re-exported)
from .tasks.constants import _WING_PARAMS

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
FLIGHT_BLOB = os.path.join(_ASSETS, "fly_flight.ffmb")
BALL_BLOB = os.path.join(_ASSETS, "fly_ball.ffmb")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def time_limit_control_steps(time_limit: float, physics_timestep: float, nsub: int) -> int:
    """Control steps after which `composer.Environment` ends an episode on `physics.time() >= time_limit`.

    MuJoCo's `time` is a float64 running sum of `opt.timestep` (one addition per `mj_step`), so the count is found by
    replaying that sum, not by `round(time_limit / control_timestep)`: 12 000 additions of 5e-5 give 0.5999999999999421
    (flight, 0.6 s -> 3001 control steps) and 10 000 additions of 2e-4 give 1.9999999999998124 (walk_on_ball, 2.0 s ->
    1001)."""
    if not (time_limit > 0 and physics_timestep > 0 and nsub >= 1):
        raise ValueError("time_limit, physics_timestep and nsub must be positive")
    if time_limit / (physics_timestep * nsub) > 5e7:
        return 2**31 - 1  # effectively unlimited
    t, n = 0.0, 0
    h = float(physics_timestep)
    while t < time_limit:
        for _ in range(nsub):
            t += h
        n += 1
    return n


class BatchedFlyEnv:
    """B independent flight-imitation environments stepping in lock-step on one MI355X.

    One `step()` = one kernel launch = one control step (4 physics substeps + task) of every env.
    Envs that returned LAST reset themselves on the next `step()` and report FIRST, as dm_control's
    composer.Environment does per instance.

    **Output aliasing.**  `reset()` / `step()` return a `TimeStep` whose tensors are views of device buffers the env owns
    and overwrites on the next call: a consumer that keeps the previous timestep (acme's `observe(action, next_timestep)`
    adders do) must copy it, or construct the env with `double_buffer=True`, which alternates two buffer sets so that the
    timestep returned by call *k* stays valid until call *k + 2*."""

    def __init__(self, wbpg, ref_qpos, ref_qvel=None, *, batch_size: int, device: int = 0, seed: int = 0, env_id_base: int = 0,
                 future_steps: int = 5, time_limit: float = 0.6, terminal_com_dist: float = 2.0, pad_first_obs: bool = False,
                 physics_flags: int = 0, canonical_actions: bool = False, clip_actions: bool = False, double_buffer: bool = False,
                 blob_path: str = FLIGHT_BLOB):
        """`ref_qpos` / `ref_qvel`: preprocessed reference set, either stacked arrays (N,T,7) / (N,T,6) or one
        `tasks.trajectories.RefSet` (trajectories of individual lengths)."""
        import json

        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("BatchedFlyEnv needs a HIP device (MI355X); there is no CPU fallback")
        self._torch = torch
        self._L = _capi.lib()
        self.batch_size = int(batch_size)
        self.device = torch.device("cuda", device)
        with open(blob_path, "rb") as f:
            blob = f.read()
        with open(os.path.splitext(blob_path)[0] + ".json") as f:
            self._meta = json.load(f)
        from .model.blob import read_blob

        tens = read_blob(blob_path)
        # the ghost is a wingless fly on a free joint with armature 1 (`tasks/base.py:142-149`): gravity moves it by
        # g * m / (m + 1)
        wing_links = [i for i, b in enumerate(tens["link_body"]) if self._meta["body_name"][b].startswith("wing")]
        m_ghost = float(tens["link_mass"].sum() - tens["link_mass"][wing_links].sum())
        ghost_accel_z = float(tens["opt"][5]) * m_ghost / (m_ghost + 1.0)
        from .tasks.trajectories import as_refset

        refs = as_refset(ref_qpos, ref_qvel)
        self.refs = refs
        self._keep = dict(bf=_f64(wbpg.beat_freqs), off=np.ascontiguousarray(wbpg.tab_off, dtype=np.int32), traj=_f64(wbpg.traj),
                          phase=_f64(wbpg.phase), rq=refs.qpos, rv=refs.qvel, roff=refs.off)
        k = self._keep
        n, t = refs.ntraj, int(refs.lengths().max())
        h_phys = float(tens["opt"][0])
        nsub = int(round(wbpg.dt_ctrl / h_phys))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        task = _capi.FlightTask(
            wb_nfreq=len(k["bf"]), wb_beat_freqs=dp(k["bf"]), wb_tab_off=k["off"].ctypes.data_as(C.POINTER(C.c_int32)),
            wb_traj=dp(k["traj"]), wb_phase=dp(k["phase"]), wb_base_freq=wbpg.base_freq, wb_rel_range=wbpg.rel_range,
            wb_rate=wbpg.rate, wb_dt_ctrl=wbpg.dt_ctrl, ntraj=n, traj_len=t, ref_qpos=dp(k["rq"]), ref_qvel=dp(k["rv"]),
            traj_off=k["roff"].ctypes.data_as(C.POINTER(C.c_int32)),
            future_steps=future_steps, time_limit_steps=int(round(time_limit / wbpg.dt_ctrl)),
            episode_limit_steps=time_limit_control_steps(time_limit, h_phys, nsub),
            terminal_com_dist=float(terminal_com_dist), ghost_accel_z=ghost_accel_z, pad_first_obs=int(pad_first_obs),
            physics_flags=int(physics_flags), canonical_actions=int(canonical_actions), clip_actions=int(clip_actions))
        h = C.c_void_p()
        rc = self._L.ffe_create_flight(blob, len(blob), C.byref(task), self.batch_size, device, seed, env_id_base, C.byref(h))
        if rc != 0:
            raise RuntimeError("ffe_create_flight: " + self._L.ffe_last_error(None).decode())
        self._h = h
        self.ghost_accel_z = ghost_accel_z
        self.canonical_actions = bool(canonical_actions)
        self.spec = _capi.Spec()
        self._check(self._L.ffe_spec(self._h, C.byref(self.spec)))
        s = self.spec
        amin, amax = (C.c_float * s.action_dim)(), (C.c_float * s.action_dim)()
        self._check(self._L.ffe_action_bounds(self._h, amin, amax))
        self._action_min, self._action_max = np.array(amin[:], dtype=np.float32), np.array(amax[:], dtype=np.float32)
        self._alloc_outputs(double_buffer)
        j, r = s.n_obs_joints, s.n_ref
        # key order: enabled walker observables alphabetically, then the task's additions (dm_control Observables)
        self._layout = collections.OrderedDict([
            ("walker/accelerometer", (s.off_accelerometer, (3,))), ("walker/actuator_activation", (0, (0,))),
            ("walker/gyro", (s.off_gyro, (3,))), ("walker/joints_pos", (s.off_joints_pos, (j,))),
            ("walker/joints_vel", (s.off_joints_vel, (j,))), ("walker/velocimeter", (s.off_velocimeter, (3,))),
            ("walker/world_zaxis", (s.off_world_zaxis, (3,))), ("walker/ref_displacement", (s.off_ref_displacement, (r, 3))),
            ("walker/ref_root_quat", (s.off_ref_root_quat, (r, 4)))])

    # ------------------------------------------------------------------------------------------ plumbing
    def _alloc_outputs(self, double_buffer: bool):
        torch, B, s = self._torch, self.batch_size, self.spec
        self._sets = []
        with torch.cuda.device(self.device):
            for _ in range(2 if double_buffer else 1):
                self._sets.append((torch.zeros(B, s.obs_dim, dtype=torch.float32, device=self.device),
                                   torch.zeros(B, dtype=torch.float32, device=self.device),
                                   torch.zeros(B, dtype=torch.float32, device=self.device),
                                   torch.zeros(B, dtype=torch.int32, device=self.device)))
        self._cur = 0
        self._obs, self._reward, self._discount, self._step_type = self._sets[0]

    def _next_outputs(self):
        """Selects the buffer set the next launch writes (the other one keeps the previous timestep when double-buffered)."""
        self._cur = (self._cur + 1) % len(self._sets)
        self._obs, self._reward, self._discount, self._step_type = self._sets[self._cur]

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("flybody_env: " + self._L.ffe_last_error(self._h).decode())

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _timestep(self):
        obs = collections.OrderedDict()
        for key, (off, shape) in self._layout.items():
            n = int(np.prod(shape))
            obs[key] = self._obs[:, off:off + n].view(self.batch_size, *shape)
        return TimeStep(self._step_type, self._reward, self._discount, obs)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ffe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------ dm_env surface
    def reset(self) -> TimeStep:
        self._next_outputs()
        self._check(self._L.ffe_reset(self._h, self._obs.data_ptr(), self._reward.data_ptr(), self._discount.data_ptr(),
                                      self._step_type.data_ptr(), self._stream()))
        return self._timestep()

    def reset_envs(self, mask) -> TimeStep:
        """Starts a new episode in the envs where `mask` (bool / uint8 [B], on the env's device) is set - the per-instance
        `Environment.reset()` of the reference's one-env-per-actor layout - and returns the timestep with those rows
        replaced by their FIRST observation; the other envs are not touched."""
        t = self._torch
        m = t.as_tensor(mask, device=self.device).to(t.uint8).contiguous()
        if tuple(m.shape) != (self.batch_size,):
            raise ValueError(f"mask must have shape ({self.batch_size},)")
        if len(self._sets) > 1:  # the untouched rows must carry over into the buffer set this call writes
            prev = self._sets[self._cur]
            self._next_outputs()
            for dst, src in zip(self._sets[self._cur], prev):
                dst.copy_(src)
        self._check(self._L.ffe_reset_envs(self._h, m.data_ptr(), self._obs.data_ptr(), self._reward.data_ptr(), self._discount.data_ptr(),
                                           self._step_type.data_ptr(), self._stream()))
        return self._timestep()

    def step(self, action) -> TimeStep:
        """`action`: float32 [B, action_dim] tensor on the env's device, in the raw action spec."""
        t = self._torch
        if not (isinstance(action, t.Tensor) and action.is_cuda and action.device == self.device and action.dtype == t.float32
                and action.is_contiguous() and tuple(action.shape) == (self.batch_size, self.spec.action_dim)):
            raise ValueError(f"action must be a contiguous float32 tensor of shape ({self.batch_size}, {self.spec.action_dim}) on {self.device}")
        self._next_outputs()
        self._check(self._L.ffe_step(self._h, action.data_ptr(), self._obs.data_ptr(), self._reward.data_ptr(),
                                     self._discount.data_ptr(), self._step_type.data_ptr(), self._stream()))
        return self._timestep()

    def action_spec(self):
        if self.canonical_actions:  # acme CanonicalSpecWrapper: every action dimension spans [-1, 1]
            return BoundedArray((self.spec.action_dim,), np.float32, -1.0, 1.0, name="\t".join(self._meta["action_names"]))
        return BoundedArray((self.spec.action_dim,), np.float32, self._action_min, self._action_max,
                            name="\t".join(self._meta["action_names"]))

    def raw_action_bounds(self):
        """(minimum, maximum) of the un-wrapped action spec (`fruitfly.py:496-526`)."""
        return self._action_min.copy(), self._action_max.copy()

    def observation_spec(self):
        return collections.OrderedDict((k, Array(shape, np.float32, name=k)) for k, (_, shape) in self._layout.items())

    def reward_spec(self):
        return Array((), np.float32, name="reward")

    def discount_spec(self):
        return BoundedArray((), np.float32, 0.0, 1.0, name="discount")

    def control_timestep(self) -> float:
        return self.spec.control_timestep

    @property
    def flat_observation(self):
        """The [B, obs_dim] buffer the observation dict views into."""
        return self._obs

    # ------------------------------------------------------------------------------------------ test / tooling hooks
    def set_next_trajectory_index(self, idx, phase):
        """`FlightImitationWBPG.set_next_trajectory_index` (`flight_imitation.py:87-91`) per env, plus the wing phase."""
        idx = np.ascontiguousarray(np.broadcast_to(idx, (self.batch_size,)), dtype=np.int32)
        phase = np.ascontiguousarray(np.broadcast_to(phase, (self.batch_size,)), dtype=np.float64)
        self._check(self._L.ffe_force_next_episode(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), phase.ctypes.data_as(C.POINTER(C.c_double)),
                                                   self._stream()))

    def get_state(self):
        t = self._torch
        qpos = t.empty(self.batch_size, self.spec.nq, dtype=t.float64, device=self.device)
        qvel = t.empty(self.batch_size, self.spec.nv, dtype=t.float64, device=self.device)
        self._check(self._L.ffe_get_state(self._h, qpos.data_ptr(), qvel.data_ptr(), self._stream()))
        return qpos, qvel

    def set_state(self, qpos, qvel):
        t = self._torch
        qpos = qpos.to(device=self.device, dtype=t.float64).contiguous()
        qvel = qvel.to(device=self.device, dtype=t.float64).contiguous()
        assert tuple(qpos.shape) == (self.batch_size, self.spec.nq) and tuple(qvel.shape) == (self.batch_size, self.spec.nv)
        self._check(self._L.ffe_set_state(self._h, qpos.data_ptr(), qvel.data_ptr(), self._stream()))
        t.cuda.current_stream(self.device).synchronize()

    def physics_step(self, ctrl, nsteps: int = 1):
        """`physics.set_control(ctrl)` then `nsteps` x `physics.step()` for every env, no task layer (BASELINE config 2).
        `ctrl`: float32 [B, nu] cuda tensor."""
        assert ctrl.is_cuda and ctrl.device == self.device and ctrl.dtype == self._torch.float32 and ctrl.is_contiguous() and tuple(ctrl.shape) == (self.batch_size, self.spec.nu)
        self._check(self._L.ffe_physics_step(self._h, ctrl.data_ptr(), int(nsteps), self._stream()))

    def get_task_state(self):
        t = self._torch
        ints = t.empty(self.batch_size, 8, dtype=t.int32, device=self.device)
        reals = t.empty(self.batch_size, 8, dtype=t.float64, device=self.device)
        self._check(self._L.ffe_get_task_state(self._h, ints.data_ptr(), reals.data_ptr(), self._stream()))
        return ints, reals

    def time_kernel(self, action, iters: int) -> float:
        """Mean milliseconds of the step kernel alone over `iters` launches (HIP events immediately around it)."""
        ms = C.c_float()
        self._check(self._L.ffe_time_kernel(self._h, action.data_ptr(), self._obs.data_ptr(), self._reward.data_ptr(),
                                            self._discount.data_ptr(), self._step_type.data_ptr(), int(iters), self._stream(), C.byref(ms)))
        return float(ms.value)

    def time_steps(self, action, iters: int) -> float:
        """Mean milliseconds per step launch over `iters` launches, by HIP events on the current stream."""
        ms = C.c_float()
        self._check(self._L.ffe_time_steps(self._h, action.data_ptr(), self._obs.data_ptr(), self._reward.data_ptr(),
                                           self._discount.data_ptr(), self._step_type.data_ptr(), int(iters), self._stream(), C.byref(ms)))
        return float(ms.value)


class BatchedBallEnv(BatchedFlyEnv):
    """B independent `walk_on_ball` environments (`fly_envs.py:125-157`) on one MI355X: a tethered fly on a floating
    ball, 10 physics substeps (contacts, elliptic friction cones, noslip, adhesion, filtered actuators) per control step.
    Same dm_env surface and hooks as `BatchedFlyEnv`; `get_state` returns qpos[B, 106] = ball quaternion + 102 hinges and
    qvel[B, 105] = ball angular velocity + hinges."""

    def __init__(self, *, batch_size: int, device: int = 0, time_limit: float = 2.0, control_timestep: float = 2e-3,
                 pad_first_obs: bool = False, physics_flags: int = 0, canonical_actions: bool = False, clip_actions: bool = False,
                 double_buffer: bool = False, blob_path: str = BALL_BLOB):
        import json

        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("BatchedBallEnv needs a HIP device (MI355X); there is no CPU fallback")
        self._torch = torch
        self._L = _capi.lib()
        self.batch_size = int(batch_size)
        self.device = torch.device("cuda", device)
        with open(blob_path, "rb") as f:
            blob = f.read()
        with open(os.path.splitext(blob_path)[0] + ".json") as f:
            self._meta = json.load(f)
        from .model.blob import read_blob

        h_phys = float(read_blob(blob_path)["opt"][0])
        nsub = int(round(control_timestep / h_phys))
        task = _capi.BallTask(control_timestep=float(control_timestep), time_limit_steps=time_limit_control_steps(time_limit, h_phys, nsub),
                              pad_first_obs=int(pad_first_obs), physics_flags=int(physics_flags),
                              canonical_actions=int(canonical_actions), clip_actions=int(clip_actions))
        h = C.c_void_p()
        rc = self._L.ffe_create_walk_on_ball(blob, len(blob), C.byref(task), self.batch_size, device, C.byref(h))
        if rc != 0:
            raise RuntimeError("ffe_create_walk_on_ball: " + self._L.ffe_last_error(None).decode())
        self._h = h
        self.canonical_actions = bool(canonical_actions)
        self.spec = _capi.Spec()
        self._check(self._L.ffe_spec(self._h, C.byref(self.spec)))
        s = self.spec
        amin, amax = (C.c_float * s.action_dim)(), (C.c_float * s.action_dim)()
        self._check(self._L.ffe_action_bounds(self._h, amin, amax))
        self._action_min, self._action_max = np.array(amin[:], dtype=np.float32), np.array(amax[:], dtype=np.float32)
        self._alloc_outputs(double_buffer)
        self._layout = collections.OrderedDict()
        off = 0
        for name, shape in (("accelerometer", (3,)), ("actuator_activation", (s.nu,)), ("appendages_pos", (21,)), ("ball_qvel", (3,)),
                            ("force", (18,)), ("gyro", (3,)), ("joints_pos", (s.n_obs_joints,)), ("joints_vel", (s.n_obs_joints,)),
                            ("touch", (6,)), ("velocimeter", (3,)), ("world_zaxis", (3,))):
            self._layout["walker/" + name] = (off, shape)
            off += int(np.prod(shape))
        assert off == s.obs_dim

    def set_next_trajectory_index(self, idx, phase):
        raise NotImplementedError("walk_on_ball has no reference trajectories")

    def get_act(self):
        t = self._torch
        act = t.empty(self.batch_size, self.spec.nu, dtype=t.float64, device=self.device)
        self._check(self._L.ffe_get_act(self._h, act.data_ptr(), self._stream()))
        return act

    def set_act(self, act):
        t = self._torch
        act = act.to(device=self.device, dtype=t.float64).contiguous()
        assert tuple(act.shape) == (self.batch_size, self.spec.nu)
        self._check(self._L.ffe_set_act(self._h, act.data_ptr(), self._stream()))
        t.cuda.current_stream(self.device).synchronize()
