"""ctypes binding of include/flybody_env.h.  Fails loudly when the HIP library is missing: the product path
has no CPU fallback."""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLYBODY_ENV_LIB") or os.path.join(_HERE, "csrc", "libflybody_env.so")  # override: tuning variants only

SYMBOLS = [
    "ffe_create_flight", "ffe_destroy", "ffe_spec", "ffe_action_bounds", "ffe_reset", "ffe_reset_envs", "ffe_step",
    "ffe_physics_step", "ffe_force_next_episode", "ffe_get_state", "ffe_set_state", "ffe_get_task_state", "ffe_time_steps", "ffe_time_kernel",
    "ffe_test_quat", "ffe_last_error", "ffe_version", "ffe_create_walk_on_ball", "ffe_get_act", "ffe_set_act",
    "ffe_nstep_create", "ffe_nstep_observe", "ffe_nstep_buffers", "ffe_nstep_destroy", "ffe_nstep_last_error",
    "ffe_pack_timestep", "ffe_episode_stats",
]


class FlightTask(C.Structure):
    _fields_ = [
        ("wb_nfreq", C.c_int32), ("wb_beat_freqs", C.POINTER(C.c_double)), ("wb_tab_off", C.POINTER(C.c_int32)),
        ("wb_traj", C.POINTER(C.c_double)), ("wb_phase", C.POINTER(C.c_double)),
        ("wb_base_freq", C.c_double), ("wb_rel_range", C.c_double), ("wb_rate", C.c_double), ("wb_dt_ctrl", C.c_double),
        ("ntraj", C.c_int32), ("traj_len", C.c_int32), ("ref_qpos", C.POINTER(C.c_double)), ("ref_qvel", C.POINTER(C.c_double)),
        ("traj_off", C.POINTER(C.c_int32)),
        ("future_steps", C.c_int32), ("time_limit_steps", C.c_int32), ("episode_limit_steps", C.c_int32), ("terminal_com_dist", C.c_double),
        ("ghost_accel_z", C.c_double), ("pad_first_obs", C.c_int32), ("physics_flags", C.c_int32),
        ("canonical_actions", C.c_int32), ("clip_actions", C.c_int32),
    ]


class BallTask(C.Structure):
    _fields_ = [("control_timestep", C.c_double), ("time_limit_steps", C.c_int32), ("pad_first_obs", C.c_int32),
                ("physics_flags", C.c_int32), ("canonical_actions", C.c_int32), ("clip_actions", C.c_int32)]


class Spec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("batch", "nq", "nv", "nu", "action_dim", "obs_dim", "nsub")] + [
        ("physics_timestep", C.c_double), ("control_timestep", C.c_double)] + [
        (n, C.c_int32) for n in ("off_accelerometer", "off_gyro", "off_joints_pos", "off_joints_vel", "off_velocimeter",
                                 "off_world_zaxis", "off_ref_displacement", "off_ref_root_quat", "n_obs_joints", "n_ref")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("FLYBODY_ENV_LIB"):
        try:  # a fresh checkout: compile in-tree once (hipcc cross-compiles gfx950 without a GPU)
            from . import build as _build

            _build.build()
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(
                f"{LIB_PATH} is missing and could not be built ({e}); run `python -m flybody_amd.build` (hipcc, gfx950). "
                "flybody_amd has no CPU fallback.") from e
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing; flybody_amd has no CPU fallback.")
    try:  # PyTorch-ROCm bundles its own HIP runtime: load it first so this library binds to the same copy (two HIP
        import torch  # noqa: F401   runtimes in one process do not see each other's devices)
    except Exception:  # noqa: BLE001
        pass
    L = C.CDLL(LIB_PATH)
    vp, fp, ip, dp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers
    L.ffe_create_flight.restype = C.c_int
    L.ffe_create_flight.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FlightTask), C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.ffe_create_walk_on_ball.restype = C.c_int
    L.ffe_create_walk_on_ball.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(BallTask), C.c_int, C.c_int, C.POINTER(vp)]
    L.ffe_get_act.argtypes = [vp, dp, vp]
    L.ffe_set_act.argtypes = [vp, dp, vp]
    L.ffe_get_act.restype = L.ffe_set_act.restype = C.c_int
    L.ffe_destroy.argtypes = [vp]
    L.ffe_spec.argtypes = [vp, C.POINTER(Spec)]
    L.ffe_action_bounds.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ffe_reset.argtypes = [vp, fp, fp, fp, ip, vp]
    L.ffe_reset_envs.argtypes = [vp, vp, fp, fp, fp, ip, vp]
    L.ffe_step.argtypes = [vp, fp, fp, fp, fp, ip, vp]
    L.ffe_physics_step.argtypes = [vp, fp, C.c_int, vp]
    L.ffe_physics_step.restype = C.c_int
    L.ffe_force_next_episode.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double), vp]
    L.ffe_get_state.argtypes = [vp, dp, dp, vp]
    L.ffe_set_state.argtypes = [vp, dp, dp, vp]
    L.ffe_get_task_state.argtypes = [vp, ip, dp, vp]
    L.ffe_time_steps.argtypes = [vp, fp, fp, fp, fp, ip, C.c_int, vp, C.POINTER(C.c_float)]
    L.ffe_time_kernel.argtypes = [vp, fp, fp, fp, fp, ip, C.c_int, vp, C.POINTER(C.c_float)]
    L.ffe_test_quat.argtypes = [C.c_int, fp, fp, fp, C.c_int, vp]
    L.ffe_nstep_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_longlong, C.c_int, C.POINTER(vp)]
    L.ffe_nstep_observe.argtypes = [vp, fp, ip, fp, fp, fp, vp]
    L.ffe_nstep_buffers.argtypes = [vp] + [C.POINTER(C.c_void_p)] * 6
    L.ffe_nstep_destroy.argtypes = [vp]
    L.ffe_pack_timestep.argtypes = [fp, fp, fp, ip, fp, C.c_int, C.c_int, vp]
    L.ffe_pack_timestep.restype = C.c_int
    L.ffe_episode_stats.argtypes = [ip, fp, fp, vp, vp, vp, C.c_int, vp]
    L.ffe_episode_stats.restype = C.c_int
    L.ffe_nstep_last_error.restype = C.c_char_p
    L.ffe_nstep_last_error.argtypes = [vp]
    for s in ("ffe_nstep_create", "ffe_nstep_observe", "ffe_nstep_buffers", "ffe_nstep_destroy"):
        getattr(L, s).restype = C.c_int
    L.ffe_last_error.restype = C.c_char_p
    L.ffe_last_error.argtypes = [vp]
    L.ffe_version.restype = C.c_char_p
    for s in ("ffe_destroy", "ffe_spec", "ffe_action_bounds", "ffe_reset", "ffe_reset_envs", "ffe_step", "ffe_physics_step", "ffe_force_next_episode", "ffe_get_state",
              "ffe_set_state", "ffe_get_task_state", "ffe_time_steps", "ffe_time_kernel", "ffe_test_quat"):
        getattr(L, s).restype = C.c_int
    _lib = L
    return L
