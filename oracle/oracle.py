"""ctypes front-end for the float64 CPU oracle (`fly_oracle.c`).

TEST INFRASTRUCTURE ONLY - see the header of `fly_oracle.c`.  Imported by tests/, by
`__graft_entry__.smoke()` and by `bench.py`'s cpu_baseline leg; never by `flybody_amd/`.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "_build", "libfly_oracle.so")

FO_NO_FLUID, FO_NO_LIMIT, FO_NO_DAMPER, FO_NO_SPRING, FO_NO_GRAVITY, FO_NO_ACTUATION = 1, 2, 4, 8, 16, 32
FO_NO_CONTACT, FO_NO_NOSLIP, FO_NO_ADHESION = 64, 128, 256


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "fly_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.fo_model_load.restype = vp
        L.fo_model_load.argtypes = [C.c_char_p]
        L.fo_data_new.restype = vp
        L.fo_data_new.argtypes = [vp]
        for n in ("fo_nq", "fo_nv", "fo_nu", "fo_nbody", "fo_njnt", "fo_nM", "fo_naction"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [vp]
        for n in ("fo_qpos", "fo_qvel", "fo_ctrl", "fo_qacc", "fo_qacc_smooth", "fo_qfrc_bias", "fo_qfrc_passive",
                  "fo_qfrc_actuator", "fo_qfrc_constraint", "fo_xpos", "fo_xquat", "fo_xipos", "fo_subtree_com",
                  "fo_cvel", "fo_sensors", "fo_time"):
            getattr(L, n).restype = dp
            getattr(L, n).argtypes = [vp]
        for n in ("fo_na", "fo_ngeom", "fo_npair", "fo_npair_unsupported", "fo_nsite", "fo_ntouch", "fo_nforce",
                  "fo_ncon", "fo_near_unsupported"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [vp]
        for n in ("fo_act", "fo_efc_force", "fo_efc_J", "fo_efc_aref", "fo_efc_D", "fo_sens_touch", "fo_sens_force",
                  "fo_site_xpos", "fo_geom_xpos", "fo_geom_xmat", "fo_qpos_spring"):
            getattr(L, n).restype = dp
            getattr(L, n).argtypes = [vp]
        L.fo_efc_type.restype = ip
        L.fo_efc_type.argtypes = [vp]
        L.fo_contact_info.argtypes = [vp, C.c_int, dp]
        L.fo_nefc.restype = C.c_int
        L.fo_nefc.argtypes = [vp]
        L.fo_solver_iter.restype = C.c_int
        L.fo_solver_iter.argtypes = [vp]
        L.fo_set_flags.argtypes = [vp, C.c_int]
        L.fo_set_timestep.argtypes = [vp, C.c_double]
        L.fo_get_timestep.restype = C.c_double
        L.fo_get_timestep.argtypes = [vp]
        L.fo_step1.argtypes = [vp, vp]
        L.fo_step2.argtypes = [vp, vp]
        L.fo_step.argtypes = [vp, vp]
        L.fo_forward.argtypes = [vp, vp, C.c_int]
        L.fo_dense_M.argtypes = [vp, vp, dp]
        L.fo_energy.restype = C.c_double
        L.fo_energy.argtypes = [vp, vp, dp, dp]
        L.fo_momentum.argtypes = [vp, vp, dp, dp]
        L.fo_rng.restype = C.c_uint64
        L.fo_rng.argtypes = [C.c_uint64] * 4
        L.fo_u01.restype = C.c_double
        L.fo_u01.argtypes = [C.c_uint64]
        L.fo_env_new.restype = vp
        L.fo_env_new.argtypes = [vp, C.c_int, dp, ip, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int, ip, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint64]
        L.fo_env_step.argtypes = [vp, dp, dp, dp, dp, ip]
        L.fo_env_force_next.argtypes = [vp, C.c_int, C.c_double]
        L.fo_env_set_pad_first_obs.argtypes = [vp, C.c_int]
        L.fo_env_request_reset.argtypes = [vp]
        L.fo_env_data.restype = vp
        L.fo_env_data.argtypes = [vp]
        L.fo_env_ghost.argtypes = [vp, dp]
        L.fo_env_wbpg_state.argtypes = [vp, ip, ip, dp]
        L.fo_env_wbpg_reset.argtypes = [vp, C.c_double, dp, dp]
        L.fo_env_wbpg_step.argtypes = [vp, C.c_double, dp]
        L.fo_test_quat.argtypes = [C.c_int, dp, dp, dp]
        L.fo_debug_constraint_eval.restype = C.c_double
        L.fo_debug_constraint_eval.argtypes = [vp, vp, dp, dp, dp]
        L.fo_debug_make_constraint.argtypes = [vp, vp]
        L.fo_debug_qcqp2.restype = C.c_int
        L.fo_debug_qcqp2.argtypes = [dp, dp, dp, dp, C.c_double]
        L.fo_debug_ray_capsule.restype = C.c_double
        L.fo_debug_ray_capsule.argtypes = [dp, dp, C.c_double, C.c_double]
        L.fo_ball_env_new.restype = vp
        L.fo_ball_env_new.argtypes = [vp, C.c_double, C.c_double]
        L.fo_ball_env_data.restype = vp
        L.fo_ball_env_data.argtypes = [vp]
        L.fo_ball_env_set_pad_first_obs.argtypes = [vp, C.c_int]
        L.fo_ball_env_request_reset.argtypes = [vp]
        L.fo_ball_obs_dim.restype = C.c_int
        L.fo_ball_obs_dim.argtypes = [vp]
        L.fo_ball_env_step.argtypes = [vp, dp, dp, dp, dp, ip]
        L.fo_ball_env_hist.argtypes = [vp, ip, dp, C.c_int]
        L.fo_ball_env_hist_detected.argtypes = [vp, ip, C.c_int]
        L.fo_set_measure_unsupported.argtypes = [vp, C.c_int]
        L.fo_unsupported_min_sep.restype = C.c_double
        L.fo_unsupported_min_sep.argtypes = [vp, vp, ip, ip]
        L.fo_unsupported_touching.restype = C.c_int
        L.fo_unsupported_touching.argtypes = [vp, vp, ip, ip, C.c_int]
        L.fo_self_min_clear.restype = C.c_double
        L.fo_self_min_clear.argtypes = [vp, ip, ip]
        L.fo_convex_separation.restype = C.c_double
        L.fo_convex_separation.argtypes = [vp, vp, C.c_int, C.c_int, dp]
        L.fo_geom_sdf.restype = C.c_double
        L.fo_geom_sdf.argtypes = [vp, vp, C.c_int, dp, dp, dp]
        L.fo_convex_distance.restype = C.c_double
        L.fo_convex_distance.argtypes = [vp, vp, C.c_int, C.c_int, dp, dp]
        L.fo_geom_support.argtypes = [vp, vp, C.c_int, dp, dp]
        L.fo_geom_type.restype = ip
        L.fo_geom_type.argtypes = [vp]
        L.fo_geom_size.restype = dp
        L.fo_geom_size.argtypes = [vp]
        L.fo_walk_env_new.restype = vp
        L.fo_walk_env_new.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_int, ip, C.c_int, ip, C.c_int, ip, dp, dp, dp, dp,
                                      C.c_double, C.c_int, ip, dp, C.c_uint64, C.c_uint64, C.c_int]
        L.fo_walk_env_data.restype = vp
        L.fo_walk_env_data.argtypes = [vp]
        L.fo_walk_env_set_pad_first_obs.argtypes = [vp, C.c_int]
        L.fo_walk_env_request_reset.argtypes = [vp]
        L.fo_walk_env_force_next.argtypes = [vp, C.c_int]
        L.fo_walk_env_traj.restype = C.c_int
        L.fo_walk_env_traj.argtypes = [vp]
        L.fo_walk_obs_dim.restype = C.c_int
        L.fo_walk_obs_dim.argtypes = [vp, C.c_int]
        L.fo_walk_env_step.argtypes = [vp, dp, dp, dp, dp, ip]
        L.fo_walk_env_features.argtypes = [vp, dp, dp, dp, dp]
        L.fo_walk_env_reward_factors.argtypes = [vp, C.c_int, dp]
        L.fo_ncon_matter.restype = C.c_int
        L.fo_ncon_matter.argtypes = [vp, vp]
        L.fo_data_contact_hist.restype = C.c_int
        L.fo_data_contact_hist.argtypes = [vp, ip, dp, C.c_int]
        L.fo_data_deep_ratio.restype = C.c_double
        L.fo_data_deep_ratio.argtypes = [vp, C.c_int]
        L.fo_env_counters.restype = C.c_int
        L.fo_env_counters.argtypes = [vp, ip, ip]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class OracleModel:
    def __init__(self, blob_path: str):
        self.L = lib()
        self.ptr = self.L.fo_model_load(blob_path.encode())
        if not self.ptr:
            raise RuntimeError(f"cannot load {blob_path}")
        for n in ("nq", "nv", "nu", "nbody", "njnt", "nM", "naction", "na", "ngeom", "npair", "npair_unsupported",
                  "nsite", "ntouch", "nforce"):
            setattr(self, n, getattr(self.L, "fo_" + n)(self.ptr))
        self.qpos_spring = np.ctypeslib.as_array(self.L.fo_qpos_spring(self.ptr), shape=(self.nq,))

    def set_flags(self, flags: int):
        self.L.fo_set_flags(self.ptr, flags)

    @property
    def timestep(self):
        return self.L.fo_get_timestep(self.ptr)

    @timestep.setter
    def timestep(self, h):
        self.L.fo_set_timestep(self.ptr, float(h))


class OracleData:
    """numpy views straight onto the C arrays."""

    def __init__(self, model: OracleModel, ptr=None):
        self.m, self.L = model, model.L
        self.ptr = ptr if ptr is not None else self.L.fo_data_new(model.ptr)
        m = model

        def view(name, n):
            return np.ctypeslib.as_array(getattr(self.L, name)(self.ptr), shape=(n,))

        self.qpos, self.qvel, self.ctrl = view("fo_qpos", m.nq), view("fo_qvel", m.nv), view("fo_ctrl", m.nu)
        self.qacc, self.qacc_smooth = view("fo_qacc", m.nv), view("fo_qacc_smooth", m.nv)
        self.qfrc_bias, self.qfrc_passive = view("fo_qfrc_bias", m.nv), view("fo_qfrc_passive", m.nv)
        self.qfrc_actuator, self.qfrc_constraint = view("fo_qfrc_actuator", m.nv), view("fo_qfrc_constraint", m.nv)
        self.xpos = view("fo_xpos", 3 * m.nbody).reshape(-1, 3)
        self.xquat = view("fo_xquat", 4 * m.nbody).reshape(-1, 4)
        self.xipos = view("fo_xipos", 3 * m.nbody).reshape(-1, 3)
        self.subtree_com = view("fo_subtree_com", 3 * m.nbody).reshape(-1, 3)
        self.cvel = view("fo_cvel", 6 * m.nbody).reshape(-1, 6)
        self.sensors = view("fo_sensors", 9)  # gyro, velocimeter, accelerometer
        self._time = view("fo_time", 1)
        if m.na:
            self.act = view("fo_act", m.na)
        if m.ngeom:
            self.sens_touch, self.sens_force = view("fo_sens_touch", m.ntouch), view("fo_sens_force", 3 * m.nforce).reshape(-1, 3)
            self.site_xpos = view("fo_site_xpos", 3 * m.nsite).reshape(-1, 3)
            self.geom_xpos = view("fo_geom_xpos", 3 * m.ngeom).reshape(-1, 3)
            self.geom_xmat = view("fo_geom_xmat", 9 * m.ngeom).reshape(-1, 3, 3)
            self.efc_force = view("fo_efc_force", 300)

    @property
    def ncon(self):
        return self.L.fo_ncon(self.ptr)

    @property
    def near_unsupported(self):
        return self.L.fo_near_unsupported(self.ptr)

    @property
    def nefc(self):
        return self.L.fo_nefc(self.ptr)

    def efc(self):
        """(J [nefc, nv], aref, D, type) of the rows instantiated by the last forward pass."""
        n, nv = self.nefc, self.m.nv
        J = np.ctypeslib.as_array(self.L.fo_efc_J(self.ptr), shape=(n * nv,)).reshape(n, nv).copy()
        aref = np.ctypeslib.as_array(self.L.fo_efc_aref(self.ptr), shape=(n,)).copy()
        D = np.ctypeslib.as_array(self.L.fo_efc_D(self.ptr), shape=(n,)).copy()
        ty = np.ctypeslib.as_array(self.L.fo_efc_type(self.ptr), shape=(n,)).copy()
        return J, aref, D, ty

    def constraint_eval(self, jar, hessian=False):
        jar = np.ascontiguousarray(jar, dtype=np.float64)
        force = np.zeros_like(jar)
        H = np.zeros((self.m.nv, self.m.nv)) if hessian else None
        cost = self.L.fo_debug_constraint_eval(self.m.ptr, self.ptr, _dp(jar), _dp(force), _dp(H) if hessian else None)
        return cost, force, H

    @property
    def ncon_matter(self):
        """Detected contacts that take part in something: active, or on a body with an adhesion actuator (what the kernels keep)."""
        return int(self.L.fo_ncon_matter(self.m.ptr, self.ptr))

    def contact_hist(self, reset=True):
        """(active contacts each substep used, closest approach of any candidate pair to a switching distance) since the last reset."""
        counts, gap = np.zeros(16, dtype=np.int32), C.c_double(0)
        n = self.L.fo_data_contact_hist(self.ptr, _ip(counts), C.byref(gap), 1 if reset else 0)
        return counts[:n].copy(), float(gap.value)

    def deep_ratio(self, reset=True):
        """Deepest overlap of a convex pair since the last call, in units of the thinner geom's smallest semi-axis (a wing blade driven
        through the abdomen by a stroke: past ~0.5 the direction of least overlap is no longer unique to float32 rounding)."""
        return float(self.L.fo_data_deep_ratio(self.ptr, 1 if reset else 0))

    def contacts(self):
        """Rows: geom1, geom2, dim, exclude, efc_adr, dist, pos[3], normal[3], mu, friction, includemargin, normal force."""
        out = np.zeros((self.ncon, 16))
        for i in range(self.ncon):
            self.L.fo_contact_info(self.ptr, i, _dp(out[i]))
        return out

    def unsupported_min_sep(self):
        """(lower bound of the smallest separation among the ellipsoid / cylinder pairs, geom1, geom2); needs
        `model.L.fo_set_measure_unsupported(model.ptr, 1)`."""
        g1, g2 = C.c_int(), C.c_int()
        s = self.L.fo_unsupported_min_sep(self.m.ptr, self.ptr, C.byref(g1), C.byref(g2))
        return s, g1.value, g2.value

    def unsupported_touching(self):
        """[(geom1, geom2)] of the ellipsoid / cylinder pairs whose separation is within their margin (measurement mode)."""
        g1, g2 = np.zeros(16, dtype=np.int32), np.zeros(16, dtype=np.int32)
        n = self.L.fo_unsupported_touching(self.m.ptr, self.ptr, _ip(g1), _ip(g2), 16)
        return n, list(zip(g1[: min(n, 16)].tolist(), g2[: min(n, 16)].tolist()))

    def self_min_clear(self):
        """(smallest dist - margin among the sphere / capsule fly-fly candidate pairs that passed the bounding test, geom1, geom2)."""
        g1, g2 = C.c_int(), C.c_int()
        s = self.L.fo_self_min_clear(self.ptr, C.byref(g1), C.byref(g2))
        return s, g1.value, g2.value

    @property
    def time(self):
        return float(self._time[0])

    @property
    def nefc(self):
        return self.L.fo_nefc(self.ptr)

    @property
    def solver_iter(self):
        return self.L.fo_solver_iter(self.ptr)

    def step1(self):
        self.L.fo_step1(self.m.ptr, self.ptr)

    def step2(self):
        self.L.fo_step2(self.m.ptr, self.ptr)

    def step(self):
        self.L.fo_step(self.m.ptr, self.ptr)

    def forward(self, skip_actuation=False):
        self.L.fo_forward(self.m.ptr, self.ptr, int(skip_actuation))

    def dense_M(self):
        out = np.zeros((self.m.nv, self.m.nv))
        self.L.fo_dense_M(self.m.ptr, self.ptr, _dp(out))
        return out

    def energy(self):
        k, p = C.c_double(), C.c_double()
        e = self.L.fo_energy(self.m.ptr, self.ptr, C.byref(k), C.byref(p))
        return e, k.value, p.value

    def momentum(self):
        lin, ang = np.zeros(3), np.zeros(3)
        self.L.fo_momentum(self.m.ptr, self.ptr, _dp(lin), _dp(ang))
        return lin, ang


class OracleFlightEnv:
    """Single-instance float64 flight-imitation env (dm_env semantics, one call = one control step).

    `wbpg` is a `flybody_amd.tasks.wbpg.WingBeatTables`-like object (attributes beat_freqs, tab_off, traj,
    phase, base_freq, rel_range, rate, dt_ctrl).  The preprocessed references come either stacked, `ref_qpos (N,T,7)`
    root poses and `ref_qvel (N,T,6)`, or as one object with `qpos (rows,7)`, `qvel (rows,6)`, `off (N+1,)` (trajectories of
    individual lengths, `trajectory_loaders.py:98-100`).  `time_limit` is in seconds (`fly_envs.py:54`)."""

    OBS = 104

    def __init__(self, model: OracleModel, wbpg, ref_qpos, ref_qvel=None, *, future_steps=5, time_limit=0.6,
                 terminal_com_dist=2.0, ghost_accel_z=0.0, seed=0, env_id=0):
        if ref_qvel is None:
            rq, rv, roff = ref_qpos.qpos, ref_qpos.qvel, ref_qpos.off
        else:
            rq, rv = np.asarray(ref_qpos, dtype=np.float64), np.asarray(ref_qvel, dtype=np.float64)
            n_, t_ = rq.shape[:2]
            rq, rv, roff = rq.reshape(n_ * t_, 7), rv.reshape(n_ * t_, 6), np.arange(n_ + 1) * t_
        self.model, self.L = model, model.L
        self._keep = dict(
            bf=np.ascontiguousarray(wbpg.beat_freqs, dtype=np.float64),
            off=np.ascontiguousarray(wbpg.tab_off, dtype=np.int32),
            traj=np.ascontiguousarray(wbpg.traj, dtype=np.float64),
            phase=np.ascontiguousarray(wbpg.phase, dtype=np.float64),
            rq=np.ascontiguousarray(rq, dtype=np.float64),
            rv=np.ascontiguousarray(rv, dtype=np.float64),
            roff=np.ascontiguousarray(roff, dtype=np.int32),
        )
        k = self._keep
        n = len(k["roff"]) - 1
        # flight_imitation.py:107: round(time_limit / control_timestep) caps the trajectory-end rule; the episode time limit
        # itself is tested on the accumulated physics time inside the oracle
        time_limit_steps = int(round(time_limit / wbpg.dt_ctrl))
        self.ptr = self.L.fo_env_new(model.ptr, len(k["bf"]), _dp(k["bf"]), _ip(k["off"]), _dp(k["traj"]), _dp(k["phase"]),
                                     float(wbpg.base_freq), float(wbpg.rel_range), float(wbpg.rate), float(wbpg.dt_ctrl),
                                     n, _ip(k["roff"]), _dp(k["rq"]), _dp(k["rv"]), future_steps, time_limit_steps, float(time_limit),
                                     float(terminal_com_dist), float(ghost_accel_z), seed, env_id)
        self.data = OracleData(model, self.L.fo_env_data(self.ptr))
        self.naction = model.naction

    def force_next(self, traj_idx: int, phase: float):
        self.L.fo_env_force_next(self.ptr, int(traj_idx), float(phase))

    def set_pad_first_obs(self, v: bool):
        self.L.fo_env_set_pad_first_obs(self.ptr, int(v))

    def reset(self):
        """dm_env reset(): returns the FIRST timestep of a new episode."""
        self.L.fo_env_request_reset(self.ptr)
        return self.step(np.zeros(self.naction))

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros(self.OBS)
        r, dsc, st = C.c_double(), C.c_double(), C.c_int()
        self.L.fo_env_step(self.ptr, _dp(a), _dp(obs), C.byref(r), C.byref(dsc), C.byref(st))
        return st.value, r.value, dsc.value, obs

    def ghost(self):
        g = np.zeros(7)
        self.L.fo_env_ghost(self.ptr, _dp(g))
        return g

    def wbpg_state(self):
        s, f, c = C.c_int(), C.c_int(), C.c_double()
        self.L.fo_env_wbpg_state(self.ptr, C.byref(s), C.byref(f), C.byref(c))
        return s.value, f.value, c.value

    def wbpg_reset(self, phase):
        q, v = np.zeros(6), np.zeros(6)
        self.L.fo_env_wbpg_reset(self.ptr, float(phase), _dp(q), _dp(v))
        return q, v

    def wbpg_step(self, ctrl_freq):
        q = np.zeros(6)
        self.L.fo_env_wbpg_step(self.ptr, float(ctrl_freq), _dp(q))
        return q

    def counters(self):
        t, s = C.c_int(), C.c_int()
        nr = self.L.fo_env_counters(self.ptr, C.byref(t), C.byref(s))
        return t.value, s.value, bool(nr)


class OracleBallEnv:
    """Single-instance float64 walk_on_ball env (`fly_envs.py:125-157`); one call = one control step."""

    #: observation slices, in emission order
    LAYOUT = (("accelerometer", 3), ("actuator_activation", 59), ("appendages_pos", 21), ("ball_qvel", 3), ("force", 18),
              ("gyro", 3), ("joints_pos", 85), ("joints_vel", 85), ("touch", 6), ("velocimeter", 3), ("world_zaxis", 3))

    def __init__(self, model: OracleModel, control_timestep=2e-3, time_limit=2.0):
        self.model, self.L = model, model.L
        self.ptr = self.L.fo_ball_env_new(model.ptr, float(control_timestep), float(time_limit))
        self.data = OracleData(model, self.L.fo_ball_env_data(self.ptr))
        self.naction = model.naction
        self.OBS = self.L.fo_ball_obs_dim(model.ptr)

    def set_pad_first_obs(self, v: bool):
        self.L.fo_ball_env_set_pad_first_obs(self.ptr, int(v))

    def reset(self):
        self.L.fo_ball_env_request_reset(self.ptr)
        return self.step(np.zeros(self.naction))

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros(self.OBS)
        r, dsc, st = C.c_double(), C.c_double(), C.c_int()
        self.L.fo_ball_env_step(self.ptr, _dp(a), _dp(obs), C.byref(r), C.byref(dsc), C.byref(st))
        return st.value, r.value, dsc.value, obs

    def contact_history(self, n=10):
        """Per substep of the last control step: (contacts inside their includemargin, smallest |dist - includemargin| over
        all candidate pairs) - what the GPU parity tests use to tell a contact flip from a numerical error."""
        counts, gaps = np.zeros(n, dtype=np.int32), np.zeros(n)
        self.L.fo_ball_env_hist(self.ptr, _ip(counts), _dp(gaps), n)
        return counts, gaps

    def detected_history(self, n=10):
        """Per substep of the last control step: contacts detected (inside their margin, active or not; adhesion acts on all of them)."""
        counts = np.zeros(n, dtype=np.int32)
        self.L.fo_ball_env_hist_detected(self.ptr, _ip(counts), n)
        return counts

    def split(self, obs):
        out, o = {}, 0
        for name, n in self.LAYOUT:
            out[name] = obs[o : o + n]
            o += n
        return out


class OracleWalkEnv:
    """Single-instance float64 walk_imitation env (`fly_envs.py:75-122`); one call = one control step.  `refs` is a
    `flybody_amd.tasks.walking.WalkRefSet`-like object (qpos, qvel, root2site, joint_quat, off); `mocap_jnt` / `mocap_site` index
    the model's joints / sites; `retract` = (qpos addresses, values) written after the reference pose at reset."""

    def __init__(self, model: OracleModel, refs, mocap_jnt, mocap_site, retract, *, control_timestep=2e-3, time_limit=10.0,
                 future_steps=64, terminal_com_dist=0.3, seed=0, env_id=0, inference_mode=False):
        self.model, self.L = model, model.L
        keep = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self._mj, self._ms = keep(mocap_jnt, np.int32), keep(mocap_site, np.int32)
        self._off = keep(refs.off, np.int32)
        self._rq, self._rv = keep(refs.qpos, np.float64), keep(refs.qvel, np.float64)
        self._rs, self._rj = keep(refs.root2site, np.float64), keep(refs.joint_quat, np.float64)
        self._oq, self._ov = keep(retract[0], np.int32), keep(retract[1], np.float64)
        self.nmj, self.nms, self.future_steps = len(self._mj), len(self._ms), int(future_steps)
        self.ptr = self.L.fo_walk_env_new(model.ptr, float(control_timestep), float(time_limit), self.future_steps, self.nmj,
                                          _ip(self._mj), self.nms, _ip(self._ms), len(self._off) - 1, _ip(self._off), _dp(self._rq),
                                          _dp(self._rv), _dp(self._rs), _dp(self._rj), float(terminal_com_dist), len(self._oq),
                                          _ip(self._oq), _dp(self._ov), int(seed), int(env_id), int(inference_mode))
        self.data = OracleData(model, self.L.fo_walk_env_data(self.ptr))
        self.naction = model.naction
        self.OBS = self.L.fo_walk_obs_dim(model.ptr, self.future_steps)
        F = self.future_steps + 1
        self.LAYOUT = (("accelerometer", 3), ("actuator_activation", model.na), ("appendages_pos", 21), ("force", 18), ("gyro", 3),
                       ("joints_pos", 85), ("joints_vel", 85), ("ref_displacement", 3 * F), ("ref_root_quat", 4 * F), ("touch", 6),
                       ("velocimeter", 3), ("world_zaxis", 3))

    def set_pad_first_obs(self, v: bool):
        self.L.fo_walk_env_set_pad_first_obs(self.ptr, int(v))

    def force_next(self, traj_idx: int):
        self.L.fo_walk_env_force_next(self.ptr, int(traj_idx))

    @property
    def traj_idx(self):
        return self.L.fo_walk_env_traj(self.ptr)

    def reset(self):
        self.L.fo_walk_env_request_reset(self.ptr)
        return self.step(np.zeros(self.naction))

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros(self.OBS)
        r, dsc, st = C.c_double(), C.c_double(), C.c_int()
        self.L.fo_walk_env_step(self.ptr, _dp(a), _dp(obs), C.byref(r), C.byref(dsc), C.byref(st))
        return st.value, r.value, dsc.value, obs

    def features(self):
        com, qv = np.zeros(3), np.zeros(6 + self.nmj)
        r2s, jq = np.zeros((self.nms, 3)), np.zeros((1 + self.nmj, 4))
        self.L.fo_walk_env_features(self.ptr, _dp(com), _dp(qv), _dp(r2s), _dp(jq))
        return {"com": com, "qvel": qv, "root2site": r2s, "joint_quat": jq}

    def reward_factors(self, step: int):
        out = np.zeros(4)
        self.L.fo_walk_env_reward_factors(self.ptr, int(step), _dp(out))
        return out

    def split(self, obs):
        out, o = {}, 0
        for name, n in self.LAYOUT:
            out[name] = obs[o : o + n]
            o += n
        return out


def rng_u64(seed, env, episode, stream):
    return int(lib().fo_rng(seed, env, episode, stream))


def rng_u01(seed, env, episode, stream):
    return float(lib().fo_u01(lib().fo_rng(seed, env, episode, stream)))


def test_quat(op: int, a, b=None, nout=4):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.zeros(4) if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros(nout)
    lib().fo_test_quat(op, _dp(a), _dp(b), _dp(out))
    return out
