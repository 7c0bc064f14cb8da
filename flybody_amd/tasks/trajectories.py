"""Reference flight trajectories: on-disk layout and the per-episode preprocessing.

The reference reads an HDF5 file with `trajectories/<zero-padded idx>/{com_qpos (T_i,7), com_qvel (T_i,6)}`
(`tasks/trajectory_loaders.py:68-132`); trajectories have their own lengths T_i (`:98-100`) and an episode's
`_traj_timesteps` follows the length of the trajectory it drew (`tasks/flight_imitation.py:107-108`).  h5py is not
available to this build, so the same content is held in an `.npz`: `com_qpos (rows,7)`, `com_qvel (rows,6)` with the rows of
all trajectories concatenated, `traj_off (N+1,)` = first row of each, and scalar `timestep_seconds`
(`tools/convert_hdf5_to_npz.py`).  The stacked layout `com_qpos (N,T,7)` / `com_qvel (N,T,6)` of equal-length sets is
still read.

`preprocess` applies, once for the whole set, what the reference does at every episode start:
xy re-centring on the first sample (`trajectory_loaders.py:130`) and CoM -> root-joint position
(`tasks/flight_imitation.py:102-104` via `task_utils.com2root`, `task_utils.py:194-213`).
"""

from __future__ import annotations

import numpy as np

from .constants import _ROOT2COM_OFFSET


def _rotate(vec, quat):
    """q v q^-1 for unit-or-not quaternions, batched (`quaternions.py:105-134`)."""
    w, x, y, z = np.moveaxis(quat, -1, 0)
    n2 = w * w + x * x + y * y + z * z
    vx, vy, vz = np.moveaxis(np.broadcast_to(vec, quat.shape[:-1] + (3,)), -1, 0)
    # rotation matrix of q / |q|
    r = np.stack([
        (w * w + x * x - y * y - z * z) * vx + 2 * (x * y - w * z) * vy + 2 * (x * z + w * y) * vz,
        2 * (x * y + w * z) * vx + (w * w - x * x + y * y - z * z) * vy + 2 * (y * z - w * x) * vz,
        2 * (x * z - w * y) * vx + 2 * (y * z + w * x) * vy + (w * w - x * x - y * y + z * z) * vz,
    ], axis=-1)
    return r / n2[..., None]


def com2root(com, quat, offset=None):
    """`task_utils.py:194-213`."""
    offset = np.asarray(_ROOT2COM_OFFSET if offset is None else offset, dtype=np.float64)
    return com + _rotate(-offset, quat)


def root2com(root_qpos, offset=None):
    """`task_utils.py:174-191`."""
    offset = np.asarray(_ROOT2COM_OFFSET if offset is None else offset, dtype=np.float64)
    return root_qpos[..., :3] + _rotate(offset, root_qpos[..., 3:7])


def preprocess(com_qpos: np.ndarray, com_qvel: np.ndarray):
    """(N,T,7),(N,T,6) CoM trajectories -> root-pose trajectories the ghost and observations use."""
    q = np.array(com_qpos, dtype=np.float64, copy=True)
    q[..., :2] -= q[..., :1, :2]
    root_pos = com2root(q[..., :3], q[..., 3:7])
    return np.concatenate((root_pos, q[..., 3:7]), axis=-1), np.asarray(com_qvel, dtype=np.float64)


class RefSet:
    """Preprocessed reference trajectories of individual lengths: `qpos (rows,7)` root poses, `qvel (rows,6)`, and
    `off (N+1,)` int32 row offsets - the layout `ffe_flight_task` takes (include/flybody_env.h)."""

    def __init__(self, qpos, qvel, off):
        self.qpos = np.ascontiguousarray(qpos, dtype=np.float64)
        self.qvel = np.ascontiguousarray(qvel, dtype=np.float64)
        self.off = np.ascontiguousarray(off, dtype=np.int32)
        assert self.qpos.ndim == 2 and self.qpos.shape[1] == 7 and self.qvel.shape == (len(self.qpos), 6)
        assert self.off[0] == 0 and self.off[-1] == len(self.qpos) and np.all(np.diff(self.off) > 0)

    @property
    def ntraj(self) -> int:
        return len(self.off) - 1

    def lengths(self) -> np.ndarray:
        return np.diff(self.off)

    def trajectory(self, i: int):
        a, b = int(self.off[i]), int(self.off[i + 1])
        return self.qpos[a:b], self.qvel[a:b]


def preprocess_ragged(com_qpos_list, com_qvel_list) -> RefSet:
    """Per-trajectory `preprocess` over a list of (T_i,7) / (T_i,6) arrays."""
    qs, vs, off = [], [], [0]
    for q, v in zip(com_qpos_list, com_qvel_list):
        rq, rv = preprocess(np.asarray(q)[None], np.asarray(v)[None])
        qs.append(rq[0]); vs.append(rv[0]); off.append(off[-1] + len(rq[0]))
    return RefSet(np.concatenate(qs), np.concatenate(vs), off)


def as_refset(ref_qpos, ref_qvel=None) -> RefSet:
    """Accepts a `RefSet` or already preprocessed stacked arrays (N,T,7) / (N,T,6)."""
    if isinstance(ref_qpos, RefSet):
        return ref_qpos
    q, v = np.asarray(ref_qpos, dtype=np.float64), np.asarray(ref_qvel, dtype=np.float64)
    n, t = q.shape[:2]
    return RefSet(q.reshape(n * t, 7), v.reshape(n * t, 6), np.arange(n + 1, dtype=np.int64) * t)


def load_npz(path: str):
    """Returns (list of com_qpos (T_i,7), list of com_qvel (T_i,6), timestep_seconds)."""
    with np.load(path) as f:
        q, v, dt = f["com_qpos"], f["com_qvel"], float(f["timestep_seconds"])
        if "traj_off" in f.files:
            off = f["traj_off"].astype(np.int64)
            return [q[a:b] for a, b in zip(off[:-1], off[1:])], [v[a:b] for a, b in zip(off[:-1], off[1:])], dt
        return list(q), list(v), dt


def save_npz(path: str, com_qpos_list, com_qvel_list, timestep_seconds: float, **extra):
    off = np.concatenate(([0], np.cumsum([len(q) for q in com_qpos_list]))).astype(np.int64)
    np.savez_compressed(path, com_qpos=np.concatenate(com_qpos_list).astype(np.float64),
                        com_qvel=np.concatenate(com_qvel_list).astype(np.float64), traj_off=off,
                        timestep_seconds=np.float64(timestep_seconds), **extra)
