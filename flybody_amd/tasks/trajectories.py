"""Reference flight trajectories: on-disk layout and the per-episode preprocessing.

The reference reads an HDF5 file with `trajectories/<zero-padded idx>/{com_qpos (T,7), com_qvel (T,6)}`
(`tasks/trajectory_loaders.py:68-132`).  h5py is not available to this build, so the same content is
held in an `.npz` with arrays `com_qpos (N,T,7)`, `com_qvel (N,T,6)` and scalar `timestep_seconds`.

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


def load_npz(path: str):
    with np.load(path) as f:
        return f["com_qpos"], f["com_qvel"], float(f["timestep_seconds"])
