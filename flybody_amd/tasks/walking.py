"""Host side of the walking imitation task (`fly_envs.walk_imitation`, `tasks/walk_imitation.py:24-191`): reference
snippets with their tracked features, the walker features of `tasks/rewards.py:36-61` on a model blob (numpy, float64) and
a synthetic snippet generator for tests and benches (the reference's walking dataset is not in its repository).

A snippet is what `HDF5WalkingTrajectoryLoader.get_trajectory` returns (`tasks/trajectory_loaders.py:163-215`): `qpos`
[T, 7 + J] (root pose, then the mocap joints' angles), `qvel` [T, 6 + J], `root2site` [T, S, 3] (tracked sites in the
root frame) and `joint_quat` [T, J, 4] (mocap joint orientations in the root frame); the x / y of the root start at 0.
"""

from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np

from ..model import quat as Q
from ..model.blob import read_blob
from ..model.pyref import kinematics

_ASSETS = os.path.join(os.path.dirname(__file__), "..", "assets")


class WalkModelView:
    """The committed `fly_walk` blob + names, exposing what forward kinematics and the task bookkeeping need."""

    def __init__(self, blob_path: str | None = None, meta_path: str | None = None):
        blob_path = blob_path or os.path.join(_ASSETS, "fly_walk.ffmb")
        meta_path = meta_path or os.path.join(_ASSETS, "fly_walk.json")
        t = read_blob(blob_path)
        with open(meta_path) as f:
            self.meta = json.load(f)
        self.m = SimpleNamespace(**{k: np.asarray(v) for k, v in t.items()})
        self.m.nbody, self.m.njnt = len(self.m.body_parentid), len(self.m.jnt_type)
        self.nq, self.nv = len(self.m.qpos0), len(self.m.dof_jntid)
        jn, sn = self.meta["jnt_name"], self.meta["site_name"]
        self.mocap_jnt = np.array([jn.index(n) for n in self.meta["mocap_joints"]], dtype=np.int32)
        self.mocap_site = np.array([sn.index(n) for n in self.meta["mocap_sites"]], dtype=np.int32)
        self.mocap_qadr = self.m.jnt_qposadr[self.mocap_jnt]
        self.mocap_dadr = self.m.jnt_dofadr[self.mocap_jnt]
        # retract_wings (tasks/task_utils.py:117-122)
        self.wing_retract = {f"wing_{a}_{s}": v for a, v in (("roll", 0.7), ("pitch", -1.0), ("yaw", 1.5)) for s in ("left", "right")}
        self.retract_qadr = np.array([self.m.jnt_qposadr[jn.index(n)] for n in self.wing_retract], dtype=np.int32)
        self.retract_val = np.array(list(self.wing_retract.values()), dtype=np.float64)

    def full_qpos(self, snippet_row_qpos):
        """qpos of the whole model for one snippet row: the reference pose on the mocap joints, the defaults elsewhere, wings
        retracted (`walk_imitation.py:112-121`)."""
        q = self.m.qpos0.astype(np.float64).copy()
        q[:7] = snippet_row_qpos[:7]
        q[self.mocap_qadr] = snippet_row_qpos[7:]
        q[self.retract_qadr] = self.retract_val
        return q

    def site_xpos(self, k):
        b = self.m.sites_bodyid
        return np.array([k["xpos"][b[s]] + Q.rot(self.m.sites_pos[s], k["xquat"][b[s]]) for s in range(len(b))])


def quat_z2vec(vec):
    """`quaternions.py:205-249`: rotation taking the z axis to `vec` (single vector)."""
    v = np.asarray(vec, dtype=np.float64).copy()
    edge = v[0] == 0.0 and v[1] == 0.0
    if edge:
        v[0] = 1.0
    v = v / np.linalg.norm(v)
    axis = np.array([-v[1], v[0], 0.0])
    axis /= np.linalg.norm(axis)
    ang = np.arccos(v[2])
    q = np.hstack((np.cos(ang / 2), np.sin(ang / 2) * axis))
    if edge:
        q = np.array([0.0, 1.0, 0.0, 0.0]) if v[2] < 0 else np.array([1.0, 0.0, 0.0, 0.0])
    return q


def joint_orientation_quat(xaxis, qpos):
    """`quaternions.py:298-321`."""
    a = np.asarray(xaxis, dtype=np.float64)
    q2 = np.hstack((np.cos(qpos / 2), np.sin(qpos / 2) * a / np.linalg.norm(a)))
    return Q.mul(q2, quat_z2vec(a))


def walker_features(view: WalkModelView, qpos, qvel):
    """`tasks/rewards.py:36-61` get_walker_features on a full-model state."""
    k = kinematics(view.m, qpos)
    root_quat = qpos[3:7]
    rinv = Q.conj(root_quat) / np.dot(root_quat, root_quat)
    sites = view.site_xpos(k)[view.mocap_site]
    root2site = np.array([Q.rot(s - qpos[:3], Q.conj(root_quat)) for s in sites])
    jq = [np.asarray(root_quat, dtype=np.float64)]
    for j, qa in zip(view.mocap_jnt, view.mocap_qadr):
        jq.append(joint_orientation_quat(Q.rot(k["xaxis"][j], rinv), qpos[qa]))
    return {"com": np.asarray(qpos[:3], dtype=np.float64), "qvel": np.hstack((qvel[:6], qvel[view.mocap_dadr])),
            "root2site": root2site, "joint_quat": np.array(jq)}


class WalkRefSet:
    """Walking snippets of individual lengths, rows concatenated (the layout the device tables and the oracle take)."""

    def __init__(self, snippets):
        self.off = np.zeros(len(snippets) + 1, dtype=np.int32)
        self.off[1:] = np.cumsum([len(s["qpos"]) for s in snippets])
        cat = lambda key: np.ascontiguousarray(np.concatenate([np.asarray(s[key], dtype=np.float64) for s in snippets], axis=0))
        self.qpos, self.qvel, self.root2site, self.joint_quat = cat("qpos"), cat("qvel"), cat("root2site"), cat("joint_quat")

    @property
    def ntraj(self):
        return len(self.off) - 1

    def snippet(self, i):
        a, b = self.off[i], self.off[i + 1]
        return {"qpos": self.qpos[a:b], "qvel": self.qvel[a:b], "root2site": self.root2site[a:b], "joint_quat": self.joint_quat[a:b]}


def synthetic_snippets(view: WalkModelView, n: int = 4, length: int = 200, dt: float = 2e-3, seed: int = 0, speed: float = 1.0,
                       amplitude: float = 0.15):
    """Walking-like snippets for tests and benches: the root moves forward at `speed` cm/s at standing height while every
    tracked leg joint swings sinusoidally (tripod phases) about its default angle; the tracked features are computed from the
    pose by `walker_features`, so a walker placed exactly on a row earns the full reward."""
    rng = np.random.RandomState(seed)
    J = len(view.mocap_jnt)
    names = view.meta["mocap_joints"]
    tripod = np.array([0.0 if (("T1_left" in n_) or ("T2_right" in n_) or ("T3_left" in n_)) else np.pi for n_ in names])
    q0 = view.m.qpos0[view.mocap_qadr]
    rng_ = view.m.jnt_range[view.mocap_jnt]
    out = []
    for s in range(n):
        L = length + 17 * s  # individual lengths
        f = 8.0 + rng.rand() * 4.0  # stride frequency, Hz
        amp = amplitude * (0.5 + rng.rand(J))
        t = np.arange(L) * dt
        ang = q0[None, :] + amp[None, :] * np.sin(2 * np.pi * f * t[:, None] + tripod[None, :])
        ang = np.clip(ang, rng_[:, 0] + 1e-3, rng_[:, 1] - 1e-3)
        dang = np.gradient(ang, dt, axis=0)
        yaw = 0.3 * s
        rq = np.array([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)])
        qpos = np.zeros((L, 7 + J)); qvel = np.zeros((L, 6 + J))
        qpos[:, 0] = speed * t * np.cos(yaw); qpos[:, 1] = speed * t * np.sin(yaw); qpos[:, 2] = view.m.qpos0[2]
        qpos[:, 3:7] = rq
        qpos[:, 7:] = ang
        qvel[:, 0] = speed * np.cos(yaw); qvel[:, 1] = speed * np.sin(yaw)
        qvel[:, 6:] = dang
        r2s = np.zeros((L, len(view.mocap_site), 3)); jq = np.zeros((L, J, 4))
        for i in range(L):
            ft = walker_features(view, view.full_qpos(qpos[i]), np.zeros(view.nv))
            r2s[i] = ft["root2site"]; jq[i] = ft["joint_quat"][1:]
        out.append({"qpos": qpos, "qvel": qvel, "root2site": r2s, "joint_quat": jq})
    return out


def save_npz(path: str, snippets, joint_names, site_names, timestep_seconds: float = 2e-3):
    """Ragged walking snippets in one file: rows concatenated + `traj_off`; names of the tracked joints / sites as the dataset
    gives them (`trajectory_loaders.py:217-223`)."""
    r = WalkRefSet(snippets)
    np.savez_compressed(path, qpos=r.qpos, qvel=r.qvel, root2site=r.root2site, joint_quat=r.joint_quat, traj_off=r.off,
                        joint_names=np.array(list(joint_names)), site_names=np.array(list(site_names)),
                        timestep_seconds=np.float64(timestep_seconds))


def load_npz(path: str):
    """-> (WalkRefSet, joint_names, site_names, timestep_seconds)"""
    z = np.load(path, allow_pickle=False)
    off = z["traj_off"]
    sn = [{k: z[k][off[i]:off[i + 1]] for k in ("qpos", "qvel", "root2site", "joint_quat")} for i in range(len(off) - 1)]
    return WalkRefSet(sn), [str(x) for x in z["joint_names"]], [str(x) for x in z["site_names"]], float(z["timestep_seconds"])


def inference_snippet(qpos7, qvel6):
    """`InferenceWalkingTrajectoryLoader.set_next_trajectory` (`trajectory_loaders.py:226-254`): a root-only trajectory (qpos
    [T, 7], qvel [T, 6]) for the task's inference mode, where the reward is the constant 1 and only the root preview is used."""
    q, v = np.asarray(qpos7, dtype=np.float64), np.asarray(qvel6, dtype=np.float64)
    assert q.ndim == 2 and q.shape[1] == 7 and v.shape == (len(q), 6)
    T = len(q)
    return {"qpos": q, "qvel": v, "root2site": np.zeros((T, 0, 3)), "joint_quat": np.zeros((T, 0, 4))}
