"""Walking-imitation reward maths (`vnl_ray/tasks/rewards.py:9-111`), restated for the `walk_imitation` row of SURVEY.md
section 8(f): DeepMimic-style factors over four feature groups - CoM position, joint velocities, egocentric end-effector
vectors, joint orientation quaternions.  Pure numpy, batched over leading dimensions; pinned bit-level against the
imported reference module by `tests/golden/rewards.npz` (`tools/gen_golden.py`, `tests/test_golden_task_math.py`).
"""

from __future__ import annotations

import numpy as np

#: default per-feature standard deviations of the fruit-fly walking task (`rewards.py:96-102`)
DEFAULT_STD = {"com": 0.078487, "qvel": 53.7801, "root2site": 0.0735, "joint_quat": 1.2247}
FEATURES = ("com", "qvel", "root2site", "joint_quat")


def quat_dist_short_arc(quat1, quat2):
    """Shortest-arc angle between orientations (`quaternions.py:273-295`): acos(min(1, 2 (p.q)^2 - 1)) on normalised inputs."""
    q1 = np.asarray(quat1, dtype=np.float64)
    q2 = np.asarray(quat2, dtype=np.float64)
    q1 = q1 / np.linalg.norm(q1, axis=-1, keepdims=True)
    q2 = q2 / np.linalg.norm(q2, axis=-1, keepdims=True)
    x = 2 * np.sum(q1 * q2, axis=-1) ** 2 - 1
    return np.arccos(np.minimum(1.0, x))


def compute_diffs(walker_features, reference_features, n: int = 2):
    """`rewards.py:9-33`: per feature, sum |a - b|^n; for keys containing "quat", sum of short-arc distances^n."""
    diffs = {}
    for k in walker_features:
        if "quat" not in k:
            diffs[k] = np.sum(np.abs(walker_features[k] - reference_features[k]) ** n)
        else:
            diffs[k] = np.sum(quat_dist_short_arc(walker_features[k], reference_features[k]) ** n)
    return diffs


def get_reference_features(reference_data, step: int):
    """`rewards.py:62-79`: features of the reference at `step`; the root quaternion heads the joint quaternions."""
    qpos_ref = reference_data["qpos"][step, :]
    return {
        "com": reference_data["qpos"][step, :3],
        "qvel": reference_data["qvel"][step, :],
        "root2site": reference_data["root2site"][step, :],
        "joint_quat": np.vstack((qpos_ref[3:7], reference_data["joint_quat"][step, :])),
    }


def reward_factors_deep_mimic(walker_features, reference_features, std=None, weights=(1, 1, 1, 1)):
    """`rewards.py:82-111`: exp(-0.5 / std_k^2 * diff_k) per feature (in the walker dict's key order), times `weights`."""
    std = DEFAULT_STD if std is None else std
    diffs = compute_diffs(walker_features, reference_features, n=2)
    factors = np.array([np.exp(-0.5 / std[k] ** 2 * diffs[k]) for k in walker_features.keys()])
    return factors * np.asarray(weights)
