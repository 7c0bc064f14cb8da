"""Scalar-first quaternion / rotation helpers used by the host-side model compiler (float64).

Conventions follow MuJoCo's `mju_*` helpers that the reference calls through `mjlib`
(`fruitfly/fruitfly.py:35-53`): Hamilton product, `q = (w, x, y, z)`, `rot(v, q) = q v q*`.
"""

from __future__ import annotations

import numpy as np


def mul(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.array(
        [
            a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
        ]
    )


def neg(q):
    """The reference's `neg_quat`: flips the scalar part (`fruitfly.py:28-32`)."""
    q = np.array(q, dtype=np.float64)
    q[0] *= -1
    return q


def conj(q):
    q = np.array(q, dtype=np.float64)
    q[1:] *= -1
    return q


def normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    if n < 1e-15:
        return np.array([1.0, 0, 0, 0])
    return q / n


def to_mat(q):
    w, x, y, z = np.asarray(q, dtype=np.float64)
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def rot(v, q):
    """`mju_rotVecQuat`: the (unnormalised) sandwich product, as MuJoCo evaluates it."""
    return to_mat(q) @ np.asarray(v, dtype=np.float64)


def from_mat(m):
    m = np.asarray(m, dtype=np.float64)
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return normalize(q)


def axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = np.sin(angle / 2)
    return np.array([np.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def z_to_vec(vec):
    """`mju_quatZ2Vec`: rotation taking +z onto `vec`."""
    v = np.asarray(vec, dtype=np.float64)
    n = np.linalg.norm(v)
    if n < 1e-15:
        return np.array([1.0, 0, 0, 0])
    v = v / n
    axis = np.cross([0.0, 0, 1], v)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if v[2] > 0 else np.array([0.0, 1, 0, 0])
    axis = axis / s
    ang = np.arctan2(s, v[2])
    return axis_angle(axis, ang)


def from_euler_xyz(e):
    """MJCF `euler` with the default `eulerseq="xyz"` (intrinsic rotations)."""
    q = np.array([1.0, 0, 0, 0])
    for k in range(3):
        ax = np.zeros(3)
        ax[k] = 1
        q = mul(q, axis_angle(ax, e[k]))
    return q
