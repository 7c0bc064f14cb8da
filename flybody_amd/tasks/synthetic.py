"""Synthetic task inputs (the reference ships neither a wing-beat pattern nor trajectory data).

Shapes and statistics follow SURVEY.md section 8(d): a closed-form 100-sample wing-beat cycle inside
the wing joint ranges (`fruitfly.xml:60,64,68`) and seeded forward-flight CoM trajectories at the
flight control step (2e-4 s) with the hover pitch of 47.5 degrees.
"""

from __future__ import annotations

import numpy as np

from .constants import _BODY_PITCH_ANGLE, _FLY_CONTROL_TIMESTEP


def base_wing_pattern(n: int = 100) -> np.ndarray:
    """One wing-beat cycle, (n, 3) = [yaw, roll, pitch] in radians."""
    phi = 2 * np.pi * np.arange(n) / n
    yaw = 1.2 * np.cos(phi)
    roll = 0.2 + 0.15 * np.sin(2 * phi)
    pitch = 0.8 + 1.0 * np.sin(phi)
    return np.stack([yaw, roll, pitch], axis=1)


def flight_trajectories(n_traj: int = 64, n_steps: int = 3006, dt: float = _FLY_CONTROL_TIMESTEP, seed0: int = 0):
    """Returns com_qpos (N,T,7), com_qvel (N,T,6): constant-speed flight along a slowly yawing heading."""
    qpos = np.zeros((n_traj, n_steps, 7))
    qvel = np.zeros((n_traj, n_steps, 6))
    pitch = np.deg2rad(_BODY_PITCH_ANGLE)
    t = np.arange(n_steps) * dt
    for k in range(n_traj):
        rng = np.random.RandomState(seed0 + k)
        speed = rng.uniform(20, 40)
        yaw_rate = rng.uniform(-2, 2)
        z0 = rng.uniform(0.5, 0.8)
        heading0 = rng.uniform(-np.pi, np.pi)
        heading = heading0 + yaw_rate * t
        vx, vy = speed * np.cos(heading), speed * np.sin(heading)
        x = np.concatenate(([0.0], np.cumsum(0.5 * (vx[1:] + vx[:-1]) * dt)))
        y = np.concatenate(([0.0], np.cumsum(0.5 * (vy[1:] + vy[:-1]) * dt)))
        qpos[k, :, 0], qpos[k, :, 1], qpos[k, :, 2] = x, y, z0
        # orientation: yaw(heading) * pitch-up(-pitch about y)
        qy = np.stack([np.cos(heading / 2), 0 * t, 0 * t, np.sin(heading / 2)], axis=1)
        qp = np.array([np.cos(pitch / 2), 0, -np.sin(pitch / 2), 0])
        w1, x1, y1, z1 = qy.T
        w2, x2, y2, z2 = qp
        qpos[k, :, 3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
        qpos[k, :, 4] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
        qpos[k, :, 5] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
        qpos[k, :, 6] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
        qvel[k, :, 0], qvel[k, :, 1] = vx, vy
        # body-frame angular velocity of a pure world-z yaw rate: R^T [0,0,yaw_rate]
        qvel[k, :, 3] = yaw_rate * np.sin(pitch)
        qvel[k, :, 5] = yaw_rate * np.cos(pitch)
    return qpos, qvel
