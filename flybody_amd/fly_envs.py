"""Environment factories with the reference's names and keyword meaning (`vnl_ray/fly_envs.py`), returning
batched MI355X environments instead of single-instance `composer.Environment`s.

Built so far: `flight_imitation` (`fly_envs.py:29-72`) and `walk_on_ball` (`fly_envs.py:125-157`).  `walk_imitation`,
`vision_guided_flight` and `template_task` are later rows of SURVEY.md section 8 and raise `NotImplementedError`.
"""

from __future__ import annotations

import numpy as np

from .batched_env import BatchedBallEnv, BatchedFlyEnv
from .tasks import synthetic, trajectories, wbpg


def flight_imitation(wpg_pattern_path: str | None = None, ref_path: str | None = None, random_state=None,
                     terminal_com_dist: float = 2.0, *, batch_size: int = 1, device: int = 0, env_id_base: int = 0,
                     **env_kwargs) -> BatchedFlyEnv:
    """Requires a fruitfly to track a flying reference (`fly_envs.py:29-72`).

    Args:
        wpg_pattern_path: `.npy` with one wing-beat cycle, shape (timesteps, 3) [yaw, roll, pitch], as the
            reference's WingBeatPatternGenerator expects.  None = the build's synthetic cycle.
        ref_path: `.npz` written by `tools/convert_hdf5_to_npz.py` from the reference's HDF5 layout
            (`tasks/trajectory_loaders.py:90-96`; h5py is unavailable here): `com_qpos (rows,7)`, `com_qvel (rows,6)`,
            `traj_off (N+1,)`, `timestep_seconds` - every trajectory keeps its own length, as in the reference.
            None = synthetic trajectories.
        random_state: int seed or `np.random.RandomState`; seeds the per-env counter-based generators that draw the
            trajectory index and the initial wing-beat phase of every episode.
        terminal_com_dist: episode terminates when the model-to-ghost CoM distance exceeds this (cm).
        batch_size, device, env_id_base: batching (env `i` of this handle is global env `env_id_base + i`).
    """
    if wpg_pattern_path is None:
        tables = wbpg.build_tables(synthetic.base_wing_pattern())
    else:
        tables = wbpg.load_tables(wpg_pattern_path)
    if ref_path is None:
        refs = trajectories.as_refset(*trajectories.preprocess(*synthetic.flight_trajectories()))
    else:
        com_qpos, com_qvel, dt = trajectories.load_npz(ref_path)
        if abs(dt - tables.dt_ctrl) > 1e-12:
            raise ValueError(f"trajectory timestep {dt} != control timestep {tables.dt_ctrl}")
        refs = trajectories.preprocess_ragged(com_qpos, com_qvel)
    if isinstance(random_state, np.random.RandomState):
        seed = int(random_state.randint(0, 2**31 - 1))
    else:
        seed = 0 if random_state is None else int(random_state)
    # fly_envs.py:54-65: time_limit 0.6 s, joint_filter 0 (compiled into the model), future_steps 5, initialize_qvel
    return BatchedFlyEnv(tables, refs, batch_size=batch_size, device=device, seed=seed, env_id_base=env_id_base,
                         future_steps=5, time_limit=0.6, terminal_com_dist=terminal_com_dist, **env_kwargs)


def walk_on_ball(random_state=None, *, batch_size: int = 1, device: int = 0, **env_kwargs) -> BatchedBallEnv:
    """Requires a tethered fruitfly to walk on a floating ball (`fly_envs.py:125-157`).

    The arena (ball at (-0.05, 0, -0.419), radius 0.454, density 0.0025), `joint_filter=0.01`, `adhesion_filter=0.007`
    and the claw friction are compiled into `assets/fly_ball.ffmb`; `time_limit` is 2.0 s.  Episodes carry no randomness
    (`walk_on_ball.py:48-54`), so `random_state` is accepted for signature parity and ignored."""
    del random_state
    return BatchedBallEnv(batch_size=batch_size, device=device, time_limit=2.0, **env_kwargs)


def walk_imitation(ref_path=None, random_state=None, terminal_com_dist: float = 0.3, **env_kwargs):
    """Requires a fruitfly to track a reference walking fly (`fly_envs.py:75-122`).

    Built so far: the compiled model (`assets/fly_walk.ffmb`: free root, floor plane, Walking
    configuration), the snippet layout and walker features (`tasks/walking.py`), the DeepMimic reward maths (`tasks/rewards.py`)
    and the float64 CPU restatement of the whole task that the tests check against (test infrastructure, outside this package).  The free-root contact step kernel is
    not written yet, and this package has no CPU path: the factory fails instead of returning a slow environment."""
    raise NotImplementedError("walk_imitation: model, reference layout, reward maths and oracle exist; the HIP step kernel "
                              "(free root + floor contacts) is not built yet - see DESIGN.md section 2 row f3")


def vision_guided_flight(*args, **kwargs):
    raise NotImplementedError("vision-guided flight needs rendering + heightfield collision; out of scope (SURVEY.md 8f)")


def template_task(*args, **kwargs):
    raise NotImplementedError("template_task is a reference test scaffold; out of scope (SURVEY.md section 2 row 12)")
