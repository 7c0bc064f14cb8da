"""Wing-beat pattern generator tables for the batched environment.

The reference builds, per env process, 201 pre-interpolated cyclic wing-angle tables (one per
beat frequency) in `WingBeatPatternGenerator.__init__` (`tasks/pattern_generators.py:18-119`) and
walks them with `reset`/`step` (`:121-191`).  Here the tables are built once on the host, packed
into flat arrays and uploaded read-only to HBM; the per-env walker state (`_step`, `_freq_idx`,
`_ctrl_freq`) lives in the device state records and is advanced inside the step kernel.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .constants import _FLY_CONTROL_TIMESTEP, _WING_PARAMS


@dataclass
class WingBeatTables:
    beat_freqs: np.ndarray  # (F,)
    tab_off: np.ndarray  # (F+1,) row offsets into traj/phase
    traj: np.ndarray  # (R, 6) wing angles [yaw, roll, pitch] x [left, right]
    phase: np.ndarray  # (R,)
    n_repeats: np.ndarray  # (F,)
    rel_errors: np.ndarray  # (F,)
    base_freq: float
    rel_range: float
    rate: float
    dt_ctrl: float

    def table(self, i: int):
        a, b = self.tab_off[i], self.tab_off[i + 1]
        return self.traj[a:b], self.phase[a:b]


def build_tables(base_pattern: np.ndarray, base_beat_freq=_WING_PARAMS["base_freq"],
                 rel_freq_range=_WING_PARAMS["rel_freq_range"], num_freqs=_WING_PARAMS["num_freqs"],
                 min_repeats: int = 10, max_repeats: int = 20, dt_ctrl: float = _FLY_CONTROL_TIMESTEP,
                 ctrl_filter: float = 0.5 / _WING_PARAMS["base_freq"]) -> WingBeatTables:
    """Same construction, same argument names and defaults as `pattern_generators.py:18-119`.

    `base_pattern` is one wing-beat cycle, shape (timesteps, 3) = yaw, roll, pitch."""
    pattern = np.tile(np.asarray(base_pattern, dtype=np.float64), (1, 2))  # both wings (`:55`)
    n_base = pattern.shape[0]
    rate = float(np.exp(-dt_ctrl / ctrl_filter)) if ctrl_filter != 0.0 else 0.0  # `:62`
    beat_freqs = np.linspace((1 - rel_freq_range) * base_beat_freq, (1 + rel_freq_range) * base_beat_freq, num_freqs)
    reps = np.arange(min_repeats, max_repeats + 1)
    trajs, phases, n_repeats, rel_errors = [], [], [], []
    for beat_freq in beat_freqs:
        beat_time = 1 / beat_freq
        rel_error = ((reps * beat_time) % dt_ctrl) / dt_ctrl  # `:82`
        over, under = int(np.argmin(rel_error)), int(np.argmin(np.abs(1 - rel_error)))
        if rel_error[over] < np.abs(1 - rel_error[under]):
            pick, shift = over, dt_ctrl
        else:
            pick, shift = under, 0.0
        # The reference takes the *position* inside arange(min, max+1) plus one as the repeat count
        # (`:93`), not reps[pick]; reproduced as is.
        n_reps = pick + 1
        rel_errors.append(rel_error[pick])
        n_repeats.append(n_reps)
        repeated = np.tile(pattern, (n_reps, 1))
        phase = np.linspace(0, n_reps, n_reps * n_base, endpoint=False)
        dt_data = beat_time / n_base
        duration = repeated.shape[0] * dt_data
        t_data = np.linspace(0, duration, repeated.shape[0])
        t_ctrl = np.arange(0, duration - shift, dt_ctrl)
        tr = np.stack([np.interp(t_ctrl, t_data, repeated[:, i]) for i in range(repeated.shape[1])], axis=1)
        trajs.append(tr)
        phases.append(np.interp(t_ctrl, t_data, phase))
    tab_off = np.zeros(num_freqs + 1, dtype=np.int32)
    tab_off[1:] = np.cumsum([len(p) for p in phases])
    return WingBeatTables(beat_freqs=beat_freqs, tab_off=tab_off, traj=np.concatenate(trajs, 0),
                          phase=np.concatenate(phases, 0), n_repeats=np.array(n_repeats, dtype=np.int32),
                          rel_errors=np.array(rel_errors), base_freq=float(base_beat_freq),
                          rel_range=float(rel_freq_range), rate=rate, dt_ctrl=float(dt_ctrl))


def load_tables(base_pattern_path: str, **kw) -> WingBeatTables:
    """`base_pattern_path` as in the reference: a `.npy` of one cycle, shape (timesteps, 3)."""
    with open(base_pattern_path, "rb") as f:
        return build_tables(np.load(f), **kw)
