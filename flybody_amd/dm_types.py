"""The slice of `dm_env` the reference's actor loop relies on (`agents/ray_distributed_dmpo.py:315,399-407`),
re-declared because dm_env is not installed here; fields and semantics are dm_env's, with a leading batch dim."""

from __future__ import annotations

import enum
from typing import Any, NamedTuple

import numpy as np


class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2


class TimeStep(NamedTuple):
    """Batched `dm_env.TimeStep`: `step_type`, `reward`, `discount` are [B] tensors, `observation` an
    OrderedDict of [B, ...] tensors.  On FIRST rows reward is 0 and discount 1 (dm_env reports None)."""

    step_type: Any
    reward: Any
    discount: Any
    observation: Any

    def first(self):
        return self.step_type == StepType.FIRST

    def mid(self):
        return self.step_type == StepType.MID

    def last(self):
        return self.step_type == StepType.LAST


class Array:
    def __init__(self, shape, dtype, name=None):
        self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name

    def __repr__(self):
        return f"Array(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self.minimum = np.broadcast_to(np.asarray(minimum, dtype=dtype), self.shape).copy()
        self.maximum = np.broadcast_to(np.asarray(maximum, dtype=dtype), self.shape).copy()

    def __repr__(self):
        return f"BoundedArray(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"
