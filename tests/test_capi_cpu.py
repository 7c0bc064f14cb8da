"""CPU-side checks of the drop-in boundary: the C ABI library builds for gfx950, loads, and exports every
symbol include/flybody_env.h declares; the Python host mirrors the reference's factory names."""
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from flybody_amd import _capi, build

    build.build()
    L = _capi.lib()
    hdr = open(os.path.join(ROOT, "include", "flybody_env.h")).read()
    declared = set(re.findall(r"\b(ffe_[a-z_]+)\s*\(", hdr))
    assert declared == set(_capi.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.ffe_version()


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from flybody_amd import fly_envs

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fly_envs.flight_imitation(batch_size=4)


def test_factory_surface_matches_reference_names():
    from flybody_amd import fly_envs

    for name in ("flight_imitation", "walk_imitation", "walk_on_ball", "vision_guided_flight", "template_task"):
        assert callable(getattr(fly_envs, name))
    import inspect

    params = list(inspect.signature(fly_envs.flight_imitation).parameters)
    assert params[:4] == ["wpg_pattern_path", "ref_path", "random_state", "terminal_com_dist"]


def test_product_path_does_not_touch_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "flybody_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"(from|import)\s+oracle|fly_oracle|oracle/", txt):
                    bad.append(f)
    assert not bad, bad


def test_hdf5_converter_keeps_individual_lengths(tmp_path):
    """tools/convert_hdf5_to_npz.py (h5py is absent here, so its layout logic is driven with an in-memory stand-in for the
    HDF5 groups): every trajectory keeps its own length, as the reference serves them (trajectory_loaders.py:98-100)."""
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_hdf5_to_npz as conv
    from flybody_amd.tasks import trajectories as T

    rng = np.random.RandomState(0)
    lens = [40, 7, 25, 3006]
    data = {}
    for i, n in enumerate(lens):
        q = rng.randn(n, 7); q[:, 3:] /= np.linalg.norm(q[:, 3:], axis=1, keepdims=True)
        data[str(i)] = (q, rng.randn(n, 6))
    dst = str(tmp_path / "set.npz")
    kept, lo, hi = conv.convert(lambda k: data[k], [str(i) for i in range(len(lens))], 2e-4, dst, min_len=8)
    assert (kept, lo, hi) == (3, 25, 3006)   # the 7-step trajectory cannot host an episode (future_steps + 2 = 7 < 8)
    qs, vs, dt = T.load_npz(dst)
    assert [len(q) for q in qs] == [40, 25, 3006] and dt == 2e-4
    assert np.array_equal(qs[1], data["2"][0]) and np.array_equal(vs[2], data["3"][1])
    refs = T.preprocess_ragged(qs, vs)
    assert refs.ntraj == 3 and refs.lengths().tolist() == [40, 25, 3006] and refs.off.tolist() == [0, 40, 65, 3071]
    # per-trajectory preprocessing == the stacked one on a trajectory of its own (xy re-centred on its first sample)
    rq, rv = T.preprocess(qs[1][None], vs[1][None])
    assert np.array_equal(refs.trajectory(1)[0], rq[0]) and np.array_equal(refs.trajectory(1)[1], rv[0])
    assert np.abs(T.root2com(refs.trajectory(2)[0])[0, :2]).max() < 1e-12
