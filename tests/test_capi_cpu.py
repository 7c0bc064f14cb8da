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
