"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol include/aiptd.h declares,
keeps the reference's POD layouts, and fails loudly (no fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from ai_path_tracer_denoiser_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "aiptd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aipt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    L = api.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/aiptd.h but not exported by libaiptd.so"
    assert sorted(n for n, _, _ in api.ABI) == declared      # the Python binding covers the whole header


def test_pod_layouts_match_reference_structs():
    # sceneStructs.h sizes [probed] in SURVEY Appendix B
    assert C.sizeof(api.Geom) == 248 and C.sizeof(api.Face) == 76 and C.sizeof(api.Material) == 44
    assert C.sizeof(api.Camera) == 84 and C.sizeof(api.AABB) == 24
    assert api.Geom.transform.offset == 44 and api.Geom.inverseTransform.offset == 108
    assert api.Camera.pixelLength.offset == 76


def test_abi_version():
    assert api.lib().aipt_abi_version() == 1


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.AiptError) as e:
        api.Context(0)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)
