"""The C-ABI library: it loads, exports every symbol include/orbx.h declares, and
refuses to work without a GPU (no CPU fallback).  No kernels run here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "orbx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orb[xm]_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("orbx_create", "orbx_destroy", "orbx_extract", "orbx_extract_batch", "orbx_get_level",
                 "orbx_scale_factors", "orbm_hamming256", "orbm_grid_build", "orbm_window_candidates",
                 "orbm_search_for_initialization", "orbm_search_by_projection", "orbm_compute_stereo_matches", "orbm_knn2",
                 "orbm_knn2_merge"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from slam_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "liborbx.so does not export %s" % name


def test_python_binding_covers_the_header():
    from slam_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load()


def test_keypoint_layout_matches_cv_keypoint():
    from slam_amd._lib import KP_DTYPE
    assert KP_DTYPE.itemsize == 28
    assert [KP_DTYPE.fields[n][1] for n in KP_DTYPE.names] == [0, 4, 8, 12, 16, 20, 24]


def test_host_hamming():
    import numpy as np
    from slam_amd.matcher import ORBmatcher
    a = np.zeros(32, np.uint8)
    b = np.zeros(32, np.uint8)
    b[3] = 0b1011
    assert ORBmatcher.DescriptorDistance(a, b) == 3


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-device failure path")
def test_create_fails_loudly_without_a_device():
    from slam_amd._lib import OrbxError
    from slam_amd.extractor import ORBextractor
    from slam_amd.matcher import ORBmatcher
    with pytest.raises(OrbxError) as e:
        ORBextractor(1000, 1.2, 8, 20, 7)
    assert "-2" in str(e.value) or "device" in str(e.value).lower()
    with pytest.raises(OrbxError):
        ORBmatcher(0.6, True)


def test_invalid_configuration_is_rejected():
    from slam_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.Config(1000, 1.0, 8, 20, 7, 640, 480, 1, 0, 0)   # scale factor must exceed 1
    assert lib.orbx_create(C.byref(cfg), C.byref(h)) == -1
    cfg = _lib.Config(1000, 1.2, 17, 20, 7, 640, 480, 1, 0, 0)  # too many levels
    assert lib.orbx_create(C.byref(cfg), C.byref(h)) == -1
    assert lib.orbx_last_error()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in text.lower() or fn == "synth.py", "%s mentions the oracle" % fn
    for fn in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", fn)).read().lower()
