"""The constant data the reference itself holds for this path (tests/golden/reference_tables.npz,
parsed out of the reference's source text by tests/golden/make_reference_tables.py) against the
copies this repository computes with: the pattern header, the oracle's tables, the Python mirror's
constants and, on a GPU, the tables in device memory."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))


def test_fixture_is_the_published_orb_pattern():
    p = REF["bit_pattern_31"]
    assert p.shape == (1024,) and list(p[:8]) == [8, -3, 9, 5, 4, 2, 7, -12]
    assert p.min() >= -13 and p.max() <= 13            # every test point lies inside the 31 x 31 patch
    assert list(REF["patch"]) == [31, 15, 19] and list(REF["matcher"]) == [100, 50, 30] and list(REF["grid"]) == [48, 64]


def test_pattern_header_equals_reference():
    text = open(os.path.join(ROOT, "include", "orbx_pattern_31.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    body = text[text.index("{") + 1:text.rindex("}")]
    vals = np.array([int(t) for t in re.findall(r"-?\d+", body)], np.int32)
    assert np.array_equal(vals, REF["bit_pattern_31"])


def test_oracle_tables_equal_reference(oracle):
    t = oracle.tables(oracle.params())
    assert np.array_equal(t["umax"], REF["umax"])
    # the oracle's descriptor reads the pattern: a one-pixel image probes single tests.  A bright pixel at
    # pattern point (x1, y1) of test t relative to the keypoint, angle 0, sets bit t (p0 < p1).
    pat = REF["bit_pattern_31"].reshape(256, 4)
    img = np.zeros((64, 64), np.uint8)
    for tst in (0, 1, 17, 100, 255):
        x0, y0, x1, y1 = pat[tst]
        im = img.copy()
        im[32 + y1, 32 + x1] = 255
        d = oracle.descriptor(im, 32, 32, 0.0)
        bits = np.unpackbits(d, bitorder="little")
        assert bits[tst] == 1
        hit = [k for k in range(256) if tuple(pat[k][2:]) == (x1, y1) and tuple(pat[k][:2]) != (x1, y1)]
        assert set(np.nonzero(bits)[0]) == set(hit)


def test_mirror_constants_equal_reference():
    from slam_amd.matcher import ORBmatcher
    th_high, th_low, histo = (int(v) for v in REF["matcher"])
    assert (ORBmatcher.TH_HIGH, ORBmatcher.TH_LOW, ORBmatcher.HISTO_LENGTH) == (th_high, th_low, histo)
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    assert int(re.search(r"#define ORBM_GRID_ROWS (\d+)", hdr).group(1)) == REF["grid"][0]
    assert int(re.search(r"#define ORBM_GRID_COLS (\d+)", hdr).group(1)) == REF["grid"][1]
    shim = open(os.path.join(ROOT, "include", "orbx_shim.hpp")).read()
    assert "TH_LOW = %d" % th_low in shim and "TH_HIGH = %d" % th_high in shim and "HISTO_LENGTH = %d" % histo in shim


@pytest.mark.gpu
def test_device_tables_equal_reference(extractor_vga):
    t = extractor_vga.debug_tables()
    pat = t[496:752].view(np.int8).reshape(256, 4).astype(np.int32)
    assert np.array_equal(pat.reshape(-1), REF["bit_pattern_31"])
    member = t[248:496].view(np.uint8).reshape(31, 32)      # rows v = -15..15, columns u = -16..15
    weight = t[0:248].view(np.uint8).reshape(31, 32)
    umax = REF["umax"]
    for v in range(-15, 16):
        want = np.array([1 if abs(u) <= umax[abs(v)] else 0 for u in range(-16, 16)], np.uint8)
        assert np.array_equal(member[v + 15], want), v
        assert np.array_equal(weight[v + 15], want * (np.arange(-16, 16) + 15).astype(np.uint8)), v
    assert member.sum() == 749
