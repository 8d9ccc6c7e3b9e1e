"""Known-answer tests that pin the CPU oracle independently of itself.

The reference ships no fixtures for this path, so these constants come from
the reference's formulas evaluated by hand / stated in its comments, and from
the published definitions of the OpenCV primitives it calls.
"""
import math
import subprocess
import os

import numpy as np
import pytest

from slam_amd import synth


def test_constructor_tables(oracle):
    t = oracle.tables(oracle.params(1000, 1.2, 8, 20, 7))
    # reference src/ORBextractor.cc:426-440 evaluated for 1000 features / 8 levels / 1.2
    assert t["quotas"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.tables(oracle.params(2000))["quotas"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    # reference :450-465 (radius-15 disc, symmetric)
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * u + 1 for u in t["umax"].tolist()) * 2 - (2 * 15 + 1) == 749
    s = np.float32(1.0)
    for i in range(8):
        assert t["scale"][i] == s
        assert t["inv_scale"][i] == np.float32(1.0) / s
        assert t["sigma2"][i] == s * s
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))


def test_level_sizes_and_cells(oracle):
    p = oracle.params()
    assert [oracle.level_size(p, 640, 480, l) for l in range(8)] == [
        (640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert [oracle.level_size(p, 1241, 376, l) for l in range(8)] == [
        (1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    # cell grids of reference :793-797 for 640x480
    cells = []
    for l in range(8):
        w, h = oracle.level_size(p, 640, 480, l)
        width, height = w - 32, h - 32
        ncols, nrows = int(width / 30.0), int(height / 30.0)
        cells.append((ncols, nrows, math.ceil(width / ncols), math.ceil(height / nrows)))
    assert cells == [(20, 14, 31, 32), (16, 12, 32, 31), (13, 10, 32, 31), (11, 8, 31, 31),
                     (9, 6, 31, 34), (7, 5, 33, 33), (6, 4, 31, 33), (4, 3, 37, 34)]
    assert sum(c[0] * c[1] for c in cells) == 815


def test_cv_round_is_half_to_even(oracle):
    assert [oracle.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001, -2.5)] == \
           [0, 2, 2, 0, -2, 2, 3, -2]


def test_fast_atan2(oracle):
    # exact on the axes, within the published 0.3 degree error elsewhere, range [0, 360)
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 5.0) == 0.0
    assert abs(oracle.fast_atan2(5.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0.0, -5.0) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-5.0, 0.0) - 270.0) < 1e-4
    for y in range(-40, 41, 7):
        for x in range(-40, 41, 9):
            if x == 0 and y == 0:
                continue
            a = oracle.fast_atan2(float(y), float(x))
            ref = math.degrees(math.atan2(y, x)) % 360.0
            assert 0.0 <= a <= 360.0
            assert min(abs(a - ref), 360.0 - abs(a - ref)) < 0.3
    # first polynomial coefficient: 0.9997878412794807f * (float)(180/pi), each factor rounded to float
    c = np.float32(0.9997878412794807) * np.float32(180.0 / math.pi)
    assert abs(float(c) - 57.283627) < 1e-5


def test_gaussian_kernel_integers(oracle):
    # getGaussianKernel(7, 2) scaled by 256 and rounded: 18 34 49 55 49 34 18 (sum 257)
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0)
    g /= g.sum()
    assert np.rint(g * 256).astype(int).tolist() == [18, 34, 49, 55, 49, 34, 18]
    imp = np.zeros((15, 15), np.uint8)
    imp[7, 7] = 255
    out = oracle.gaussian7(imp).astype(int)
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    expect = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[4:11, 4:11], expect)
    flat = np.full((20, 31), 200, np.uint8)
    assert (oracle.gaussian7(flat) == min(255, (200 * 257 * 257 + 32768) >> 16)).all()
    assert (oracle.gaussian7(flat, 1) == 200).all()  # the 256-sum variant preserves constants
    # reflect-101 border: a vertical step stays a function of x only
    step = np.zeros((12, 24), np.uint8)
    step[:, 12:] = 180
    b = oracle.gaussian7(step)
    assert (b == b[0:1, :]).all()


def test_resize_properties(oracle):
    flat = np.full((120, 160), 77, np.uint8)
    assert (oracle.resize_linear(flat, 133, 100) == 77).all()
    ramp = np.tile(np.arange(0, 240, dtype=np.uint8), (50, 1))
    r = oracle.resize_linear(ramp, 200, 42).astype(int)
    # the two truncating shifts of the vertical pass may move a value by one between rows
    assert (np.diff(r, axis=1) >= -1).all() and np.abs(r - r[0:1]).max() <= 1
    # exact 2:1 reduction samples midway between two pixels: (a+b+c+d+2)>>2 on a 2x2 block
    img = synth.synth_image(64, 48, 3)
    half = oracle.resize_linear(img, 32, 24).astype(int)
    i = img.astype(int)
    expect = (i[0::2, 0::2] + i[0::2, 1::2] + i[1::2, 0::2] + i[1::2, 1::2] + 2) >> 2
    assert np.abs(half - expect).max() <= 1
    assert np.array_equal(oracle.resize_linear(img, 64, 48), img)


def fast_reference(roi, thr):
    """Independent statement of FAST-9/16 + 3x3 NMS from its definition."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3),
            (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = roi.shape
    score = np.zeros((h, w), int)
    r = roi.astype(int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = r[y, x]
            d = [v - r[y + dy, x + dx] for dx, dy in ring]
            best = -999
            for k in range(16):
                arc = [d[(k + i) % 16] for i in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > thr:
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0 and all(s > score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                             if (dx, dy) != (0, 0)):
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,thr", [(1, 20), (2, 7), (3, 40)])
def test_fast_matches_definition(oracle, seed, thr):
    roi = synth.synth_image(43, 40, seed)
    xs, ys, sc = oracle.fast9_nms(roi, thr)
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == fast_reference(roi, thr)


def test_fast_on_synthetic_corner(oracle):
    roi = np.full((21, 21), 50, np.uint8)
    roi[10:, 10:] = 200  # one bright quadrant: its corner pixel is a FAST corner
    roi[10, 10] = 230    # make it the strict maximum (equal scores suppress each other)
    xs, ys, sc = oracle.fast9_nms(roi, 20)
    assert len(xs) >= 1 and (10, 10) in set(zip(xs.tolist(), ys.tolist()))
    assert len(oracle.fast9_nms(np.full((21, 21), 50, np.uint8), 7)[0]) == 0
    assert len(oracle.fast9_nms(np.zeros((6, 30), np.uint8), 7)[0]) == 0  # no interior


def test_distribute_small_cases(oracle):
    # fewer candidates than the target: every candidate in a distinct node survives
    xs = np.array([5, 100, 300, 590], np.int32)
    ys = np.array([5, 200, 40, 430], np.int32)
    rs = np.array([10, 20, 30, 40], np.int32)
    sel = oracle.distribute(xs, ys, rs, 16, 624, 16, 464, 50)
    assert sorted(sel.tolist()) == [0, 1, 2, 3]
    # two candidates that can never be separated keep the stronger, first on ties
    xs = np.array([10, 10, 400], np.int32)
    ys = np.array([10, 10, 400], np.int32)
    for rs_, keep in (([5, 9, 1], 1), ([9, 9, 1], 0)):
        sel = oracle.distribute(xs, ys, np.array(rs_, np.int32), 16, 624, 16, 464, 10)
        assert sorted(sel.tolist()) == sorted([keep, 2])
    assert len(oracle.distribute(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                                 16, 624, 16, 464, 10)) == 0


def test_distribute_respects_quota_bounds(oracle):
    p = oracle.params()
    img = synth.synth_image(640, 480, 12)
    xs, ys, rs = oracle.detect_cells(img)
    assert len(xs) > 1000
    for n in (1, 50, 217, 600):
        sel = oracle.distribute(xs, ys, rs, 16, 624, 16, 464, n)
        assert n <= len(sel) <= n + 3
        assert len(set(sel.tolist())) == len(sel)


def test_extract_invariants(oracle):
    p = oracle.params()
    img = synth.synth_image(640, 480, 4)
    k, d = oracle.extract(p, img)
    assert 1000 <= len(k) <= 1024
    assert (np.diff(k["octave"]) >= 0).all()          # levels concatenated 0..7
    sizes = {0: 31, 1: 37, 2: 44, 3: 53, 4: 64, 5: 77, 6: 92, 7: 111}
    assert all(sizes[o] == s for o, s in zip(k["octave"].tolist(), k["size"].tolist()))
    assert (k["class_id"] == -1).all() and (k["angle"] >= 0).all() and (k["angle"] < 360.0001).all()
    sc = oracle.tables(p)["scale"]
    lx = k["x"] / sc[k["octave"]]
    ws = np.array([oracle.level_size(p, 640, 480, l)[0] for l in range(8)])[k["octave"]]
    assert (lx > 18.99).all() and (lx < ws - 19 + 0.01).all()  # EDGE_THRESHOLD
    k2, d2 = oracle.extract(p, img)
    assert np.array_equal(k, k2) and np.array_equal(d, d2)


def test_descriptor_of_flat_patch_is_zero(oracle):
    flat = np.full((80, 80), 123, np.uint8)
    assert (oracle.descriptor(flat, 40, 40, 33.0) == 0).all()
    grad = np.tile(np.arange(80, dtype=np.uint8) * 3, (80, 1))
    d0 = oracle.descriptor(grad, 40, 40, 0.0)
    d180 = oracle.descriptor(grad, 40, 40, 180.0)
    assert d0.any() and not np.array_equal(d0, d180)


def test_ic_angle_directions(oracle):
    for ang, (dx, dy) in {0: (1, 0), 90: (0, 1), 180: (-1, 0), 270: (0, -1)}.items():
        yy, xx = np.mgrid[0:61, 0:61]
        img = np.clip(128 + 3 * ((xx - 30) * dx + (yy - 30) * dy), 0, 255).astype(np.uint8)
        a = oracle.ic_angle(img, 30, 30)
        assert min(abs(a - ang), 360 - abs(a - ang)) < 0.5


def test_hamming(oracle):
    a = np.zeros(32, np.uint8)
    b = np.full(32, 255, np.uint8)
    assert oracle.hamming256(a, a) == 0 and oracle.hamming256(a, b) == 256
    c = a.copy()
    c[[0, 7, 31]] = [1, 0x80, 0xF0]
    assert oracle.hamming256(a, c) == 6
    d = synth.synth_descriptors(50, 3)
    for i in range(49):
        assert oracle.hamming256(d[i], d[i + 1]) == int(np.unpackbits(d[i] ^ d[i + 1]).sum())


def test_three_maxima(oracle):
    h = [0] * 30
    assert oracle.three_maxima(h) == (-1, -1, -1)
    h[4], h[9], h[20] = 50, 30, 10
    assert oracle.three_maxima(h) == (4, 9, 20)
    h[9], h[20] = 4, 3            # second < 0.1 * first: only the best survives
    assert oracle.three_maxima(h) == (4, -1, -1)
    h[9], h[20] = 30, 4           # third < 0.1 * first
    assert oracle.three_maxima(h) == (4, 9, -1)
    h[9], h[20] = 5, 5            # 5 is not < 0.1f * 50 (= 5.0): both stay; ties keep the first seen
    assert oracle.three_maxima(h) == (4, 9, 20)


def test_grid_rounding(oracle):
    from oracle.pyoracle import KP_DTYPE, FrameData
    k = np.zeros(4, KP_DTYPE)
    k["x"] = [0.0, 4.99, 5.0, 639.9]   # cell width 10: round(0.499)=0, round(0.5)=1, round(63.99)=64 -> dropped
    k["y"] = [0.0, 0.0, 0.0, 0.0]
    f = FrameData(k, np.zeros((4, 32), np.uint8), (0, 0, 640, 480), np.ones(8, np.float32))
    start, items = oracle.grid_build(f)
    assert start[-1] == 3
    assert items.tolist() == [0, 1, 2]
    assert start[0] == 0 and start[1] == 2 and start[48] == 2 and start[49] == 3


def test_sincosf_restatement_matches_libm():
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-s", "-C", here])
    stride = "1" if os.environ.get("ORBX_EXHAUSTIVE") else "211"
    out = subprocess.run([os.path.join(here, "check_sincosf"), stride], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "bad_sin 0 bad_cos 0" in out.stdout
