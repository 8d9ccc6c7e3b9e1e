"""Whole-image cross-checks of the oracle's image primitives against statements of the same
arithmetic that share no code with it: scipy's integer correlation for the 7x7 blur, a vectorised
numpy statement of the fixed-point bilinear resize, torch's float bilinear interpolation (within one
grey level), and numpy statements of the orientation moments and the descriptor tests.

These pin the oracle to the published algorithms; they do not pin it to an OpenCV binary (none is
available, and the reference holds no fixtures for this path)."""
import numpy as np
import pytest

from slam_amd import synth


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle
    return pyoracle


IMAGES = [(640, 480, 11), (533, 400, 12), (179, 134, 13), (61, 47, 14), (9, 8, 15)]


@pytest.mark.parametrize("w,h,seed", IMAGES)
@pytest.mark.parametrize("variant,kernel", [(0, [18, 34, 49, 55, 49, 34, 18]), (1, [18, 34, 48, 56, 48, 34, 18])])
def test_blur_equals_integer_correlation(oracle, w, h, seed, variant, kernel):
    """(sum_y k_y sum_x k_x p + 32768) >> 16 with reflect-101 borders: scipy's `mirror` mode."""
    from scipy.ndimage import correlate1d
    img = synth.synth_image(w, h, seed)
    k = np.array(kernel, np.int64)
    rows = correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")
    full = correlate1d(rows, k, axis=0, mode="mirror")
    expect = np.minimum((full + 32768) >> 16, 255).astype(np.uint8)
    assert np.array_equal(oracle.gaussian7(img, variant), expect)


@pytest.mark.parametrize("w,h,seed", IMAGES[:4])
def test_blur_float_column_variant(oracle, w, h, seed):
    """Variant 2: the same integer sums, divided by 2^16 and rounded to nearest even (the last w % 4
    columns half up): differs from variant 0 only where the sum sits exactly on a half."""
    from scipy.ndimage import correlate1d
    img = synth.synth_image(w, h, seed)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    full = correlate1d(correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    q, r = full >> 16, full & 0xFFFF
    even = q + ((r > 0x8000) | ((r == 0x8000) & ((q & 1) == 1)))
    half_up = (full + 32768) >> 16
    expect = even.copy()
    tail = w - (w % 4)
    expect[:, tail:] = half_up[:, tail:]
    assert np.array_equal(oracle.gaussian7(img, 2), np.minimum(expect, 255).astype(np.uint8))


def _resize_numpy(src, dw, dh):
    """cv::resize, 8-bit INTER_LINEAR, generic fixed-point path, written with whole-array operations:
    source position fx = (dx + 0.5) * scale - 0.5 in single precision, 11-bit weights rounded to
    nearest even, columns clamped with the weight moved to the kept tap, rows clamped by index only,
    vertical combine ((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2 >> 2."""
    sh, sw = src.shape

    def taps(dn, sn, zero_weight_at_edges):
        scale = 1.0 / (float(dn) / float(sn))
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if zero_weight_at_edges:
            lo, hi = s < 0, s >= sn - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, sn - 1, s))
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, w0, w1

    sx, a0, a1 = taps(dw, sw, True)
    sy, b0, b1 = taps(dh, sh, False)
    s = src.astype(np.int64)
    x1 = np.minimum(sx + 1, sw - 1)
    t = s[:, sx] * a0[None, :] + s[:, x1] * a1[None, :]          # (sh, dw) horizontal sums
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    top = (b0[:, None] * (t[y0] >> 4)) >> 16
    bot = (b1[:, None] * (t[y1] >> 4)) >> 16
    return ((top + bot + 2) >> 2).astype(np.uint8)


@pytest.mark.parametrize("w,h,dw,dh,seed", [(640, 480, 533, 400, 21), (533, 400, 444, 333, 22), (214, 161, 179, 134, 23),
                                            (1241, 376, 1034, 313, 24), (100, 80, 50, 40, 25), (97, 61, 81, 51, 26),
                                            (64, 48, 64, 48, 27), (640, 480, 320, 200, 28)])
def test_resize_equals_vectorised_statement(oracle, w, h, dw, dh, seed):
    img = synth.synth_image(w, h, seed)
    assert np.array_equal(oracle.resize_linear(img, dw, dh), _resize_numpy(img, dw, dh))


@pytest.mark.parametrize("w,h,dw,dh,seed", [(640, 480, 533, 400, 31), (309, 231, 257, 193, 32)])
def test_resize_within_one_level_of_float_bilinear(oracle, w, h, dw, dh, seed):
    """Half-pixel-centre bilinear interpolation in float (torch, align_corners=False) is the
    real-valued function the fixed-point path approximates."""
    import torch
    img = synth.synth_image(w, h, seed)
    t = torch.from_numpy(img.astype(np.float32))[None, None]
    ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
    got = oracle.resize_linear(img, dw, dh).astype(np.float32)
    assert np.abs(got - ref).max() <= 1.0 + 1e-3


def test_pyramid_levels_are_successive_resizes(oracle):
    """ComputePyramid (reference src/ORBextractor.cc:1117-1145): level l is the resize of level l - 1
    to cvRound(size / scale_l), not of level 0."""
    p = oracle.params()
    img = synth.synth_image(640, 480, 41)
    sizes = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    prev = img
    for lvl in range(1, 8):
        w, h = sizes[lvl]
        cur = _resize_numpy(prev, w, h)
        assert np.array_equal(oracle.pyramid_level(p, img, lvl), cur), lvl
        prev = cur


def _tables():
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_tables.npz"))
    return d["bit_pattern_31"].reshape(256, 4), d["umax"]


def test_orientation_moments_from_the_reference_tables(oracle):
    """IC_Angle (reference src/ORBextractor.cc:59-88): m10 = sum u I, m01 = sum v I over the disc
    rows |u| <= umax[|v|]; the angle is fastAtan2(m01, m10), checked here against atan2 in double
    precision (the polynomial's published error bound is 0.3 degrees)."""
    _, umax = _tables()
    img = synth.synth_image(200, 160, 51)
    rng = np.random.default_rng(5)
    I = img.astype(np.int64)
    for _ in range(200):
        x, y = int(rng.integers(20, 180)), int(rng.integers(20, 140))
        m10 = m01 = 0
        for v in range(-15, 16):
            d = int(umax[abs(v)])
            row = I[y + v, x - d:x + d + 1]
            m10 += int((np.arange(-d, d + 1) * row).sum())
            m01 += v * int(row.sum())
        want = np.degrees(np.arctan2(m01, m10)) % 360.0
        got = oracle.ic_angle(img, x, y)
        assert abs((got - want + 180.0) % 360.0 - 180.0) < 0.31, (x, y, got, want)
        assert got == oracle.fast_atan2(np.float32(m01), np.float32(m10))


def test_descriptor_from_the_reference_pattern(oracle):
    """computeOrbDescriptor (reference src/ORBextractor.cc:92-131) with the pattern read from the
    reference's source text: bit i of the descriptor = I(p0_i) < I(p1_i), each point steered by
    the keypoint angle, coordinates rounded to nearest even, in single precision and in the order
    written (x*b + y*a, x*a - y*b; no fused multiply-add)."""
    pat, _ = _tables()
    img = oracle.gaussian7(synth.synth_image(160, 120, 61))
    rng = np.random.default_rng(6)
    f32 = np.float32
    for _ in range(60):
        x, y = int(rng.integers(20, 140)), int(rng.integers(20, 100))
        deg = f32(rng.uniform(0, 360))
        rad = f32(deg * f32(np.pi / 180.0))
        sn, cs = oracle.sincosf(np.array([rad], np.float32))
        a, b = f32(cs[0]), f32(sn[0])

        def at(px, py):
            px, py = f32(px), f32(py)
            r = int(np.rint(f32(f32(px * b) + f32(py * a))))
            c = int(np.rint(f32(f32(px * a) - f32(py * b))))
            return int(img[y + r, x + c])

        bits = [at(p[0], p[1]) < at(p[2], p[3]) for p in pat]
        want = np.packbits(np.array(bits, np.uint8), bitorder="little")
        assert np.array_equal(oracle.descriptor(img, x, y, float(deg)), want), (x, y, float(deg))


def _distribute_lists(xs, ys, rs, min_x, max_x, min_y, max_y, n_target):
    """DistributeOctTree / DivideNode (reference src/ORBextractor.cc:478-765) on Python lists:
    insert(0) for push_front, the expandable nodes sorted by (count, creation order).  Returns the
    indices of the kept candidates in final list order."""
    import math
    f32 = np.float32
    ratio = f32(max_x - min_x) / f32(max_y - min_y)
    n_ini = int(math.floor(float(ratio) + 0.5))                      # round(): half away from zero, ratio > 0
    hx = f32(max_x - min_x) / f32(n_ini)
    created = [0]

    def node(ulx, uly, brx, bry):
        created[0] += 1
        return {"ul": (ulx, uly), "br": (brx, bry), "keys": [], "nomore": False, "id": created[0]}

    nodes = [node(int(hx * f32(i)), 0, int(hx * f32(i + 1)), max_y - min_y) for i in range(n_ini)]
    ini = list(nodes)
    for i in range(len(xs)):
        ini[int(f32(xs[i]) / hx)]["keys"].append(i)
    nodes = [n for n in nodes if n["keys"]]
    for n in nodes:
        n["nomore"] = len(n["keys"]) == 1

    def divide(n):
        (ulx, uly), (brx, bry) = n["ul"], n["br"]
        half_x = int(math.ceil(float(f32(brx - ulx) / f32(2))))
        half_y = int(math.ceil(float(f32(bry - uly) / f32(2))))
        mx, my = ulx + half_x, uly + half_y
        kids = [node(ulx, uly, mx, my), node(mx, uly, brx, my), node(ulx, my, mx, bry), node(mx, my, brx, bry)]
        for i in n["keys"]:
            left, top = xs[i] < mx, ys[i] < my
            kids[(0 if left else 1) + (0 if top else 2)]["keys"].append(i)
        for k in kids:
            k["nomore"] = len(k["keys"]) == 1
        return kids

    def split_into(lst, n, expandable):
        for k in divide(n):
            if k["keys"]:
                lst.insert(0, k)
                if len(k["keys"]) > 1:
                    expandable.append(k)

    finish = False
    while not finish:
        prev = len(nodes)
        expandable = []
        i = 0
        while i < len(nodes):
            n = nodes[i]
            if n["nomore"]:
                i += 1
                continue
            before = len(nodes)
            split_into(nodes, n, expandable)
            i += len(nodes) - before          # the children went in at the front
            del nodes[i]                      # erase the parent; i now names its successor
        if len(nodes) >= n_target or len(nodes) == prev:
            finish = True
        elif len(nodes) + 3 * len(expandable) > n_target:
            while not finish:
                prev = len(nodes)
                todo = sorted(expandable, key=lambda n: (len(n["keys"]), n["id"]))
                expandable = []
                for n in reversed(todo):
                    split_into(nodes, n, expandable)
                    nodes.remove(n)
                    if len(nodes) >= n_target:
                        break
                if len(nodes) >= n_target or len(nodes) == prev:
                    finish = True
    out = []
    for n in nodes:
        best = n["keys"][0]
        for i in n["keys"][1:]:
            if rs[i] > rs[best]:
                best = i
        out.append(best)
    return np.array(out, np.int64)


@pytest.mark.parametrize("seed", range(12))
def test_quadtree_equals_list_statement(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    w = int(rng.integers(40, 700))
    h = int(rng.integers(max(30, (w + 3) // 4), 500))
    if int(np.floor(w / h + 0.5)) < 1:
        h = w
    n = int(rng.integers(1, 4000))
    # distinct pixels in raster order, as the detector emits them; a few large clusters so that the
    # size-sorted phase meets equal counts
    pix = rng.choice(w * h, size=min(n, w * h), replace=False)
    pix.sort()
    xs, ys = (pix % w).astype(np.int32), (pix // w).astype(np.int32)
    rs = rng.integers(7, 60, size=len(pix)).astype(np.int32)
    for target in (1, 5, int(rng.integers(10, 600)), 5000):
        want = _distribute_lists(xs, ys, rs, 16, 16 + w, 16, 16 + h, target)
        got = np.asarray(oracle.distribute(xs, ys, rs, 16, 16 + w, 16, 16 + h, target))
        assert np.array_equal(got, want), (seed, w, h, len(pix), target)


# ---------------------------------------------------------------- matcher: sequential-state core
def _c_round(v):
    """C round(): half away from zero."""
    import math
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


class _PyFrame:
    """Frame grid (reference src/Frame.cc:302-317, 494-502, 211-212) and GetFeaturesInArea (:437-490)
    on Python lists, in single precision where the reference computes in float."""

    def __init__(self, kps, desc, bounds):
        f32 = np.float32
        self.kps, self.desc = kps, desc
        self.min_x, self.min_y, self.max_x, self.max_y = (f32(b) for b in bounds)
        self.inv_w = f32(64) / f32(self.max_x - self.min_x)
        self.inv_h = f32(48) / f32(self.max_y - self.min_y)
        self.grid = [[[] for _ in range(48)] for _ in range(64)]
        for i in range(len(kps)):
            px = _c_round(float(f32(f32(kps["x"][i]) - self.min_x) * self.inv_w))
            py = _c_round(float(f32(f32(kps["y"][i]) - self.min_y) * self.inv_h))
            if 0 <= px < 64 and 0 <= py < 48:
                self.grid[px][py].append(i)

    def in_area(self, x, y, r, min_level, max_level):
        import math
        f32 = np.float32
        x, y, r = f32(x), f32(y), f32(r)
        out = []
        x0 = max(0, int(math.floor(float(f32(f32(x - self.min_x) - r) * self.inv_w))))
        if x0 >= 64:
            return out
        x1 = min(63, int(math.ceil(float(f32(f32(x - self.min_x) + r) * self.inv_w))))
        if x1 < 0:
            return out
        y0 = max(0, int(math.floor(float(f32(f32(y - self.min_y) - r) * self.inv_h))))
        if y0 >= 48:
            return out
        y1 = min(47, int(math.ceil(float(f32(f32(y - self.min_y) + r) * self.inv_h))))
        if y1 < 0:
            return out
        check = min_level > 0 or max_level >= 0
        for ix in range(x0, x1 + 1):
            for iy in range(y0, y1 + 1):
                for i in self.grid[ix][iy]:
                    if check:
                        if self.kps["octave"][i] < min_level:
                            continue
                        if max_level >= 0 and self.kps["octave"][i] > max_level:
                            continue
                    if abs(f32(self.kps["x"][i]) - x) < r and abs(f32(self.kps["y"][i]) - y) < r:
                        out.append(i)
        return out


def _three_maxima(sizes):
    """ComputeThreeMaxima (reference src/ORBmatcher.cc:1849-1894)."""
    f32 = np.float32
    m1 = m2 = m3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > m1:
            m3, m2, m1 = m2, m1, s
            i3, i2, i1 = i2, i1, i
        elif s > m2:
            m3, m2 = m2, s
            i3, i2 = i2, i
        elif s > m3:
            m3, i3 = s, i
    if f32(m2) < f32(0.1) * f32(m1):
        i2 = i3 = -1
    elif f32(m3) < f32(0.1) * f32(m1):
        i3 = -1
    return i1, i2, i3


def _search_for_initialization(f1, f2, prev, window, ratio, check_ori):
    """ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:501-628)."""
    f32 = np.float32
    big = 2 ** 31 - 1
    n1, n2 = len(f1.kps), len(f2.kps)
    m12, m21, dist2 = [-1] * n1, [-1] * n2, [big] * n2
    hist = [[] for _ in range(30)]
    nmatches = 0
    bits = np.unpackbits(f2.desc, axis=1) if n2 else None
    for i1 in range(n1):
        if f1.kps["octave"][i1] > 0:
            continue
        cand = f2.in_area(prev[i1][0], prev[i1][1], window, 0, 0)
        if not cand:
            continue
        d1 = np.unpackbits(f1.desc[i1])
        best = best2 = big
        best_idx = -1
        for i2 in cand:
            d = int((d1 != bits[i2]).sum())
            if dist2[i2] <= d:
                continue
            if d < best:
                best2, best, best_idx = best, d, i2
            elif d < best2:
                best2 = d
        if best <= 50 and f32(best) < f32(best2) * f32(ratio):
            if m21[best_idx] >= 0:
                m12[m21[best_idx]] = -1
                nmatches -= 1
            m12[i1], m21[best_idx], dist2[best_idx] = best_idx, i1, best
            nmatches += 1
            if check_ori:
                rot = f32(f1.kps["angle"][i1]) - f32(f2.kps["angle"][best_idx])
                if rot < 0:
                    rot = f32(rot + f32(360))
                b = _c_round(float(f32(rot * f32(f32(1) / f32(30)))))
                hist[0 if b == 30 else b].append(i1)
    if check_ori:
        keep = _three_maxima([len(h) for h in hist])
        for b in range(30):
            if b in keep:
                continue
            for i1 in hist[b]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    nmatches -= 1
    prev = np.array(prev, np.float32).copy()
    for i1 in range(n1):
        if m12[i1] >= 0:
            prev[i1] = (f2.kps["x"][m12[i1]], f2.kps["y"][m12[i1]])
    return nmatches, np.array(m12, np.int32), prev


def _random_frame(rng, n, protos, w=640.0, h=480.0):
    from oracle.pyoracle import KP_DTYPE
    k = np.zeros(n, KP_DTYPE)
    k["x"] = rng.uniform(-4, w + 4, n).astype(np.float32)       # a few points outside the grid bounds
    k["y"] = rng.uniform(-4, h + 4, n).astype(np.float32)
    k["x"][::7] = np.rint(k["x"][::7])                           # integer coordinates sit on cell edges
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["octave"] = rng.choice([0, 0, 0, 1, 2], n)
    k["size"], k["class_id"] = 31, -1
    d = protos[rng.integers(0, len(protos), n)].copy()
    flips = rng.integers(0, 256, (n, 12))
    for i in range(n):
        for b in flips[i, :rng.integers(0, 12)]:
            d[i, b >> 3] ^= 1 << (b & 7)
    return k, d


@pytest.mark.parametrize("seed", range(8))
def test_search_for_initialization_equals_list_statement(oracle, seed):
    """Few descriptor prototypes with a few flipped bits: many equal distances, contested
    candidates (the steal-back branch) and ratio-test failures."""
    rng = np.random.default_rng(300 + seed)
    protos = rng.integers(0, 256, (int(rng.integers(3, 40)), 32), dtype=np.uint8)
    n1, n2 = int(rng.integers(0, 500)), int(rng.integers(1, 500))
    k1, d1 = _random_frame(rng, n1, protos)
    k2, d2 = _random_frame(rng, n2, protos)
    bounds = (0.0, 0.0, 640.0, 480.0)
    sf = 1.2 ** np.arange(8)
    f1, f2 = oracle.FrameData(k1, d1, bounds, sf), oracle.FrameData(k2, d2, bounds, sf)
    p1, p2 = _PyFrame(k1, d1, bounds), _PyFrame(k2, d2, bounds)
    # grid and window queries first
    start, items = oracle.grid_build(f2)
    for ix in range(64):
        for iy in range(48):
            c = ix * 48 + iy
            assert list(items[start[c]:start[c + 1]]) == p2.grid[ix][iy]
    for _ in range(40):
        x, y, r = rng.uniform(-30, 670), rng.uniform(-30, 510), rng.uniform(1, 120)
        lv = [(-1, -1), (0, 0), (1, 2), (2, -1)][int(rng.integers(0, 4))]
        assert list(oracle.features_in_area(f2, start, items, x, y, r, *lv)) == p2.in_area(x, y, r, *lv)
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32) + rng.uniform(-20, 20, (n1, 2)).astype(np.float32)
    for window, ratio, ori in ((100, 0.9, True), (30, 0.75, False)):
        want = _search_for_initialization(p1, p2, prev, window, ratio, ori)
        got = oracle.search_for_initialization(f1, f2, prev, window, ratio, ori)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        prev = got[2]                                            # second call from the updated positions


def _search_by_projection_frames(cur, last, sf, valid, u, v, invz, has_obs, u_right, bf, th, direction, check_ori,
                                 claimed, claimed_obs):
    """ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) from the projections on
    (reference src/ORBmatcher.cc:1560-1677).  claimed[i2] = index of the last-frame point whose map
    point the current keypoint holds (-1: none), claimed_obs[i2] = that map point has observations."""
    f32 = np.float32
    claimed, claimed_obs = list(claimed), list(claimed_obs)
    hist = [[] for _ in range(30)]
    nmatches = 0
    cbits = np.unpackbits(cur.desc, axis=1)
    for i in range(len(last.kps)):
        if not valid[i] or invz[i] < 0:
            continue
        if u[i] < cur.min_x or u[i] > cur.max_x or v[i] < cur.min_y or v[i] > cur.max_y:
            continue
        octv = int(last.kps["octave"][i])
        radius = f32(f32(th) * f32(sf[octv]))
        if direction == 1:
            cand = cur.in_area(u[i], v[i], radius, octv, -1)
        elif direction == 2:
            cand = cur.in_area(u[i], v[i], radius, 0, octv)
        else:
            cand = cur.in_area(u[i], v[i], radius, octv - 1, octv + 1)
        if not cand:
            continue
        d1 = np.unpackbits(last.desc[i])
        best, best_idx = 256, -1
        for i2 in cand:
            if claimed[i2] >= 0 and claimed_obs[i2]:
                continue
            if u_right[i2] > 0:
                ur = f32(f32(u[i]) - f32(f32(bf) * f32(invz[i])))
                if abs(f32(ur - f32(u_right[i2]))) > radius:
                    continue
            d = int((d1 != cbits[i2]).sum())
            if d < best:
                best, best_idx = d, i2
        if best <= 100:
            claimed[best_idx], claimed_obs[best_idx] = i, int(has_obs[i])
            nmatches += 1
            if check_ori:
                rot = f32(last.kps["angle"][i]) - f32(cur.kps["angle"][best_idx])
                if rot < 0:
                    rot = f32(rot + f32(360))
                b = _c_round(float(f32(rot * f32(f32(1) / f32(30)))))
                hist[0 if b == 30 else b].append(best_idx)
    if check_ori:
        keep = _three_maxima([len(h) for h in hist])
        for b in range(30):
            if b not in keep:
                for i2 in hist[b]:
                    claimed[i2] = -1
                    nmatches -= 1
    return nmatches, np.array(claimed, np.int32)


@pytest.mark.parametrize("seed", range(8))
def test_search_by_projection_frames_equals_list_statement(oracle, seed):
    rng = np.random.default_rng(500 + seed)
    protos = rng.integers(0, 256, (int(rng.integers(3, 60)), 32), dtype=np.uint8)
    n_cur, n_last = int(rng.integers(1, 600)), int(rng.integers(1, 600))
    kc, dc = _random_frame(rng, n_cur, protos)
    kl, dl = _random_frame(rng, n_last, protos)
    bounds = (0.0, 0.0, 640.0, 480.0)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    fc, fl = oracle.FrameData(kc, dc, bounds, sf), oracle.FrameData(kl, dl, bounds, sf)
    pc, pl = _PyFrame(kc, dc, bounds), _PyFrame(kl, dl, bounds)
    valid = (rng.uniform(size=n_last) < 0.8).astype(np.uint8)
    # projections near current keypoints (so that windows hold candidates), some outside the image,
    # some behind the camera
    tgt = rng.integers(0, n_cur, n_last)
    u = (kc["x"][tgt] + rng.uniform(-6, 6, n_last)).astype(np.float32)
    v = (kc["y"][tgt] + rng.uniform(-6, 6, n_last)).astype(np.float32)
    u[::17] = -50
    invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
    invz[::13] = -0.1
    has_obs = (rng.uniform(size=n_last) < 0.5).astype(np.uint8)
    u_right = np.where(rng.uniform(size=n_cur) < 0.5, kc["x"] - rng.uniform(0, 30, n_cur), -1).astype(np.float32)
    claimed = np.where(rng.uniform(size=n_cur) < 0.15, rng.integers(0, n_last, n_cur), -1).astype(np.int32)
    claimed_obs = ((claimed >= 0) & (rng.uniform(size=n_cur) < 0.5)).astype(np.uint8)
    for th, direction, ori in ((7.0, 0, True), (15.0, 1, True), (15.0, 2, False), (3.0, 0, False)):
        want = _search_by_projection_frames(pc, pl, sf, valid, u, v, invz, has_obs, u_right, 40.0, th, direction, ori,
                                            claimed, claimed_obs)
        got = oracle.search_by_projection_frames(fc, fl, valid, u, v, invz, has_obs, u_right, 40.0, th, direction, ori,
                                                 claimed=claimed, claimed_obs=claimed_obs)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (seed, th, direction, ori, got[0], want[0])
