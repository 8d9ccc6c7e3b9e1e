"""Known-answer tests for the oracle's matcher-side restatements that have no reference
fixture: vocabulary-guided searches, the Fuse / Sim3 candidate loop, stereo matching,
distinctive descriptors, mask filter and map-file records.  Every expected value below is
worked out by hand from the reference's rules."""
import struct

import numpy as np

from oracle import pyoracle as O

BOUNDS = (0.0, 0.0, 640.0, 480.0)
SF = np.array([1.2 ** i for i in range(8)], np.float32)


def desc_with_bits(base, flips):
    d = base.copy()
    for b in flips:
        d[b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def kp(x, y, octave=0, angle=0.0):
    k = np.zeros(1, O.KP_DTYPE)
    k["x"], k["y"], k["octave"], k["angle"], k["size"], k["class_id"] = x, y, octave, angle, 31, -1
    return k


def frame(kps, descs):
    return O.FrameData(np.concatenate(kps), np.stack(descs), BOUNDS, SF)


def test_bow_claims_ratio_threshold_and_rotation():
    A = np.arange(32, dtype=np.uint8)
    B = (np.arange(32, dtype=np.uint8) * 7 + 3).astype(np.uint8)
    # frame 1: kp0 ~ A, kp1 ~ A (both want the same target), kp2 ~ B; all in node 9
    f1 = frame([kp(10, 10), kp(20, 20), kp(30, 30)], [A, desc_with_bits(A, [1]), B])
    # frame 2: t0 = A + 2 bits, t1 = A + 30 bits, t2 = B + 51 bits (too far), node 9; t3 = B exact but in node 4
    f2 = frame([kp(11, 11), kp(21, 21), kp(31, 31), kp(41, 41)],
               [desc_with_bits(A, [5, 9]), desc_with_bits(A, range(40, 70)), desc_with_bits(B, range(100, 151)), B])
    fv1 = {9: [0, 1, 2]}
    fv2 = {4: [3], 9: [0, 1, 2]}
    v1 = np.ones(3, np.uint8)
    # kp0: best t0 (2) vs second t1 (30): 2 < 0.6*30 -> match.  kp1: t0 is taken; best t1 at 31 bits,
    # second t2 far: 31 <= 50 and 31 < 0.6 * d(A', B-ish) -> accepted if the second is > 51.7.
    n, m12, m21 = O.search_by_bow(f1, fv1, v1, f2, fv2, None, 50, 0.6, False)
    d_second = O.hamming256(desc_with_bits(A, [1]), desc_with_bits(B, range(100, 151)))
    expect1 = 1 if 31 < 0.6 * d_second else -1
    assert m12.tolist() == [0, expect1, -1] and n == (2 if expect1 >= 0 else 1)
    # kp2 only meets t2 at 51 bits > TH_LOW: never matched; t3 (exact) sits in another node
    assert m21[3] == -1 and m21[0] == 0
    # the keyframe-keyframe overload accepts < TH_LOW only: a best distance of exactly 50 flips
    g1 = frame([kp(1, 1)], [A])
    g2 = frame([kp(1, 1)], [desc_with_bits(A, range(50))])
    assert O.search_by_bow(g1, {1: [0]}, np.ones(1, np.uint8), g2, {1: [0]}, None, 50, 0.9, False)[0] == 1
    assert O.search_by_bow(g1, {1: [0]}, np.ones(1, np.uint8), g2, {1: [0]}, np.ones(1, np.uint8), 49, 0.9, False)[0] == 0
    # invalid first-side keypoints and ineligible second-side keypoints are skipped
    assert O.search_by_bow(g1, {1: [0]}, np.zeros(1, np.uint8), g2, {1: [0]}, None, 50, 0.9, False)[0] == 0
    # rotation cull: 12 matches rotated by 0 degrees, 1 by 90: 1 < 0.1 * 12 -> the odd one goes
    kps1, kps2, d1, d2 = [], [], [], []
    rng = np.random.default_rng(0)
    for i in range(13):
        base = rng.integers(0, 256, 32).astype(np.uint8)
        kps1.append(kp(i, i, angle=90.0 if i == 5 else 10.0))
        kps2.append(kp(i, i, angle=10.0))
        d1.append(base)
        d2.append(base)
    h1, h2 = frame(kps1, d1), frame(kps2, d2)
    fv = {i: [i] for i in range(13)}
    n, m12, _ = O.search_by_bow(h1, fv, np.ones(13, np.uint8), h2, fv, None, 50, 0.9, True)
    assert n == 12 and m12[5] == -1 and all(m12[i] == i for i in range(13) if i != 5)
    assert O.search_by_bow(h1, fv, np.ones(13, np.uint8), h2, fv, None, 50, 0.9, False)[0] == 13


def test_triangulation_gates():
    A = np.arange(32, dtype=np.uint8)
    f1 = frame([kp(100, 100)], [A])
    # candidates in the same node: c0 on the epipolar line y = 100 at distance 5,
    # c1 exact descriptor but 3 px off the line (3^2 = 9 > 3.84), c2 on the line, same distance as c0 (later wins)
    f2 = frame([kp(150, 100), kp(160, 103), kp(170, 100)],
               [desc_with_bits(A, range(5)), A, desc_with_bits(A, range(10, 15))])
    # horizontal epipolar lines: l = (0, 1, -y1): a = x*F00 + y*F10 + F20 = 0, b = 1, c = -y
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    sigma2 = (SF ** 2).astype(np.float32)
    no = np.zeros(3, np.uint8)
    ur = np.full(3, -1, np.float32)
    n, m = O.search_for_triangulation(f1, {3: [0]}, no[:1], ur[:1], f2, {3: [0, 1, 2]}, no, ur, F12, 5000.0, 100.0, sigma2,
                                      False, False)
    assert n == 1 and m.tolist() == [2]
    # a keypoint closer than sqrt(100 * scale) to the epipole is skipped in the monocular case
    n, m = O.search_for_triangulation(f1, {3: [0]}, no[:1], ur[:1], f2, {3: [0, 1, 2]}, no, ur, F12, 172.0, 100.0, sigma2,
                                      False, False)
    assert m.tolist() == [0]
    # ... but not when either side is a stereo point
    n, m = O.search_for_triangulation(f1, {3: [0]}, no[:1], np.array([90.0], np.float32), f2, {3: [0, 1, 2]}, no, ur, F12,
                                      172.0, 100.0, sigma2, False, False)
    assert m.tolist() == [2]
    # keypoints that already hold a map point do not take part; stereo-only mode needs mvuRight >= 0
    n, m = O.search_for_triangulation(f1, {3: [0]}, no[:1], ur[:1], f2, {3: [0, 1, 2]}, np.array([0, 0, 1], np.uint8), ur,
                                      F12, 5000.0, 100.0, sigma2, False, False)
    assert m.tolist() == [0]
    n, m = O.search_for_triangulation(f1, {3: [0]}, no[:1], ur[:1], f2, {3: [0, 1, 2]}, no, ur, F12, 5000.0, 100.0, sigma2,
                                      True, False)
    assert n == 0


def test_window_best_levels_chi2_and_ties():
    A = np.arange(32, dtype=np.uint8)
    kf = frame([kp(100, 100, 2), kp(101, 100, 1), kp(102, 100, 3), kp(100.5, 100, 2)],
               [desc_with_bits(A, [0, 1]), desc_with_bits(A, [2, 3]), A, desc_with_bits(A, [4])])
    one = np.ones(1, np.uint8)
    u, v, r = np.array([100.0], np.float32), np.array([100.0], np.float32), np.array([5.0], np.float32)
    # level 2 query: octave 3 (exact descriptor) is outside [1, 2]; kp3 (1 bit) beats kp0 and kp1 (2 bits)
    bi, bd = O.window_best(kf, one, u, v, r, [2], A[None])
    assert (bi[0], bd[0]) == (3, 1)
    # ties: kp0 and kp1 both at 2 bits; kp0 comes first in the cell's bucket order
    kf2 = frame([kp(100, 100, 2), kp(101, 100, 1)], [desc_with_bits(A, [0, 1]), desc_with_bits(A, [2, 3])])
    bi, bd = O.window_best(kf2, one, u, v, r, [2], A[None])
    assert (bi[0], bd[0]) == (0, 2)
    # chi2 gate, monocular: e2 * invSigma2 > 5.99 drops a keypoint 2.5 px away at level 0 (6.25 > 5.99)
    kf3 = frame([kp(102.5, 100, 0), kp(102.4, 100, 0)], [A, desc_with_bits(A, [7])])
    isg = (1.0 / SF ** 2).astype(np.float32)
    bi, bd = O.window_best(kf3, one, u, v, r, [0], A[None], True, np.full(2, -1, np.float32), np.zeros(1, np.float32), isg)
    assert (bi[0], bd[0]) == (1, 1)     # 2.4^2 = 5.76 passes
    bi, bd = O.window_best(kf3, one, u, v, r, [0], A[None])
    assert (bi[0], bd[0]) == (0, 0)
    # nothing in the window
    bi, bd = O.window_best(kf3, one, np.array([400.0], np.float32), v, r, [0], A[None])
    assert (bi[0], bd[0]) == (-1, 256)


def test_distinctive_descriptor_median_rule():
    Z = np.zeros(32, np.uint8)
    # three descriptors at pairwise distances d01 = 4, d02 = 10, d12 = 6:
    # rows sorted: [0,4,10], [0,4,6], [0,6,10]; element (N-1)/2 = 1 -> medians 4, 4, 6 -> first minimum: index 0
    d0, d1, d2 = Z, desc_with_bits(Z, range(4)), desc_with_bits(Z, range(10))
    assert O.hamming256(d0, d1) == 4 and O.hamming256(d0, d2) == 10 and O.hamming256(d1, d2) == 6
    best, med = O.distinctive_descriptors(np.stack([d0, d1, d2]), [0, 3])
    assert (best[0], med[0]) == (0, 4)
    # two descriptors: element 0 of the sorted row is the zero self-distance
    best, med = O.distinctive_descriptors(np.stack([d2, d1]), [0, 2])
    assert (best[0], med[0]) == (0, 0)
    # four: element 1 (0.5 * 3 truncated)
    d3 = desc_with_bits(Z, range(100, 103))
    best, med = O.distinctive_descriptors(np.stack([d0, d1, d2, d3]), [0, 4, 4])
    rows = [[O.hamming256(a, b) for b in (d0, d1, d2, d3)] for a in (d0, d1, d2, d3)]
    meds = [sorted(r)[1] for r in rows]
    assert best[0] == int(np.argmin(meds)) and med[0] == min(meds) and best[1] == -1


def test_mask_filter_and_record_layout():
    ks = np.concatenate([kp(10.9, 20.9, 1, 45.0), kp(300, 200, 2, 90.0), kp(50, 60, 0, 1.5)])
    ks["response"] = [7, 8, 9]
    ds = np.stack([np.full(32, i, np.uint8) for i in (1, 2, 3)])
    mask = np.ones((480, 640), np.uint8)
    mask[200, 300] = 0
    k, d = O.filter_by_mask(ks, ds, mask)
    assert k["x"].tolist() == [np.float32(10.9), 50.0] and d[:, 0].tolist() == [1, 3]
    mask[20, 10] = 3      # truncated coordinates (10, 20): value 3 is not 1
    k, d = O.filter_by_mask(ks, ds, mask)
    assert d[:, 0].tolist() == [3]
    sparse = np.zeros((480, 640), np.uint8)
    sparse[:312] = 1      # exactly 65 %: not "more than", so nothing is filtered
    assert len(O.filter_by_mask(ks, ds, sparse)[0]) == 3
    rec = O.pack_keyframe_records(ks, ds, np.array([5, 2 ** 64 - 1, 7], np.uint64))
    assert len(rec) == 192
    x, y, size, angle, resp, octave = struct.unpack("<fffffi", rec[:24].tobytes())
    assert (x, y, size, angle, resp, octave) == (np.float32(10.9), np.float32(20.9), 31.0, 45.0, 7.0, 1)
    assert rec[24:56].tolist() == [1] * 32 and struct.unpack("<Q", rec[56:64].tobytes())[0] == 5
    assert struct.unpack("<Q", rec[64 + 56:128].tobytes())[0] == 2 ** 64 - 1
    k2, d2, mi = O.unpack_keyframe_records(rec)
    assert k2["octave"].tolist() == [1, 2, 0] and np.array_equal(d2, ds) and mi.tolist() == [5, 2 ** 64 - 1, 7]
    assert np.all(k2["class_id"] == -1)


def test_stereo_row_band_and_thresholds():
    from slam_amd import synth
    p = O.params(nfeatures=500)
    left, right = synth.synth_stereo_pair(320, 240, 4, 6, 6, 0)
    kl, dl = O.extract(p, left, cap=800)
    kr, dr = O.extract(p, right, cap=800)
    n, ur, dp = O.compute_stereo_matches(p, left, right, kl, dl, kr, dr, 0.5, 200.0)
    ok = ur >= 0
    assert n == ok.sum() and n > 50
    # planted disparity 6 px: matches recover it to sub-pixel accuracy, depth = mbf / disparity
    disp = kl["x"][ok] - ur[ok]
    assert abs(np.median(disp) - 6.0) < 0.2
    assert np.array_equal(dp[ok], (np.float32(200.0) / disp.astype(np.float32)).astype(np.float32))
    # no candidate can exceed the maximum disparity mbf / mb
    n2, ur2, _ = O.compute_stereo_matches(p, left, right, kl, dl, kr, dr, 0.5, 2.0)   # maxD = 4 < 6
    assert n2 <= n // 10
    # right keypoints moved 40 rows away: the true partners leave every row band (r = 2 * scale <= 7.2);
    # what still matches does so with some other keypoint that lies within the band of its row
    kr2 = kr.copy()
    kr2["y"] += 40
    n3, ur3, _ = O.compute_stereo_matches(p, left, right, kl, dl, kr2, dr, 0.5, 200.0)
    assert n3 <= n // 4
    for i in np.nonzero(ur3 >= 0)[0]:
        r = 2.0 * SF[kr2["octave"]]
        row = int(kl["y"][i])
        assert np.any((np.floor(kr2["y"] - r) <= row) & (row <= np.ceil(kr2["y"] + r)))
