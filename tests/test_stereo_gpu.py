"""Parity of the HIP stereo matcher (Frame::ComputeStereoMatches, reference
src/Frame.cc:603-774) against the CPU oracle, through the C ABI.

Bar: mvuRight and mvDepth bit-identical (compared as raw 32-bit patterns) and the same
number of stereo points.  The reference holds no fixture for this function: the oracle is
pinned by its known-answer checks (a constant-disparity pair must come back with that
disparity) only, so this is "parity unpinned" beyond those.
"""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


stereo_pair = synth.synth_stereo_pair


def run_both(oracle, left, right, nfeatures, mb, mbf):
    from slam_amd.extractor import ORBextractor
    from slam_amd.matcher import ORBmatcher
    h, w = left.shape
    p = oracle.params(nfeatures=nfeatures)
    el = ORBextractor(nfeatures, 1.2, 8, 20, 7, max_width=w, max_height=h)
    er = ORBextractor(nfeatures, 1.2, 8, 20, 7, max_width=w, max_height=h)
    m = ORBmatcher(0.6, True)
    try:
        kl, dl = el(left, None)
        kr, dr = er(right, None)
        okl, odl = oracle.extract(p, left, cap=len(kl) + 64)
        okr, odr = oracle.extract(p, right, cap=len(kr) + 64)
        assert kl.tobytes() == okl.tobytes() and np.array_equal(dl, odl)
        assert kr.tobytes() == okr.tobytes() and np.array_equal(dr, odr)
        ur, dp, n = m.ComputeStereoMatches(el, er, kl, dl, kr, dr, mb, mbf)
        on, our, odp = oracle.compute_stereo_matches(p, left, right, okl, odl, okr, odr, mb, mbf)
    finally:
        m.close()
        el.close()
        er.close()
    return (kl, ur, dp, n), (our, odp, on)


def assert_same(got, want):
    kl, ur, dp, n = got
    our, odp, on = want
    assert n == on
    assert np.array_equal(ur.view(np.uint32), our.view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), odp.view(np.uint32))


@pytest.mark.parametrize("w,h,nf,dt,db,noise", [(640, 480, 1000, 18, 18, 0), (640, 480, 1000, 7, 31, 3),
                                                (1241, 376, 2000, 12, 40, 2), (752, 480, 1500, 0, 3, 4)])
def test_stereo_matches_equal_oracle(oracle, w, h, nf, dt, db, noise):
    left, right = stereo_pair(w, h, 100 + dt, dt, db, noise)
    got, want = run_both(oracle, left, right, nf, 0.54, 386.1448)
    assert_same(got, want)
    kl, ur, dp, n = got
    assert n > 100                                    # the pair does produce stereo points
    if dt == db and noise == 0:                       # known answer: constant disparity
        d = (kl["x"] - ur)[ur >= 0]
        assert abs(np.median(d) - dt) < 0.25


def test_stereo_small_max_disparity_and_zero_disparity(oracle):
    # mbf/mb = 10 px of disparity: most of the 18 px pair is out of range.  A noisy copy at zero
    # shift sits on the disparity >= 0 gate (reference src/Frame.cc:744-750); an exact copy
    # has median SAD 0, and then the cut at 2.1 * median removes every match (:768-774).
    left, right = stereo_pair(640, 480, 5, 18, 18, 0)
    got, want = run_both(oracle, left, right, 1000, 1.0, 10.0)
    assert_same(got, want)
    got, want = run_both(oracle, left, left.copy(), 1000, 0.54, 386.1448)
    assert_same(got, want)
    assert got[3] == 0
    left, right = stereo_pair(640, 480, 6, 0, 0, 3)
    got, want = run_both(oracle, left, right, 1000, 0.54, 386.1448)
    assert_same(got, want)
    kl, ur, dp, n = got
    assert n > 100 and np.all(dp[ur >= 0] > 0)


def test_stereo_arbitrary_keypoints(oracle):
    """Keypoints the extractor would never emit (image border, octaves outside the pyramid,
    rows outside the image): both sides must agree on "no match" for them."""
    from slam_amd.extractor import ORBextractor
    from slam_amd.matcher import ORBmatcher
    from slam_amd._lib import KP_DTYPE
    left, right = stereo_pair(640, 480, 9, 10, 10, 2)
    p = oracle.params()
    el = ORBextractor(1000, 1.2, 8, 20, 7)
    er = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(0.6, True)
    try:
        kl, dl = el(left, None)
        kr, dr = er(right, None)
        rng = np.random.default_rng(3)
        n = 1500
        ks = np.zeros(n, KP_DTYPE)
        ks["x"] = rng.uniform(-5, 645, n).astype(np.float32)
        ks["y"] = rng.uniform(-5, 485, n).astype(np.float32)
        ks["octave"] = rng.integers(-1, 9, n)
        ks["size"] = 31
        kr2 = ks.copy()
        kr2["x"] = (ks["x"] - rng.uniform(0, 30, n)).astype(np.float32)
        kr2["y"] = (ks["y"] + rng.uniform(-1, 1, n)).astype(np.float32)
        kr2["octave"] = np.clip(ks["octave"] + rng.integers(-1, 2, n), -1, 8)
        d1 = synth.synth_descriptors(n, 21)
        d2 = synth.perturb_descriptors(d1, 22, 20)
        ur, dp, cnt = m.ComputeStereoMatches(el, er, ks, d1, kr2, d2, 0.54, 386.1448)
        on, our, odp = oracle.compute_stereo_matches(p, left, right, ks, d1, kr2, d2, 0.54, 386.1448)
        assert cnt == on
        assert np.array_equal(ur.view(np.uint32), our.view(np.uint32))
        assert np.array_equal(dp.view(np.uint32), odp.view(np.uint32))
        assert (ur >= 0).sum() > 50
        # empty sides
        ur, dp, cnt = m.ComputeStereoMatches(el, er, kl, dl, kr[:0], dr[:0], 0.54, 386.1448)
        assert cnt == 0 and np.all(ur == -1) and np.all(dp == -1)
        ur, dp, cnt = m.ComputeStereoMatches(el, er, kl[:0], dl[:0], kr, dr, 0.54, 386.1448)
        assert cnt == 0 and len(ur) == 0
    finally:
        m.close()
        el.close()
        er.close()


def test_random_stereo_configurations_equal_oracle():
    """A seeded sweep over image sizes, pyramid shapes, disparity ranges, noise and baselines
    (tools/fuzz_stereo.py runs the same generator for as long as asked)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_stereo.py")
    spec = importlib.util.spec_from_file_location("fuzz_stereo", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(4242)
    for case in range(60):
        c = fz.draw(rng)
        err = fz.run_one(c)
        assert err is None, "%s for %r" % (err, c)
    assert fz.STATS["points"] > 5000
