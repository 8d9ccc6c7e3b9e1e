"""Parity of the HIP extractor against the CPU oracle, through the C ABI.

Bar: bit-exact keypoints (all seven fields), descriptors and their order.
"""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def assert_same_features(kps, desc, rk, rd, what=""):
    assert len(kps) == len(rk), "%s keypoint count %d vs oracle %d" % (what, len(kps), len(rk))
    for name in FIELDS:
        bad = np.nonzero(kps[name] != rk[name])[0]
        assert len(bad) == 0, "%s field %s differs at %s: %s vs %s" % (
            what, name, bad[:5], kps[name][bad[:5]], rk[name][bad[:5]])
    bad = np.nonzero((desc != rd).any(axis=1))[0]
    assert len(bad) == 0, "%s %d descriptors differ (first %s)" % (what, len(bad), bad[:5])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_stages_vga(oracle, extractor_vga, seed):
    ex = extractor_vga
    p = oracle.params()
    img = synth.synth_image(640, 480, seed)
    kps, desc = ex(img, None)
    for lvl in range(8):
        ref = oracle.pyramid_level(p, img, lvl)
        got = ex.image_pyramid_level(0, lvl)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), "pyramid level %d differs in %d px" % (lvl, (got != ref).sum())
        assert np.array_equal(ex.debug_blurred(0, lvl), oracle.gaussian7(ref)), "blur level %d" % lvl
        rx, ry, rr = oracle.detect_cells(ref)
        gx, gy, gr = ex.debug_candidates(0, lvl)
        assert len(gx) == len(rx), "level %d: %d candidates vs %d" % (lvl, len(gx), len(rx))
        assert np.array_equal(gx, rx) and np.array_equal(gy, ry) and np.array_equal(gr, rr)
        h, w = ref.shape
        sel = oracle.distribute(rx, ry, rr, 16, w - 16, 16, h - 16, int(ex.features_per_level()[lvl]))
        sx, sy, sr = ex.debug_selected(0, lvl)
        assert len(sx) == len(sel), "level %d: %d selected vs %d" % (lvl, len(sx), len(sel))
        assert np.array_equal(sx, rx[sel] + 16) and np.array_equal(sy, ry[sel] + 16)
        assert np.array_equal(sr, rr[sel])
    rk, rd = oracle.extract(p, img)
    assert_same_features(kps, desc, rk, rd, "seed %d" % seed)


def test_getters(oracle, extractor_vga):
    ex = extractor_vga
    t = oracle.tables(oracle.params())
    assert ex.GetLevels() == 8
    assert ex.GetScaleFactor() == np.float32(1.2)
    assert np.array_equal(ex.GetScaleFactors(), t["scale"])
    assert np.array_equal(ex.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(ex.GetScaleSigmaSquares(), t["sigma2"])
    assert np.array_equal(ex.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(ex.features_per_level(), t["quotas"])


def test_batch_host_matches_single(oracle, extractor_vga):
    ex = extractor_vga
    imgs = synth.synth_batch(11, 640, 480, 100)  # 11 > max_batch=8: two internal passes
    kps, desc, n = ex.extract_batch(imgs)
    p = oracle.params()
    for f in range(len(imgs)):
        rk, rd = oracle.extract(p, imgs[f])
        assert_same_features(kps[f, :n[f]], desc[f, :n[f]], rk, rd, "frame %d" % f)


def test_batch_device_resident(oracle, extractor_vga):
    import torch
    ex = extractor_vga
    imgs = synth.synth_batch(5, 640, 480, 200)
    cap = ex.max_keypoints()
    d_img = torch.from_numpy(imgs).cuda()
    d_kps = torch.zeros(5 * cap * 28, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(5 * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.extract_batch_device(d_img, d_kps, d_desc, d_n, cap)
    ex.synchronize()
    from slam_amd._lib import KP_DTYPE
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KP_DTYPE).reshape(5, cap)
    desc = d_desc.cpu().numpy().reshape(5, cap, 32)
    p = oracle.params()
    for f in range(5):
        rk, rd = oracle.extract(p, imgs[f])
        assert_same_features(kps[f, :n[f]], desc[f, :n[f]], rk, rd, "frame %d" % f)


def test_strided_input(oracle, extractor_vga):
    big = np.zeros((480, 700), np.uint8)
    img = synth.synth_image(640, 480, 9)
    big[:, 30:670] = img
    view = big[:, 30:670]
    kps, desc = extractor_vga(view, None)
    rk, rd = oracle.extract(oracle.params(), img)
    assert_same_features(kps, desc, rk, rd, "strided")


def test_flat_image_uses_fallback_and_finds_nothing(extractor_vga):
    img = np.full((480, 640), 90, np.uint8)
    kps, desc = extractor_vga(img, None)
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_empty_image(extractor_vga):
    kps, desc = extractor_vga(np.zeros((0, 0), np.uint8), None)
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_low_contrast_image_takes_min_threshold(oracle, extractor_vga):
    # contrast between the two FAST thresholds: only the fallback threshold fires
    img = (synth.synth_image(640, 480, 21).astype(np.int32) // 12 + 100).astype(np.uint8)
    kps, desc = extractor_vga(img, None)
    rk, rd = oracle.extract(oracle.params(), img)
    assert_same_features(kps, desc, rk, rd, "low contrast")
    assert len(kps) > 0


def test_sparse_corners_fewer_than_quota(oracle, extractor_vga):
    img = np.full((480, 640), 40, np.uint8)
    img[100:140, 100:180] = 200
    img[300:380, 400:430] = 220
    img[200:210, 50:600] = 10
    kps, desc = extractor_vga(img, None)
    rk, rd = oracle.extract(oracle.params(), img)
    assert 0 < len(rk) < 1000
    assert_same_features(kps, desc, rk, rd, "sparse")


@pytest.mark.parametrize("w,h,nf", [(1241, 376, 2000), (320, 240, 500), (752, 480, 1200)])
def test_other_shapes(oracle, w, h, nf):
    from slam_amd.extractor import ORBextractor
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    try:
        imgs = synth.synth_batch(2, w, h, 300 + w)
        kps, desc, n = ex.extract_batch(imgs)
        p = oracle.params(nfeatures=nf)
        for f in range(2):
            rk, rd = oracle.extract(p, imgs[f], cap=nf + 128)
            assert_same_features(kps[f, :n[f]], desc[f, :n[f]], rk, rd, "%dx%d frame %d" % (w, h, f))
    finally:
        ex.close()


def test_smaller_image_than_configured(oracle, extractor_vga):
    img = synth.synth_image(400, 300, 5)
    kps, desc = extractor_vga(img, None)
    rk, rd = oracle.extract(oracle.params(), img)
    assert_same_features(kps, desc, rk, rd, "400x300")
    # and back to the configured size
    img = synth.synth_image(640, 480, 6)
    kps, desc = extractor_vga(img, None)
    rk, rd = oracle.extract(oracle.params(), img)
    assert_same_features(kps, desc, rk, rd, "640x480 again")


def test_too_large_image_is_rejected(extractor_vga):
    from slam_amd._lib import OrbxError
    with pytest.raises(OrbxError):
        extractor_vga(np.zeros((481, 640), np.uint8), None)


def test_device_sincos_equals_host_libm(oracle, extractor_vga):
    # every 257th float of [0, 2*pi] plus the angles the extractor can produce near quadrant edges
    hi = np.float32(6.2832).view(np.uint32)
    a = np.arange(0, int(hi), 257, dtype=np.uint32).view(np.float32)
    extra = (np.arange(0, 360, 0.25, dtype=np.float32) * np.float32(np.pi / 180.0)).astype(np.float32)
    a = np.concatenate([a, extra])
    sn, cs = extractor_vga.debug_sincos(a)
    rs, rc = oracle.sincosf(a)
    assert np.array_equal(sn.view(np.uint32), rs.view(np.uint32))
    assert np.array_equal(cs.view(np.uint32), rc.view(np.uint32))


def test_device_fast_atan2_equals_oracle(oracle, extractor_vga):
    rng = np.random.RandomState(0)
    y = rng.randint(-200000, 200000, 200000).astype(np.float32)
    x = rng.randint(-200000, 200000, 200000).astype(np.float32)
    y[:1000] = 0
    x[1000:2000] = 0
    y[2000:2010] = 0
    x[2000:2010] = 0
    got = extractor_vga.debug_atan2(y, x)
    ref = np.array([oracle.fast_atan2(float(a), float(b)) for a, b in zip(y[:20000], x[:20000])], np.float32)
    assert np.array_equal(got[:20000].view(np.uint32), ref.view(np.uint32))
    assert (got >= 0).all() and (got <= 360).all()


def test_4k_10k_features(oracle):
    """BASELINE config 5 shape: 3840x2160, 10 000 features (largest quadtree tables)."""
    from slam_amd.extractor import ORBextractor
    ex = ORBextractor(10000, 1.2, 8, 20, 7, max_width=3840, max_height=2160, max_batch=1)
    try:
        img = synth.synth_image(3840, 2160, 4000)
        kps, desc = ex(img, None)
        rk, rd = oracle.extract(oracle.params(nfeatures=10000), img, cap=10400)
        assert len(rk) >= 10000
        assert_same_features(kps, desc, rk, rd, "4K")
    finally:
        ex.close()


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (1241, 376, 2000), (322, 243, 400)])
def test_pyramid_forms_equal_oracle(oracle, monkeypatch, fused, w, h, nf):
    """The pyramid is built either level by level (small batches) or by one launch in which a
    workgroup runs all levels of a frame (batches that give most compute units a frame); here both
    forms are forced on a small batch and compared with the oracle level by level and end to end."""
    from slam_amd.extractor import ORBextractor
    monkeypatch.setenv("ORBX_PYR_FUSED", fused)
    imgs = synth.synth_batch(3, w, h, 4100 + w)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=3)
    try:
        kps, desc, n = ex.extract_batch(imgs)
        p = oracle.params(nfeatures=nf)
        for f in range(3):
            for lvl in range(1, 8):
                assert np.array_equal(ex.image_pyramid_level(f, lvl), oracle.pyramid_level(p, imgs[f], lvl)), (f, lvl)
            rk, rd = oracle.extract(p, imgs[f], cap=nf + 128)
            assert_same_features(kps[f, :n[f]], desc[f, :n[f]], rk, rd, "%dx%d fused=%s frame %d" % (w, h, fused, f))
    finally:
        ex.close()


def test_large_batch_properties(extractor_vga):
    """At a batch size the oracle is too slow for: size-independent properties.  Identical
    frames give identical outputs wherever they sit in the batch, and every output obeys
    the extractor's invariants."""
    import torch
    from slam_amd._lib import KP_DTYPE
    ex = extractor_vga
    base = synth.synth_batch_torch(8, 640, 480, 5000, device="cuda")
    B = 96
    frames = base.repeat(B // 8, 1, 1).contiguous()
    cap = ex.max_keypoints()
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.extract_batch_device(frames, d_kps, d_desc, d_n, cap)
    ex.synchronize()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().reshape(B, cap, 32)
    for f in range(8, B):
        assert n[f] == n[f % 8]
        assert np.array_equal(kps[f, :n[f]], kps[f % 8, :n[f]])
        assert np.array_equal(desc[f, :n[f]], desc[f % 8, :n[f]])
    for f in range(8):
        k = kps[f, :n[f]]
        assert 1000 <= n[f] <= cap
        assert (np.diff(k["octave"]) >= 0).all() and (k["class_id"] == -1).all()
        assert (k["angle"] >= 0).all() and (k["angle"] <= 360).all()


@pytest.mark.parametrize("w,h,nl", [(640, 480, 8), (171, 107, 1), (151, 107, 1), (120, 107, 1), (121, 107, 1),
                                    (82, 107, 1), (87, 107, 1), (91, 107, 1)])
def test_fast_row_pitch_forms_equal_oracle(oracle, monkeypatch, w, h, nl):
    """The FAST kernel is compiled once per row pitch of its cell buffer (the widest cell of the
    pyramid, 30 .. 59 pixels, gives pitches 40 .. 68) and once with the pitch as a run-time value.
    The shapes here have widest cells of 32, 35, 40, 44, 45, 50, 55 and 59 pixels, i.e. the pitches
    40, 44, 48, 52, 56, 60, 64 and 68; each runs in both forms."""
    from slam_amd.extractor import ORBextractor
    imgs = synth.synth_batch(2, w, h, 900 + w)
    p = oracle.params(nfeatures=600, nlevels=nl)
    want = [oracle.extract(p, imgs[f], cap=900) for f in range(2)]
    assert sum(len(k) for k, _ in want) > 0
    for generic, groups in (("0", "2"), ("1", "2"), ("0", "1"), ("0", "0")):
        # groups "2": one launch per group of levels (640x480: levels 0..3 and 4..7) whatever the batch;
        # "1": one launch with the buffers of the largest cell; "0": the library's choice (one launch
        # for a batch this small)
        monkeypatch.setenv("ORBX_FAST_GENERIC", generic)
        monkeypatch.setenv("ORBX_FAST_GROUPS", groups)
        ex = ORBextractor(600, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=2)
        try:
            kps, desc, n = ex.extract_batch(imgs)
            for f in range(2):
                assert_same_features(kps[f, :n[f]], desc[f, :n[f]], want[f][0], want[f][1],
                                     "%dx%d generic=%s groups=%s frame %d" % (w, h, generic, groups, f))
        finally:
            ex.close()


def test_passes_in_flight_equal_single_frames(oracle):
    """A device-resident batch longer than one pass is spread over several passes in flight
    (own streams, own scratch slices); the result must equal frame-by-frame extraction, with
    the overlap on and off, and repeated calls must not disturb each other."""
    import torch
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    B, chunk = 44, 8   # 6 passes: 5 full + 1 of 4 frames, over 3 pipes
    frames = synth.synth_batch_torch(B, 640, 480, 7100, device="cuda")
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=chunk)
    one = ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        cap = ex.max_keypoints()
        host = frames.cpu().numpy()
        want = [one(host[f], None) for f in (0, 7, 8, 23, 40, 43)]
        for overlap in (True, False, True):
            ex.set_overlap(overlap)
            d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
            d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
            d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ex.extract_batch_device(frames, d_kps, d_desc, d_n, cap)
            ex.extract_batch_device(frames, d_kps, d_desc, d_n, cap)   # back to back on the same buffers
            ex.synchronize()
            n = d_n.cpu().numpy()
            kps = d_kps.cpu().numpy().view(KP_DTYPE).reshape(B, cap)
            desc = d_desc.cpu().numpy().reshape(B, cap, 32)
            for (k, d), f in zip(want, (0, 7, 8, 23, 40, 43)):
                assert n[f] == len(k)
                assert kps[f, :n[f]].tobytes() == k.tobytes() and np.array_equal(desc[f, :n[f]], d)
        # the pyramid of the last pass stays addressable (frames 40..43)
        lvl = ex.image_pyramid_level(41, 3)
        assert np.array_equal(lvl, oracle.pyramid_level(oracle.params(), host[41], 3))
    finally:
        ex.close()
        one.close()


@pytest.mark.parametrize("w,h,nf,blur", [(640, 480, 1000, 0), (1241, 376, 2000, 2), (333, 250, 400, 1)])
def test_one_walk_per_level_equals_oracle(oracle, monkeypatch, w, h, nf, blur):
    """ORBX_WALK=1: every level is read once by a kernel that writes its blurred image and the next
    pyramid level (instead of the separate pyramid and blur kernels): same pyramid, same blurred images,
    same keypoints and descriptors, for the three blur variants and lane splits of both widths."""
    from slam_amd.extractor import ORBextractor
    monkeypatch.setenv("ORBX_WALK", "1")
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=3, blur_variant=blur)
    try:
        imgs = synth.synth_batch(3, w, h, 900 + w)
        kps, desc, n = ex.extract_batch(imgs)
        p = oracle.params(nfeatures=nf, blur_variant=blur)
        for f in range(3):
            rk, rd = oracle.extract(p, imgs[f], cap=nf + 128)
            assert_same_features(kps[f, :n[f]], desc[f, :n[f]], rk, rd, "walk %dx%d frame %d" % (w, h, f))
        for lvl in range(8):
            ref = oracle.pyramid_level(p, imgs[2], lvl)
            assert np.array_equal(ex.image_pyramid_level(2, lvl), ref), "level %d" % lvl
            assert np.array_equal(ex.debug_blurred(2, lvl), oracle.gaussian7(ref, blur)), "blurred level %d" % lvl
    finally:
        ex.close()


def test_null_stream_then_reset_stream_with_passes_in_flight():
    """set_stream(NULL) selects the legacy default stream (it does not mean "back to the handle's own
    stream": that is reset_stream).  A multi-pass device batch gives the same bytes on the handle's own
    stream, on the null stream, and on the own stream again after reset_stream."""
    import torch
    from slam_amd.extractor import ORBextractor
    from slam_amd import _lib
    assert b"0.2" in _lib.load().orbx_version()
    B, chunk = 20, 8
    frames = synth.synth_batch_torch(B, 640, 480, 7300, device="cuda")
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=chunk)
    try:
        cap = ex.max_keypoints()
        res = []
        for mode in ("own", "null", "reset"):
            if mode == "null":
                ex.set_stream(None)
            elif mode == "reset":
                ex.reset_stream()
            k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
            d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
            n = torch.zeros(B, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ex.extract_batch_device(frames, k, d, n, cap)
            ex.synchronize()
            torch.cuda.synchronize()
            res.append((k.cpu().numpy(), d.cpu().numpy(), n.cpu().numpy()))
        assert res[0][2].min() > 900
        for r in res[1:]:
            for a, b in zip(res[0], r):
                assert np.array_equal(a, b)
    finally:
        ex.close()


@pytest.mark.parametrize("no_repack", ["0", "1"])
def test_width_not_multiple_of_four_device_resident(oracle, monkeypatch, no_repack):
    """KITTI-sized frames (1241 px wide) in device memory with rows padded to 1244 bytes take the
    aligned word paths.  With a tight pitch of 1241, or a base address that is not a multiple of 4,
    the frames are first copied to aligned rows on the device (no_repack 0), or read in place by the
    byte forms of the kernels (no_repack 1).  Same result; several passes in flight, each with its own
    slice of the aligned copies."""
    import torch
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    monkeypatch.setenv("ORBX_NO_REPACK", no_repack)
    w, h, B = 1241, 376, 5
    host = synth.synth_batch(B, w, h, 900)
    ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    try:
        cap = ex.max_keypoints()
        want = [oracle.extract(oracle.params(nfeatures=2000), host[f], cap=cap) for f in range(B)]
        for pitch, shift in ((1244, 0), (1241, 0), (1280, 0), (1244, 1), (1241, 3)):
            flat = torch.zeros(B * h * pitch + 8, dtype=torch.uint8, device="cuda")
            buf = flat[shift:shift + B * h * pitch].view(B, h, pitch)
            buf[:, :, :w] = torch.from_numpy(host).cuda()
            d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
            d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
            d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ex.extract_batch_device(buf[:, :, :w], d_kps, d_desc, d_n, cap)
            ex.synchronize()
            n = d_n.cpu().numpy()
            kps = d_kps.cpu().numpy().view(KP_DTYPE).reshape(B, cap)
            desc = d_desc.cpu().numpy().reshape(B, cap, 32)
            for f in range(B):
                k, d = want[f]
                assert n[f] == len(k), pitch
                assert kps[f, :n[f]].tobytes() == k.tobytes() and np.array_equal(desc[f, :n[f]], d), pitch
    finally:
        ex.close()


def test_random_configurations_equal_oracle():
    """A seeded sweep over image sizes, level counts, scale factors, thresholds and image
    statistics (tools/fuzz_extract.py runs the same generator for as long as asked)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_extract.py")
    spec = importlib.util.spec_from_file_location("fuzz_extract", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from slam_amd._lib import OrbxError
    rng = np.random.default_rng(2024)
    done = 0
    while done < 120:
        c = fz.draw(rng)
        try:
            err, _ = fz.run_one(c)
        except OrbxError as e:
            assert "LDS budget" in str(e), (c, e)
            continue
        assert err is None, "%s for %r" % (err, c)
        done += 1


def test_level_taller_than_twice_its_width_is_rejected(oracle):
    # round(width / height) of the detection area is the number of initial quadtree nodes
    # (reference src/ORBextractor.cc:553); at 0 the reference divides by zero.  Both sides
    # report an error instead.
    from slam_amd._lib import OrbxError
    from slam_amd.extractor import ORBextractor
    img = synth.synth_image(332, 668, 3)
    with pytest.raises(RuntimeError):
        oracle.extract(oracle.params(nfeatures=500, nlevels=1), img)
    with pytest.raises(OrbxError):
        ex = ORBextractor(500, 1.2, 1, 20, 7, max_width=332, max_height=668, max_batch=1)
        try:
            ex(img, None)
        finally:
            ex.close()


def test_two_extractors_from_two_threads(oracle):
    """The reference extracts the left and right image on two threads with two ORBextractor
    objects (src/Frame.cc:122-125): two handles used concurrently, each from its own thread."""
    import threading
    from slam_amd.extractor import ORBextractor
    imgs = [synth.synth_batch(6, 640, 480, 700), synth.synth_batch(6, 640, 480, 900)]
    p = oracle.params()
    want = [[oracle.extract(p, im) for im in side] for side in imgs]
    exs = [ORBextractor(1000, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7)]
    errors = []

    def work(side):
        try:
            for rep in range(5):
                for f in range(len(imgs[side])):
                    k, d = exs[side](imgs[side][f], None)
                    rk, rd = want[side][f]
                    if k.tobytes() != rk.tobytes() or not np.array_equal(d, rd):
                        errors.append("side %d frame %d rep %d differs" % (side, f, rep))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append("side %d: %r" % (side, e))

    threads = [threading.Thread(target=work, args=(s,)) for s in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for ex in exs:
        ex.close()
    assert not errors, errors


def test_device_capacity_error_surfaces_at_synchronize(extractor_vga):
    """A device-space batch only enqueues work, so a capacity below a frame's keypoint count
    cannot be reported by the call itself: the next synchronize reports it, once."""
    import torch
    from slam_amd._lib import OrbxError
    ex = extractor_vga
    frames = synth.synth_batch_torch(3, 640, 480, 8100, device="cuda")
    cap = 100   # far below the ~1000 keypoints of a textured frame
    d_kps = torch.zeros(3 * cap * 28, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(3 * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(3, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.extract_batch_device(frames, d_kps, d_desc, d_n, cap)
    with pytest.raises(OrbxError) as e:
        ex.synchronize()
    assert "-4" in str(e.value) and "capacity" in str(e.value)
    assert (d_n.cpu().numpy() == cap).all()      # truncated, never written past the capacity
    ex.synchronize()                             # the error state was cleared by the call that reported it
    # an unrelated host-space call afterwards is not affected
    kps, desc = ex(synth.synth_image(640, 480, 1), None)
    assert len(kps) >= 1000


def test_torch_work_and_library_share_a_stream_without_manual_waits(oracle):
    """The library runs on the stream it is given (the legacy default stream included), so work
    that torch enqueues on that stream before and after a device-space call is ordered with it:
    no synchronize between producer, library and consumer."""
    import torch
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    host = synth.synth_batch(4, 640, 480, 8200)
    p = oracle.params()
    want = [oracle.extract(p, host[f]) for f in range(4)]
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2)
    try:
        cap = ex.max_keypoints()
        for stream in (torch.cuda.Stream(), torch.cuda.default_stream()):
            ex.set_stream(stream.cuda_stream)
            with torch.cuda.stream(stream):
                pinned = torch.from_numpy(host).pin_memory()
                # producer: an upload and a chain of device ops that end in the frames
                noise = torch.randint(0, 255, (64, 4, 480, 640), dtype=torch.uint8, device="cuda")
                frames = pinned.cuda(non_blocking=True)
                for k in range(64):
                    frames = frames ^ noise[k]
                for k in range(64):
                    frames = frames ^ noise[k]
                d_kps = torch.empty(4 * cap * 28, dtype=torch.uint8, device="cuda")
                d_desc = torch.empty(4 * cap * 32, dtype=torch.uint8, device="cuda")
                d_n = torch.empty(4, dtype=torch.int32, device="cuda")
                ex.extract_batch_device(frames, d_kps, d_desc, d_n, cap)
                # consumers on the same stream
                n_copy = d_n.clone()
                desc_sum = d_desc.view(4, cap, 32).to(torch.int64).sum()
                kps_host = d_kps.cpu()
                desc_host = d_desc.cpu()
                n = n_copy.cpu().numpy()
            kps = kps_host.numpy().view(KP_DTYPE).reshape(4, cap)
            desc = desc_host.numpy().reshape(4, cap, 32)
            for f in range(4):
                rk, rd = want[f]
                assert_same_features(kps[f, :n[f]], desc[f, :n[f]], rk, rd, "frame %d" % f)
            assert int(desc_sum.item()) == int(desc.astype(np.int64).sum())
        ex.reset_stream()
    finally:
        ex.close()


def test_debug_taps_number_frames_like_get_level(oracle):
    """With a batch longer than max_batch only the last pass stays resident; every tap takes the
    frame index within the whole call and refuses frames of earlier passes."""
    from slam_amd._lib import OrbxError
    from slam_amd.extractor import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2)
    try:
        imgs = synth.synth_batch(5, 640, 480, 8300)   # passes: frames 0-1, 2-3, 4
        ex.extract_batch(imgs)
        p = oracle.params()
        ref = oracle.pyramid_level(p, imgs[4], 2)
        assert np.array_equal(ex.image_pyramid_level(4, 2), ref)
        assert np.array_equal(ex.debug_blurred(4, 2), oracle.gaussian7(ref))
        rx, ry, rr = oracle.detect_cells(ref)
        gx, gy, gr = ex.debug_candidates(4, 2)
        assert np.array_equal(gx, rx) and np.array_equal(gy, ry) and np.array_equal(gr, rr)
        sx, sy, sr = ex.debug_selected(4, 2)
        h, w = ref.shape
        sel = oracle.distribute(rx, ry, rr, 16, w - 16, 16, h - 16, int(ex.features_per_level()[2]))
        assert np.array_equal(sx, rx[sel] + 16) and np.array_equal(sy, ry[sel] + 16)
        for tap in (lambda: ex.image_pyramid_level(0, 2), lambda: ex.debug_blurred(3, 2),
                    lambda: ex.debug_candidates(1, 2), lambda: ex.debug_selected(5, 2)):
            with pytest.raises(OrbxError):
                tap()
    finally:
        ex.close()


@pytest.mark.parametrize("blur,fma", [(2, 0), (0, 1), (2, 1), (1, 1)])
def test_semantics_variants_equal_oracle(oracle, blur, fma):
    """The results of the reference depend on the OpenCV version (Gaussian blur) and on whether its
    build contracts the descriptor's steering sums into fused multiply-adds; every variant is
    reproduced bit for bit, on widths with and without a tail of width % 4 columns, from host
    memory and from a tightly pitched device image (the byte-addressed kernels)."""
    import torch
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    p = oracle.params(blur_variant=blur, fma_variant=fma)
    for w, h in ((640, 480), (642, 480), (517, 389)):
        ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2, blur_variant=blur, fma_variant=fma)
        try:
            imgs = synth.synth_batch(2, w, h, 9000 + w)
            kps, desc, n = ex.extract_batch(imgs)
            want = [oracle.extract(p, imgs[f]) for f in range(2)]
            for f in range(2):
                assert_same_features(kps[f, :n[f]], desc[f, :n[f]], want[f][0], want[f][1], "%dx%d frame %d" % (w, h, f))
            for lvl in (0, 3):
                assert np.array_equal(ex.debug_blurred(1, lvl), oracle.gaussian7(oracle.pyramid_level(p, imgs[1], lvl), blur))
            cap = ex.max_keypoints()
            d_img = torch.from_numpy(imgs).cuda()          # pitch = w: unaligned rows when w % 4 != 0
            d_kps = torch.zeros(2 * cap * 28, dtype=torch.uint8, device="cuda")
            d_desc = torch.zeros(2 * cap * 32, dtype=torch.uint8, device="cuda")
            d_n = torch.zeros(2, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ex.extract_batch_device(d_img, d_kps, d_desc, d_n, cap)
            ex.synchronize()
            nn = d_n.cpu().numpy()
            k2 = d_kps.cpu().numpy().view(KP_DTYPE).reshape(2, cap)
            dd2 = d_desc.cpu().numpy().reshape(2, cap, 32)
            for f in range(2):
                assert_same_features(k2[f, :nn[f]], dd2[f, :nn[f]], want[f][0], want[f][1], "device %dx%d frame %d" % (w, h, f))
        finally:
            ex.close()


def test_variant_mismatch_rates(oracle):
    """How far the variants are from variant 0 (reported in the design notes): the single-precision
    blur changes a few pixels per million and a few descriptors per ten thousand; the fused
    steering sums change a descriptor bit only when a rotated coordinate lands within half an ulp of
    a rounding boundary."""
    from slam_amd.extractor import ORBextractor
    imgs = synth.synth_batch(6, 640, 480, 9500)
    base = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=6)
    v2 = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=6, blur_variant=2)
    vf = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=6, fma_variant=1)
    try:
        k0, d0, n0 = base.extract_batch(imgs)
        k2, d2, n2 = v2.extract_batch(imgs)
        kf, df, nf = vf.extract_batch(imgs)
        assert np.array_equal(n0, n2) and np.array_equal(n0, nf)
        for f in range(6):   # detection and orientation do not depend on either
            assert k0[f, :n0[f]].tobytes() == k2[f, :n0[f]].tobytes() == kf[f, :n0[f]].tobytes()
        tot = int(n0.sum())
        blur_diff = sum(int((d0[f, :n0[f]] != d2[f, :n0[f]]).any(axis=1).sum()) for f in range(6))
        fma_diff = sum(int((d0[f, :n0[f]] != df[f, :n0[f]]).any(axis=1).sum()) for f in range(6))
        assert blur_diff < tot // 500 and fma_diff < tot // 500
    finally:
        base.close()
        v2.close()
        vf.close()


@pytest.mark.parametrize("w,h,nf,kind", [(1280, 960, 20000, "scene"), (960, 720, 30000, "noise"), (1920, 1080, 16000, "scene")])
def test_feature_counts_beyond_the_lds_node_tables(oracle, w, h, nf, kind):
    """The reference has no feature limit (src/ORBextractor.cc:540-765).  Quotas whose quadtree
    node tables do not fit the LDS run with the tables in global memory; same results."""
    from slam_amd.extractor import ORBextractor
    if kind == "noise":
        img = (synth.hash_u64(77, w * h, 3) & np.uint64(255)).astype(np.uint8).reshape(h, w)
    else:
        img = synth.synth_image(w, h, 4100 + nf)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    try:
        kps, desc = ex(img, None)
        rk, rd = oracle.extract(oracle.params(nfeatures=nf), img, cap=nf + 4096)
        assert len(rk) > nf // 2
        assert_same_features(kps, desc, rk, rd, "%dx%d / %d features" % (w, h, nf))
    finally:
        ex.close()
