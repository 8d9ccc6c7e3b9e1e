"""The header-only C++ shim (include/orbx_shim.hpp): it compiles as plain C++ against
the C ABI, and on a GPU its ORB_SLAM2::ORBextractor gives the same bytes as ctypes."""
import os
import subprocess

import numpy as np
import pytest

from slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "shim_check.cpp")
LIBDIR = os.path.join(ROOT, "slam_amd")


def build(tmp_path, src=SRC, name="shim_check"):
    exe = os.path.join(str(tmp_path), name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, src, "-L" + LIBDIR, "-lorbx",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_shim_compiles_and_fails_cleanly_without_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("no-device path only")
    img = synth.synth_image(320, 240, 1)
    raw = os.path.join(str(tmp_path), "i.raw")
    img.tofile(raw)
    r = subprocess.run([exe, raw, "320", "240", os.path.join(str(tmp_path), "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "orbx_create" in r.stderr


@pytest.mark.gpu
def test_shim_matches_ctypes(tmp_path, extractor_vga):
    from slam_amd._lib import KP_DTYPE
    exe = build(tmp_path)
    img = synth.synth_image(640, 480, 17)
    raw = os.path.join(str(tmp_path), "i.raw")
    out = os.path.join(str(tmp_path), "o.bin")
    img.tofile(raw)
    subprocess.check_call([exe, raw, "640", "480", out])
    blob = open(out, "rb").read()
    n = int(np.frombuffer(blob[:4], np.int32)[0])
    kps = np.frombuffer(blob[4:4 + 28 * n], KP_DTYPE)
    desc = np.frombuffer(blob[4 + 28 * n:4 + 60 * n], np.uint8).reshape(n, 32)
    k2, d2 = extractor_vga(img, None)
    assert n == len(k2) and np.array_equal(kps, k2) and np.array_equal(desc, d2)
    # mvImagePyramid read on the host, as src/Frame.cc:610 and :700 do
    dims = np.frombuffer(blob[4 + 60 * n:20 + 60 * n], np.int32)
    assert dims[0] == 480 and dims[1] == 8
    lvl = np.frombuffer(blob[20 + 60 * n:20 + 60 * n + dims[2] * dims[3]], np.uint8).reshape(dims[2], dims[3])
    assert np.array_equal(lvl, extractor_vga.image_pyramid_level(0, 3))


@pytest.mark.gpu
def test_shim_stereo_matches_ctypes(tmp_path):
    from slam_amd.extractor import ORBextractor
    from slam_amd.matcher import ORBmatcher
    exe = build(tmp_path)
    left, right = synth.synth_stereo_pair(640, 480, 3, 14, 25, 2)
    rawl, rawr = os.path.join(str(tmp_path), "l.raw"), os.path.join(str(tmp_path), "r.raw")
    out = os.path.join(str(tmp_path), "o.bin")
    left.tofile(rawl)
    right.tofile(rawr)
    subprocess.check_call([exe, rawl, "640", "480", out, rawr])
    blob = open(out, "rb").read()
    n = int(np.frombuffer(blob[:4], np.int32)[0])
    dims = np.frombuffer(blob[4 + 60 * n:20 + 60 * n], np.int32)
    tail = blob[20 + 60 * n + dims[2] * dims[3]:]
    ns = int(np.frombuffer(tail[:4], np.int32)[0])
    ur = np.frombuffer(tail[4:4 + 4 * n], np.float32)
    dp = np.frombuffer(tail[4 + 4 * n:4 + 8 * n], np.float32)
    el, er, m = ORBextractor(1000, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7), ORBmatcher(0.6, True)
    kl, dl = el(left, None)
    kr, dr = er(right, None)
    ur2, dp2, ns2 = m.ComputeStereoMatches(el, er, kl, dl, kr, dr, 0.54, 386.1448)
    assert n == len(kl) and ns == ns2 and ns > 100
    assert np.array_equal(ur.view(np.uint32), ur2.view(np.uint32)) and np.array_equal(dp.view(np.uint32), dp2.view(np.uint32))
    # SearchByBoW through the shim against the ctypes path, same stand-in vocabulary
    from slam_amd.matcher import FrameView
    rest = tail[4 + 8 * n:]
    nb = int(np.frombuffer(rest[:4], np.int32)[0])
    m12 = np.frombuffer(rest[4:4 + 4 * n], np.int32)
    fvl, fvr = {}, {}
    for i, y in enumerate(kl["y"]):
        fvl.setdefault(int(np.float32(y) / np.float32(16)), []).append(i)
    for i, y in enumerate(kr["y"]):
        fvr.setdefault(int(np.float32(y) / np.float32(16)), []).append(i)
    sf = el.GetScaleFactors()
    mb = ORBmatcher(0.7, True)
    nb2, m12b, _ = mb.SearchByBoW(FrameView(kl, dl, (0, 0, 640, 480), sf), fvl, np.ones(n, np.uint8),
                                  FrameView(kr, dr, (0, 0, 640, 480), sf), fvr)
    mb.close()
    assert nb == nb2 and nb > 100 and np.array_equal(m12, m12b)
    # ComputeBoW through the shim's Transform against the ctypes path, same stand-in vocabulary
    bw = rest[4 + 4 * n:]
    nv = int(np.frombuffer(bw[:4], np.int32)[0])
    pairs = np.frombuffer(bw[4:4 + 12 * nv], np.dtype([("id", "<u4"), ("v", "<f8")]))
    chk = int(np.frombuffer(bw[4 + 12 * nv:12 + 12 * nv], np.uint64)[0])
    nw = 16
    voc = dict(levels=1, node_desc=np.concatenate([np.zeros((1, 32), np.uint8), dl[:nw]]),
               child_start=np.array([0] + [nw] * (nw + 1), np.int32), children=np.arange(1, nw + 1, dtype=np.int32),
               word_id=np.array([-1] + list(range(nw)), np.int32),
               weight=np.array([0.0] + [1.0 + 0.125 * i for i in range(nw)], np.float64))
    ids, vals, fv = m.BowTransform(voc, dl, 0)
    assert nv == len(ids) == nw and np.array_equal(pairs["id"], ids.astype(np.uint32))
    assert np.array_equal(pairs["v"].view(np.uint64), vals.view(np.uint64))
    assert chk == sum(k * 1000003 + len(v) * 7 + v[-1] for k, v in fv.items())
    m.close()
    el.close()
    er.close()


DROPIN = os.path.join(ROOT, "tests", "cpp", "dropin_check.cpp")


def test_reference_call_sites_compile_against_the_shim(tmp_path):
    """tests/cpp/dropin_check.cpp holds the text of the reference's call sites
    (src/Tracking.cc:944-950, :1325-1340, :1933-1943); it must build as plain C++."""
    build(tmp_path, DROPIN, "dropin_check")


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (rz @ ry @ rx).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("sensor,tz", [(2, -0.3), (0, -0.3), (1, 0.35), (2, 0.0)])
def test_reference_call_sites_match_oracle(tmp_path, oracle, sensor, tz):
    """The reference's call-site text, through the shim's reference-signature adapters and the HIP
    kernels, against the oracle started from the same poses: motion-model search, local-map search
    on top of its claims, monocular initialisation, DescriptorDistance."""
    exe = build(tmp_path, DROPIN, "dropin_check")
    p = oracle.params()
    sf = oracle.tables(p)["scale"]
    scene = synth.synth_image(700, 520, 91)
    a = np.ascontiguousarray(scene[20:500, 30:670])
    b = np.ascontiguousarray(scene[24:504, 18:658])   # view b sees the scene shifted by (+12, -4)
    ka, da = oracle.extract(p, a)
    kb, db = oracle.extract(p, b)
    rng = np.random.default_rng(100 + sensor)
    fx = fy = np.float32(500.0)
    cx, cy = np.float32(320.0), np.float32(240.0)
    bounds = (0.0, 0.0, 640.0, 480.0)
    mbf, mb = np.float32(40.0), np.float32(0.08)
    nl, nc = len(ka), len(kb)
    # last frame at the origin; world points behind its keypoints at depths 4..6
    z = rng.uniform(4.0, 6.0, nl).astype(np.float32)
    X = np.stack([(ka["x"] - cx) / fx * z, (ka["y"] - cy) / fy * z, z], axis=1).astype(np.float32)
    Tlw = np.eye(4, dtype=np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = _rot(0.002, -0.003, 0.004)
    Tcw[:3, 3] = np.array([12.0 * 5 / 500, -4.0 * 5 / 500, tz], np.float32)
    has_point = rng.uniform(size=nl) < 0.85
    outlier = rng.uniform(size=nl) < 0.1
    obs_last = np.where(rng.uniform(size=nl) < 0.7, 3, 0).astype(np.int32)
    desc_last = synth.perturb_descriptors(da, 5, 10)
    ur_cur = np.where(rng.uniform(size=nc) < 0.5, kb["x"] - mbf / np.float32(5.0), -1.0).astype(np.float32)
    ur_last = np.full(nl, -1.0, np.float32)
    # local map: points that project near keypoints of the current frame
    nm = 1200
    pick = rng.integers(0, nc, nm)
    lpx = (kb["x"][pick] + rng.normal(0, 1.5, nm)).astype(np.float32)
    lpy = (kb["y"][pick] + rng.normal(0, 1.5, nm)).astype(np.float32)
    lpxr = (lpx - mbf / np.float32(5.0) + rng.normal(0, 2.0, nm)).astype(np.float32)
    llevel = kb["octave"][pick].astype(np.int32)
    lcos = rng.uniform(0.99, 1.0, nm).astype(np.float32)
    ldesc = synth.perturb_descriptors(db[pick], 6, 25)
    linview = rng.uniform(size=nm) < 0.9
    lbad = rng.uniform(size=nm) < 0.05
    lobs = np.where(rng.uniform(size=nm) < 0.6, 2, 0).astype(np.int32)

    def frame_blob(k, d, ur, T):
        return b"".join([np.int32(len(k)).tobytes(), k.tobytes(), d.tobytes(), ur.astype(np.float32).tobytes(),
                         np.ascontiguousarray(T, np.float32).tobytes(), sf.astype(np.float32).tobytes(),
                         mbf.tobytes(), mb.tobytes()])

    def points_blob(X_, desc, obs, bad, inview, level, vcos, px, py, pxr):
        out = [np.int32(len(X_)).tobytes()]
        for i in range(len(X_)):
            out += [X_[i].astype(np.float32).tobytes(), desc[i].tobytes(), np.int32(obs[i]).tobytes(),
                    np.int32(bad[i]).tobytes(), np.int32(inview[i]).tobytes(), np.int32(level[i]).tobytes(),
                    np.float32(vcos[i]).tobytes(), np.float32(px[i]).tobytes(), np.float32(py[i]).tobytes(),
                    np.float32(pxr[i]).tobytes()]
        return b"".join(out)

    zl, zf = np.zeros(nl, np.int32), np.zeros(nl, np.float32)
    blob = b"".join([
        np.array([fx, fy, cx, cy, bounds[0], bounds[1], bounds[2], bounds[3]], np.float32).tobytes(),
        np.int32(sensor).tobytes(),
        frame_blob(ka, da, ur_last, Tlw), frame_blob(kb, db, ur_cur, Tcw), frame_blob(ka, da, ur_last, Tlw),
        points_blob(X, desc_last, obs_last, ~has_point, zl, zl, zf, zf, zf, zf),
        points_blob(np.zeros((nm, 3), np.float32), ldesc, lobs, lbad, linview, llevel, lcos, lpx, lpy, lpxr),
        outlier.astype(np.int32).tobytes()])
    scen, out = os.path.join(str(tmp_path), "s.bin"), os.path.join(str(tmp_path), "o.bin")
    open(scen, "wb").write(blob)
    subprocess.check_call([exe, scen, out])
    res = open(out, "rb").read()
    head = np.frombuffer(res[:24], np.int32)
    assert head[5] == nc
    after_motion = np.frombuffer(res[24:24 + 8 * nc], np.int64)
    after_local = np.frombuffer(res[24 + 8 * nc:24 + 16 * nc], np.int64)
    rest = res[24 + 16 * nc:]
    ni = int(np.frombuffer(rest[:4], np.int32)[0])
    ini = np.frombuffer(rest[4:4 + 4 * ni], np.int32)
    prev = np.frombuffer(rest[4 + 4 * ni:4 + 12 * ni], np.float32).reshape(ni, 2)

    # ---- the oracle, from the same poses
    mono = sensor == 0
    th = 7 if sensor == 1 else 15
    u, v, iz, direction = oracle.project_last_frame(Tcw, Tlw, X, fx, fy, cx, cy, mb, mono)
    assert direction == (0 if (mono or tz == 0.0) else (1 if tz < 0 else 2))
    fcur = oracle.FrameData(kb, db, bounds, sf)
    flast = oracle.FrameData(ka, da, bounds, sf)
    valid = (has_point & ~outlier).astype(np.uint8)
    n1, cl, co = oracle.search_by_projection_frames(fcur, flast, valid, u, v, iz, (obs_last > 0).astype(np.uint8), ur_cur,
                                                    mbf, th, direction, True, mp_desc=desc_last)
    wide = -1
    if n1 < 20:
        n1w, cl, co = oracle.search_by_projection_frames(fcur, flast, valid, u, v, iz, (obs_last > 0).astype(np.uint8),
                                                         ur_cur, mbf, 2 * th, direction, True, mp_desc=desc_last)
        wide = n1w
    assert head[0] == n1 and head[1] == wide and n1 > 300
    assert np.array_equal(after_motion, cl.astype(np.int64))
    off = 5_000_000
    n2, cl2, _ = oracle.search_by_projection_mappoints(fcur, ur_cur, (linview & ~lbad).astype(np.uint8), lpx, lpy, lpxr,
                                                       llevel, lcos, ldesc, (lobs > 0).astype(np.uint8),
                                                       3 if sensor == 2 else 1, 0.8,
                                                       claimed=np.where(cl >= 0, cl + off, -1), claimed_obs=co)
    want = np.where(cl2 >= off, cl2.astype(np.int64) - off, np.where(cl2 >= 0, cl2.astype(np.int64) + 1000000, -1))
    assert head[2] == n2 and n2 > 50
    assert np.array_equal(after_local, want)
    n3, m12, pm = oracle.search_for_initialization(flast, fcur, np.stack([ka["x"], ka["y"]], axis=1), 100, 0.9, True)
    assert head[3] == n3 and n3 > 100 and np.array_equal(ini, m12)
    assert np.array_equal(prev.view(np.uint32), pm.view(np.uint32).reshape(-1, 2))
    assert head[4] == oracle.hamming256(db[0], da[1])


# ---- the ORBX_WITH_OPENCV branch of the shim.  This image has no OpenCV: the branch is compiled against
# tests/cpp/opencv_api_subset, a declaration-compatible subset of cv::Mat / cv::KeyPoint /
# cv::InputArray / cv::OutputArray, so what is checked is that the branch is valid C++ against those
# signatures and moves the right bytes; a build against a real OpenCV remains to be done by whoever has one.
OCV_SRC = os.path.join(ROOT, "tests", "cpp", "opencv_branch_check.cpp")
OCV_INC = os.path.join(ROOT, "tests", "cpp", "opencv_api_subset")


def build_opencv_branch(tmp_path):
    exe = os.path.join(str(tmp_path), "opencv_branch_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + OCV_INC, "-o", exe, OCV_SRC, "-L" + LIBDIR,
                           "-lorbx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_opencv_branch_of_the_shim_compiles(tmp_path):
    build_opencv_branch(tmp_path)


@pytest.mark.gpu
def test_opencv_branch_of_the_shim_matches_ctypes(tmp_path, extractor_vga):
    from slam_amd._lib import KP_DTYPE
    exe = build_opencv_branch(tmp_path)
    img = synth.synth_image(640, 480, 23)
    raw = os.path.join(str(tmp_path), "i.raw")
    out = os.path.join(str(tmp_path), "o.bin")
    img.tofile(raw)
    subprocess.check_call([exe, raw, "640", "480", out])
    blob = open(out, "rb").read()
    n = int(np.frombuffer(blob[:4], np.int32)[0])
    kps = np.frombuffer(blob[4:4 + 28 * n], KP_DTYPE)
    desc = np.frombuffer(blob[4 + 28 * n:4 + 60 * n], np.uint8).reshape(n, 32)
    k2, d2 = extractor_vga(img, None)
    assert n == len(k2) and n > 900 and np.array_equal(kps, k2) and np.array_equal(desc, d2)
    rows, cols = np.frombuffer(blob[4 + 60 * n:12 + 60 * n], np.int32)
    lvl = np.frombuffer(blob[12 + 60 * n:12 + 60 * n + rows * cols], np.uint8).reshape(rows, cols)
    assert np.array_equal(lvl, extractor_vga.image_pyramid_level(0, 2))
    assert int(np.frombuffer(blob[12 + 60 * n + rows * cols:16 + 60 * n + rows * cols], np.int32)[0]) == 1
