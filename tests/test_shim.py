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


def build(tmp_path):
    exe = os.path.join(str(tmp_path), "shim_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, SRC, "-L" + LIBDIR, "-lorbx",
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
    desc = np.frombuffer(blob[4 + 28 * n:], np.uint8).reshape(n, 32)
    k2, d2 = extractor_vga(img, None)
    assert n == len(k2) and np.array_equal(kps, k2) and np.array_equal(desc, d2)


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
    tail = blob[4 + 60 * n:]
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
    m.close()
    el.close()
    er.close()
