"""Host images in, host keypoints out (`orbx_extract_batch(..., ORBX_MEM_HOST)`): the shape of the
reference's own call, where the caller owns a cv::Mat and two output containers
(src/Frame.cc:320-326, Examples/RGB-D/rgbd_tum.cc:77-119).

A batch goes through the library in passes that are staged through three page-locked slots while the
uploads, kernels and downloads of neighbouring passes overlap.  Checked here, through the C ABI:
  * many passes per call (more passes than slots, ragged last pass): byte-identical to the
    device-resident call and to the CPU oracle;
  * buffers that are already page-locked (no staging copies), strided host views, calls of different
    sizes in a row on one handle (nothing of an earlier call's slots may leak into a later one);
  * `orbx_register_host` / `orbx_unregister_host`: whatever the machine answers, the handle keeps working.
Bar: bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu

W, H, NF, B = 322, 243, 400, 37
MEM_HOST, MEM_DEVICE = 0, 1


@pytest.fixture(scope="module")
def setup():
    import torch
    from slam_amd import _lib
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    lib = _lib.load()
    imgs = synth.synth_batch(B, W, H, 0x5EED0B00)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=8)   # 8 frames per pass: 5 passes, 3 slots
    cap = ex.max_keypoints()
    # the device-resident result of the same frames
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(imgs).to(dev)
    dk = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    dd = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    dn = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img, dk, dd, dn, cap)
    ex.synchronize()
    ref = (dk.cpu().numpy().view(KP_DTYPE).reshape(B, cap), dd.cpu().numpy().reshape(B, cap, 32), dn.cpu().numpy())
    yield dict(lib=lib, check=_lib.check, ex=ex, cap=cap, imgs=imgs, ref=ref, KP=KP_DTYPE)
    ex.close()


def host_call(s, src, b, row_stride, img_stride, kps, desc, n):
    s["check"](s["lib"].orbx_extract_batch(s["ex"]._h, src.ctypes.data, b, W, H, row_stride, img_stride, kps.ctypes.data,
                                           desc.ctypes.data, s["cap"], n.ctypes.data, MEM_HOST))


def assert_equal_to_reference(s, kps, desc, n, frames=None, sentinel=None):
    rk, rd, rn = s["ref"]
    frames = range(len(n)) if frames is None else frames
    for i, f in enumerate(frames):
        assert n[i] == rn[f], (i, f)
        assert kps[i, :n[i]].tobytes() == rk[f, :n[i]].tobytes(), "frame %d: keypoints differ" % f
        assert np.array_equal(desc[i, :n[i]], rd[f, :n[i]]), "frame %d: descriptors differ" % f
        if sentinel is not None:   # nothing is written past a frame's own entries
            assert (desc[i, n[i]:] == sentinel).all(), "frame %d: bytes past the last descriptor were written" % f


def outputs(s, b, fill=0):
    kps = np.zeros((b, s["cap"]), s["KP"])
    desc = np.full((b, s["cap"], 32), fill, np.uint8)
    n = np.full(b, -1, np.int32)
    return kps, desc, n


def test_many_passes_equal_device_path_and_oracle(setup, oracle):
    s = setup
    kps, desc, n = outputs(s, B, fill=0xA5)
    host_call(s, s["imgs"], B, W, W * H, kps, desc, n)
    assert n.min() > 50
    assert_equal_to_reference(s, kps, desc, n, sentinel=0xA5)
    p = oracle.params(nfeatures=NF)
    for f in (0, 7, 8, 23, B - 1):   # first and last frame of a pass, a middle pass, the ragged last pass
        ok, od = oracle.extract(p, s["imgs"][f], cap=s["cap"])
        assert kps[f, :n[f]].tobytes() == ok.tobytes() and np.array_equal(desc[f, :n[f]], od), f


def test_calls_of_different_sizes_in_a_row(setup):
    s = setup
    for b, first in ((B, 0), (5, 11), (20, 3), (1, B - 1), (9, 20)):
        kps, desc, n = outputs(s, b)
        host_call(s, s["imgs"][first:first + b], b, W, W * H, kps, desc, n)
        assert_equal_to_reference(s, kps, desc, n, frames=range(first, first + b))


def test_page_locked_buffers_skip_the_staging_copies(setup):
    import torch
    s = setup

    def pinned(shape, dtype):
        t = torch.zeros(int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=torch.uint8, pin_memory=True)
        return t, t.numpy().view(dtype).reshape(shape)
    ti, pi = pinned(s["imgs"].shape, np.uint8)
    pi[...] = s["imgs"]
    tk, kps = pinned((B, s["cap"]), s["KP"])
    td, desc = pinned((B, s["cap"], 32), np.uint8)
    tn, n = pinned((B,), np.int32)
    host_call(s, pi, B, W, W * H, kps, desc, n)
    assert_equal_to_reference(s, kps, desc, n)
    # page-locked images with pageable outputs, and the other way round
    k2, d2, n2 = outputs(s, B)
    host_call(s, pi, B, W, W * H, k2, d2, n2)
    assert_equal_to_reference(s, k2, d2, n2)
    kps[...] = 0
    n[...] = -1
    host_call(s, s["imgs"], B, W, W * H, kps, desc, n)
    assert_equal_to_reference(s, kps, desc, n)


def test_strided_host_view(setup):
    s = setup
    rs, ist = W + 13, (W + 13) * H + 101   # rows and frames with gaps, nothing aligned
    raw = np.full(B * ist + 64, 0x3C, np.uint8)
    view = np.lib.stride_tricks.as_strided(raw[7:], shape=(B, H, W), strides=(ist, rs, 1))
    view[...] = s["imgs"]
    kps, desc, n = outputs(s, B)
    s["check"](s["lib"].orbx_extract_batch(s["ex"]._h, view.ctypes.data, B, W, H, rs, ist, kps.ctypes.data, desc.ctypes.data,
                                           s["cap"], n.ctypes.data, MEM_HOST))
    assert_equal_to_reference(s, kps, desc, n)


def test_register_host_leaves_the_handle_working(setup):
    s = setup
    lib = s["lib"]
    raw = np.zeros(s["imgs"].nbytes + 8192, np.uint8)
    off = (-raw.ctypes.data) % 4096
    rimgs = raw[off:off + s["imgs"].nbytes].reshape(s["imgs"].shape)
    rimgs[...] = s["imgs"]
    rc = lib.orbx_register_host(C.c_void_p(rimgs.ctypes.data), C.c_size_t(rimgs.nbytes))
    if rc != 0:
        # some machines refuse to page-lock an existing allocation: a clean error, and nothing sticks
        assert rc < 0 and len(lib.orbx_last_error()) > 0
    kps, desc, n = outputs(s, B)
    host_call(s, rimgs, B, W, W * H, kps, desc, n)
    assert_equal_to_reference(s, kps, desc, n)
    if rc == 0:
        s["check"](lib.orbx_unregister_host(C.c_void_p(rimgs.ctypes.data)))
        host_call(s, rimgs, B, W, W * H, kps, desc, n)   # pageable again
        assert_equal_to_reference(s, kps, desc, n)
    # an address that was never registered is an error, not a crash
    assert lib.orbx_unregister_host(C.c_void_p(raw.ctypes.data + 1)) != 0
    host_call(s, s["imgs"][:3], 3, W, W * H, kps[:3], desc[:3], n[:3])
    assert_equal_to_reference(s, kps[:3], desc[:3], n[:3], frames=range(3))


def test_one_frame_per_call_and_empty_batch(setup):
    s = setup
    ex = s["ex"]
    rk, rd, rn = s["ref"]
    for f in (0, 17):
        k1, d1 = ex(s["imgs"][f])
        assert len(k1) == rn[f] and k1.tobytes() == rk[f, :rn[f]].tobytes() and np.array_equal(d1, rd[f, :rn[f]])
    kps, desc, n = outputs(s, 1)
    rc = s["lib"].orbx_extract_batch(ex._h, s["imgs"].ctypes.data, 0, W, H, W, W * H, kps.ctypes.data, desc.ctypes.data,
                                     s["cap"], n.ctypes.data, MEM_HOST)
    assert rc == 0 and n[0] == -1
