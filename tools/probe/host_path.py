#!/usr/bin/env python3
"""Rate of the host-buffer entry point (images in host memory, keypoints back to host memory):
the PCIe-inclusive figure that goes with the device-resident headline."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
imgs = synth.synth_batch(64, 640, 480, 1)
imgs = np.ascontiguousarray(np.tile(imgs, (B // 64, 1, 1)))
for chunk in (128, 256, 512):
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=chunk)
    kps, desc, n = ex.extract_batch(imgs)        # allocates the output arrays; the timed calls reuse them (touched pages)
    import ctypes as C
    from slam_amd import _lib
    lib = _lib.load()
    cap = kps.shape[1]
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        _lib.check(lib.orbx_extract_batch(ex._h, imgs.ctypes.data, B, 640, 480, 640, 640 * 480, kps.ctypes.data, desc.ctypes.data,
                                          cap, n.ctypes.data, 0))
    dt = time.perf_counter() - t0
    print("host buffers in and out, passes of %d: %.0f frames/s (%d frames per call, %.1f MB in, %.1f MB out per call)" % (
        chunk, B * reps / dt, B, imgs.nbytes / 1e6, (kps.nbytes + desc.nbytes + n.nbytes) / 1e6), flush=True)
    if chunk == 256:
        # the same call on page-locked buffers (orbx_register_host): no host-side copies
        import torch
        # page-locked arrays from torch's pinned allocator (hipHostMalloc underneath)
        def pinned(shape, dtype):
            t = torch.empty(int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=torch.uint8, pin_memory=True)
            return t, t.numpy().view(dtype).reshape(shape)
        keep = []
        tin, pimgs = pinned(imgs.shape, np.uint8)
        pimgs[...] = imgs
        tk, k2 = pinned(kps.shape, kps.dtype)
        td, d2 = pinned(desc.shape, np.uint8)
        tn, n2 = pinned(n.shape, np.int32)
        def call(src):
            _lib.check(lib.orbx_extract_batch(ex._h, src.ctypes.data, B, 640, 480, 640, 640 * 480, k2.ctypes.data, d2.ctypes.data,
                                              cap, n2.ctypes.data, 0))
        call(pimgs)
        t0 = time.perf_counter()
        for _ in range(reps):
            call(pimgs)
        dt = time.perf_counter() - t0
        same = np.array_equal(n, n2) and all(np.array_equal(kps[f, :n[f]], k2[f, :n[f]]) and np.array_equal(desc[f, :n[f]], d2[f, :n[f]]) for f in range(0, B, 37))
        print("page-locked buffers (hipHostMalloc), passes of %d: %.0f frames/s; same results: %s" % (chunk, B * reps / dt, bool(same)), flush=True)
        # orbx_register_host on an ordinary page-aligned allocation
        raw = np.zeros(imgs.nbytes + 4096, np.uint8)
        off = (-raw.ctypes.data) % 4096
        rimgs = raw[off:off + imgs.nbytes].reshape(imgs.shape)
        rimgs[...] = imgs
        import resource
        print("RLIMIT_MEMLOCK:", resource.getrlimit(resource.RLIMIT_MEMLOCK), flush=True)
        rc = lib.orbx_register_host(rimgs.ctypes.data, rimgs.nbytes)
        if rc != 0:
            print("orbx_register_host failed:", lib.orbx_last_error().decode(), flush=True)
        else:
            n2[...] = 0
            call(rimgs)
            t0 = time.perf_counter()
            for _ in range(reps):
                call(rimgs)
            dt = time.perf_counter() - t0
            print("registered images + pinned outputs: %.0f frames/s; same counts: %s" % (B * reps / dt, bool(np.array_equal(n, n2))), flush=True)
            _lib.check(lib.orbx_unregister_host(rimgs.ctypes.data))
    ex.close()

# a width that is not a multiple of 64 (rows are repacked on the device) and a strided host view
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=256)
imgs = np.ascontiguousarray(np.tile(synth.synth_batch(32, 1241, 376, 5), (32, 1, 1)))
kps, desc, n = ex.extract_batch(imgs)
t0 = time.perf_counter()
for _ in range(3):
    _lib.check(lib.orbx_extract_batch(ex._h, imgs.ctypes.data, len(imgs), 1241, 376, 1241, 1241 * 376, kps.ctypes.data,
                                      desc.ctypes.data, kps.shape[1], n.ctypes.data, 0))
dt = (time.perf_counter() - t0) / 3
print("1241x376, contiguous rows: %.0f frames/s" % (len(imgs) / dt))
big = np.zeros((len(imgs), 376, 1300), np.uint8)
big[:, :, 20:1261] = imgs
t0 = time.perf_counter()
kps2, desc2, n2 = ex.extract_batch(big[:, :, 20:1261])
dt = time.perf_counter() - t0
print("1241x376, strided view: %.0f frames/s; same results: %s" % (len(imgs) / dt, bool(np.array_equal(n, n2) and np.array_equal(desc, desc2))))
ex.close()

# single-frame latency of the drop-in call (host image in, host keypoints out)
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
img = synth.synth_image(640, 480, 3)
for _ in range(5):
    ex(img, None)
lat = []
for _ in range(200):
    t0 = time.perf_counter()
    k, d = ex(img, None)
    lat.append(time.perf_counter() - t0)
lat = np.array(lat) * 1e6
print("one 640x480 frame per call: median %.0f us, p90 %.0f us, min %.0f us (%d keypoints)" % (
    np.median(lat), np.percentile(lat, 90), lat.min(), len(k)))
ex.close()
