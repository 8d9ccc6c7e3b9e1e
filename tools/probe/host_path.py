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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=512)
imgs = synth.synth_batch(64, 640, 480, 1)
imgs = np.ascontiguousarray(np.tile(imgs, (B // 64, 1, 1)))
ex.extract_batch(imgs[:512])
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    kps, desc, n = ex.extract_batch(imgs)
dt = time.perf_counter() - t0
print("host buffers in and out: %.0f frames/s (%d frames per call, %.1f MB in, %.1f MB out per call)" % (
    B * reps / dt, B, imgs.nbytes / 1e6, (kps.nbytes + desc.nbytes + n.nbytes) / 1e6))
ex.close()

# a width that is not a multiple of 64 (rows are repacked on the device) and a strided host view
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=256)
imgs = np.ascontiguousarray(np.tile(synth.synth_batch(32, 1241, 376, 5), (16, 1, 1)))
ex.extract_batch(imgs[:256])
t0 = time.perf_counter()
kps, desc, n = ex.extract_batch(imgs)
dt = time.perf_counter() - t0
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
