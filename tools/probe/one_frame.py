"""Latency of one 640x480 frame through ORBextractor::operator() (host image in, host keypoints out):
median of 200 calls; ORBX_RESIZE_ROWS / ORBX_BLUR_ROWS select the older forms of two kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
img = synth.synth_image(640, 480, 3)
for _ in range(50):
    ex(img, None)
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    ex(img, None)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("one frame per call: median %.1f us, p10 %.1f, p90 %.1f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
ex.close()
