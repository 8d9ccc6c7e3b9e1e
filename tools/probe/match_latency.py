#!/usr/bin/env python3
"""Per-call latency of the matcher entry points with host arrays in and out (the shape of the
reference's calls: one frame pair per call)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402
from slam_amd.matcher import FrameView, ORBmatcher  # noqa: E402

B = (0.0, 0.0, 640.0, 480.0)
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
ex2 = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
scene = synth.synth_image(700, 520, 77)
a = np.ascontiguousarray(scene[20:500, 30:670])
b = np.ascontiguousarray(scene[26:506, 21:661])
ka, da = ex(a, None)
kb, db = ex2(b, None)
sf = ex.GetScaleFactors()
fa, fb = FrameView(ka, da, B, sf), FrameView(kb, db, B, sf)
m = ORBmatcher(0.9, True)


def timeit(name, fn, reps=100):
    for _ in range(3):
        out = fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    print("%-28s median %7.0f us  p90 %7.0f us   (%s)" % (name, np.median(t), np.percentile(t, 90), out if np.isscalar(out) else out[0]))


n = len(ka)
valid = np.ones(n, np.uint8)
u = (ka["x"] + 9).astype(np.float32)
v = (ka["y"] - 6).astype(np.float32)
invz = np.full(n, 0.5, np.float32)
prev = np.stack([ka["x"] + 9, ka["y"] - 6], axis=1).astype(np.float32)
timeit("AssignFeaturesToGrid", lambda: (len(m.AssignFeaturesToGrid(fb)[1]),))
timeit("SearchForInitialization", lambda: m.SearchForInitialization(fa, fb, prev, 100))
timeit("SearchByProjection (frames)", lambda: m.SearchByProjection(fb, fa, 15.0, valid, u, v, invz, valid, 40.0, 0))
lvl = ka["octave"].astype(np.int32)
cosv = np.full(n, 0.9, np.float32)
timeit("SearchByProjection (points)", lambda: m.SearchByProjectionMapPoints(fb, valid, u, v, u - 5, lvl, cosv, da, valid, 3.0))
node = ((np.floor(ka["x"] / 24) * 7919 + np.floor(ka["y"] / 24) * 104729) % 97).astype(int)
node2 = ((np.floor((kb["x"] - 9) / 24) * 7919 + np.floor((kb["y"] + 6) / 24) * 104729) % 97).astype(int)
fva, fvb = {}, {}
for i, nd in enumerate(node):
    fva.setdefault(int(nd), []).append(i)
for i, nd in enumerate(node2):
    fvb.setdefault(int(nd), []).append(i)
timeit("SearchByBoW", lambda: m.SearchByBoW(fa, fva, valid, fb, fvb))
timeit("knn2 1000 x 1000", lambda: (int(m.knn2(da, db)[1].sum()),))
left, right = synth.synth_stereo_pair(640, 480, 5, 12, 30, 2)
kl, dl = ex(left, None)
kr, dr = ex2(right, None)
timeit("ComputeStereoMatches", lambda: (m.ComputeStereoMatches(ex, ex2, kl, dl, kr, dr, 0.54, 386.1448)[2],))
m.close()
ex.close()
ex2.close()
