"""Print where the device blur differs from the CPU checker (level by level) for one frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle as O
from slam_amd import synth
from slam_amd.extractor import ORBextractor

img = synth.synth_image(640, 480, 7)
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
ex(img, None)
for lvl in range(8):
    ref = O.gaussian7(ex.image_pyramid_level(0, lvl))
    got = ex.debug_blurred(0, lvl)
    bad = np.argwhere(got != ref)
    print("level", lvl, got.shape, "differing px", len(bad))
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "distinct cols", len(set(xs)), "distinct rows", len(set(ys)))
        print("  first", bad[:8].tolist(), [int(got[y, x]) for y, x in bad[:8]], [int(ref[y, x]) for y, x in bad[:8]])
