import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from slam_amd import synth
from slam_amd.extractor import ORBextractor
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
img = synth.synth_image(640, 480, 3)
for _ in range(50):
    ex(img, None)
ex.close()
