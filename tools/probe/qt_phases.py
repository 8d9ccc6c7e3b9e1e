import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from slam_amd import _lib
_lib.LIB_PATH = os.path.abspath("tools/probe/bin/liborbx_qtdiag.so")
from slam_amd import synth
from slam_amd.extractor import ORBextractor
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=int(sys.argv[1]))
imgs = synth.synth_batch(int(sys.argv[1]), 640, 480, 0x5EED0001)
ex.extract_batch(imgs)
print("---- second call", flush=True)
ex.extract_batch(imgs)
ex.close()
