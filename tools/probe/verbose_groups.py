import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["ORBX_VERBOSE"] = "1"
from slam_amd.extractor import ORBextractor
for (w, h, nf) in ((640, 480, 1000), (1241, 376, 2000), (3840, 2160, 10000)):
    print(w, h, flush=True)
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    import numpy as np
    L = [(ex.level_info, )]
    ex.close()
