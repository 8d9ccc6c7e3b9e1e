#!/usr/bin/env python3
"""Per-phase cycle shares of k_fast and the distribution of its queue lengths.

Needs the diagnostic build of the library (s_memtime stamps around the phases):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DORBX_FAST_DIAG \
        -o tools/probe/bin/liborbx_diag.so slam_amd/csrc/*.hip
    python3 tools/probe/fast_phases.py [--frames 512]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slam_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "bin", "liborbx_diag.so")
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=512)
a = ap.parse_args()
lib = _lib.load()
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=a.frames)
cap = ex.max_keypoints()
fr = torch.empty((a.frames, 480, 640), dtype=torch.uint8, device="cuda")
for f0 in range(0, a.frames, 128):
    n = min(128, a.frames - f0)
    fr[f0:f0 + n] = synth.synth_batch_torch(n, 640, 480, 0x5EED0001 + f0, device="cuda")
k = torch.zeros(a.frames * cap * 28, dtype=torch.uint8, device="cuda")
d = torch.zeros(a.frames * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(a.frames, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
cells = 815
rec = torch.zeros((a.frames * cells, 8), dtype=torch.int64, device="cuda")
ex.extract_batch_device(fr, k, d, n, cap)   # warm-up
ex.synchronize()
lib.orbx_debug_fast_diag(C.c_void_p(rec.data_ptr()))
ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
lib.orbx_debug_fast_diag(C.c_void_p(0))
r = rec.cpu().numpy()
r = r[r[:, :5].sum(axis=1) > 0]   # one record per wave (a wave may take several cells)
import numpy as np  # noqa: E402
ph = r[:, :5].astype(np.float64)
n1 = (r[:, 5] & 0xFFFFFFFF)
n2 = (r[:, 5] >> 32)
surv = (r[:, 6] & 0xFFFFFFFF)
p2 = (r[:, 6] >> 32)
tot = ph.sum()
names = ["roi_copy+plane_clear", "pair_test+compaction", "strength", "nms+emit", "tail"]
out = {"frames": a.frames, "waves": int(len(r)), "waves_second_threshold": int((p2 > 0).sum()),
       "cycles_per_wave": round(tot / len(r), 1),
       "phase_share": {nm: round(ph[:, i].sum() / tot, 4) for i, nm in enumerate(names)},
       "phase_cycles_per_wave": {nm: round(ph[:, i].mean(), 1) for i, nm in enumerate(names)},
       "mean_n1": round(float(n1.mean()), 2), "mean_n2": round(float(n2.mean()), 2),
       "mean_survivors": round(float(surv.mean()), 2),
       "strength_trips_mean": round(float(np.ceil(n1 / 64).mean()), 3),
       "nms_trips_mean": round(float(np.ceil(n2 / 64).mean()), 3),
       "n1_percentiles_10_50_90_99": [int(x) for x in np.percentile(n1, [10, 50, 90, 99])],
       "n2_percentiles_10_50_90_99": [int(x) for x in np.percentile(n2, [10, 50, 90, 99])],
       "n1_hist_step16": np.bincount(np.minimum(n1 >> 4, 63), minlength=64).tolist(),
       "n2_hist_step4": np.bincount(np.minimum(n2 >> 2, 63), minlength=64).tolist()}
print(json.dumps(out))
