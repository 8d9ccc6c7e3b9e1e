#!/usr/bin/env python3
"""Where a workgroup of k_pyramid_frame spends its time: per level the walk of the slowest and of the
average wave and the wait at the barrier, from s_memtime stamps (100 MHz).

Needs the diagnostic build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DORBX_PYR_DIAG \
        -o tools/probe/bin/liborbx_pyrdiag.so slam_amd/csrc/*.hip
    python3 tools/probe/pyr_phases.py [--frames 512]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slam_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "bin", "liborbx_pyrdiag.so")
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=512)
a = ap.parse_args()
lib = _lib.load()
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=a.frames)
ex.set_overlap(False)
cap = ex.max_keypoints()
fr = torch.empty((a.frames, 480, 640), dtype=torch.uint8, device="cuda")
for f0 in range(0, a.frames, 128):
    n = min(128, a.frames - f0)
    fr[f0:f0 + n] = synth.synth_batch_torch(n, 640, 480, 0x5EED0001 + f0, device="cuda")
k = torch.zeros(a.frames * cap * 28, dtype=torch.uint8, device="cuda")
d = torch.zeros(a.frames * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(a.frames, dtype=torch.int32, device="cuda")
rec = torch.zeros((a.frames, 16, 16, 4), dtype=torch.int64, device="cuda")
ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
lib.orbx_debug_pyr_diag(C.c_void_p(rec.data_ptr()))
ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
lib.orbx_debug_pyr_diag(C.c_void_p(0))
r = rec.cpu().numpy().astype(np.float64)[:, :, 1:8, :3]   # frames, waves, levels 1..7, stamps
t0 = r[:, :, 0, 0].min(axis=1)                            # a workgroup's first stamp
walk = r[..., 1] - r[..., 0]
wait = r[..., 2] - r[..., 1]
lvl = r[:, :, :, 2].max(axis=1) - r[:, :, :, 0].min(axis=1)   # level duration per workgroup
TICK = 0.01  # s_memtime ticks are shown in units of 100
out = {"frames": a.frames,
       "workgroup_us_mean": round(float((r[:, :, -1, 2].max(axis=1) - t0).mean() * TICK), 2),
       "kernel_span_us": round(float((r[..., 2].max() - r[..., 0].min()) * TICK), 2),
       "start_spread_us": round(float((t0.max() - t0.min()) * TICK), 2),
       "level_us_mean": [round(float(x) * TICK, 2) for x in lvl.mean(axis=0)],
       "walk_us_mean_wave": [round(float(x) * TICK, 2) for x in walk.mean(axis=(0, 1))],
       "walk_us_slowest_wave": [round(float(x) * TICK, 2) for x in walk.max(axis=1).mean(axis=0)],
       "walk_us_fastest_wave": [round(float(x) * TICK, 2) for x in walk.min(axis=1).mean(axis=0)],
       "walk_level1_by_wave": [round(float(x) * TICK, 1) for x in walk[:, :, 0].mean(axis=0)],
       "walk_level2_by_wave": [round(float(x) * TICK, 1) for x in walk[:, :, 1].mean(axis=0)],
       "walk_level4_by_wave": [round(float(x) * TICK, 1) for x in walk[:, :, 3].mean(axis=0)],
       "barrier_wait_us_mean_wave": [round(float(x) * TICK, 2) for x in wait.mean(axis=(0, 1))]}
print(json.dumps(out))
