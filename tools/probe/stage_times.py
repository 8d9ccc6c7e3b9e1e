#!/usr/bin/env python3
"""Serialised per-stage times of the extractor for an arbitrary frame shape.

    python3 tools/probe/stage_times.py --w 3840 --h 2160 --nf 10000 --frames 64 --chunk 16
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=640)
ap.add_argument("--h", type=int, default=480)
ap.add_argument("--nf", type=int, default=1000)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--chunk", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--serial-only", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ex = ORBextractor(a.nf, 1.2, 8, 20, 7, max_width=a.w, max_height=a.h, max_batch=min(a.chunk, a.frames))
ex.set_stream(torch.cuda.current_stream().cuda_stream)
cap = ex.max_keypoints()
pitch = (a.w + 3) // 4 * 4
buf = torch.zeros((a.frames, a.h, pitch), dtype=torch.uint8, device=dev)
for f0 in range(0, a.frames, 16):
    n = min(16, a.frames - f0)
    buf[f0:f0 + n, :, :a.w] = synth.synth_batch_torch(n, a.w, a.h, 0x5EED7001 + f0, device=dev)
k = torch.zeros(a.frames * cap * 28, dtype=torch.uint8, device=dev)
d = torch.zeros(a.frames * cap * 32, dtype=torch.uint8, device=dev)
n = torch.zeros(a.frames, dtype=torch.int32, device=dev)
for overlap in ((False,) if a.serial_only else (False, True)):
    ex.set_overlap(overlap)
    ex.extract_batch_device(buf[:, :, :a.w], k, d, n, cap)
    torch.cuda.synchronize()
    ex.profile_enable(True)
    ex.profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        ex.extract_batch_device(buf[:, :, :a.w], k, d, n, cap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex.profile_enable(False)
    prof = ex.profile_read()
    tot = a.frames * a.reps
    print("%s: %.1f frames/s, %.2f us/frame; stages us/frame: %s" % (
        "default schedule" if overlap else "serialised", tot / dt, dt / tot * 1e6,
        ", ".join("%s %.2f" % (s, 1000.0 * ms / tot) for s, (ms, _) in prof.items())), flush=True)
print("mean keypoints %.1f" % float(n.float().mean().item()))
ex.close()
