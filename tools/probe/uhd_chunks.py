#!/usr/bin/env python3
"""3840x2160 / 10 000 features: frames/s against the number of frames per internal pass (device-resident)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

w, h, nf = 3840, 2160, 10000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
buf = torch.zeros((B, h, w), dtype=torch.uint8, device=dev)
for f0 in range(0, B, 16):
    n = min(16, B - f0)
    buf[f0:f0 + n] = synth.synth_batch_torch(n, w, h, 0x5EED7001 + f0, device=dev)
for chunk in [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32", "64", "128"])]:
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=chunk)
    cap = ex.max_keypoints()
    k = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(buf, k, d, n, cap)
    ex.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ex.extract_batch_device(buf, k, d, n, cap)
    ex.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("frames per pass %4d: %.0f frames/s (mean keypoints %.0f)" % (chunk, B * reps / dt, n.float().mean().item()), flush=True)
    ex.close()
    del k, d, n
    torch.cuda.empty_cache()
