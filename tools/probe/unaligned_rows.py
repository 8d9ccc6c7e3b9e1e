#!/usr/bin/env python3
"""Device-resident 1241x376 frames stored with a row stride of 1241 bytes (nothing aligned): frames/s with the
per-pass repack to aligned rows (default) and with the frames read in place (ORBX_NO_REPACK=1), beside rows the
caller padded to 1244 bytes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

W, H, NF, B, CH = 1241, 376, 2000, 1024, 512
dev = torch.device("cuda", 0)
for stride in (1241, 1244):
    buf = torch.zeros((B, H, stride), dtype=torch.uint8, device=dev)
    for f0 in range(0, B, 64):
        buf[f0:f0 + 64, :, :W] = synth.synth_batch_torch(64, W, H, 0x5EED7001 + f0, device=dev)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=CH)
    cap = ex.max_keypoints()
    k = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(2):
        ex.extract_batch_device(buf[:, :, :W], k, d, n, cap)
    ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        ex.extract_batch_device(buf[:, :, :W], k, d, n, cap)
    ex.synchronize()
    dt = time.perf_counter() - t0
    print("row stride %d, ORBX_NO_REPACK=%s: %.0f frames/s, checksum %d" % (
        stride, os.environ.get("ORBX_NO_REPACK", "0"), 4 * B / dt, int(n.sum().item()) + int(d.view(torch.int32).sum().item() & 0xFFFF)), flush=True)
    ex.close()
