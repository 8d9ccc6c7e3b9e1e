"""Throughput of the extractor with 1, 2 or 3 handles (each with its own stream and scratch)
taking alternate 512-frame chunks of the same resident batch."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slam_amd import synth
from slam_amd.extractor import ORBextractor

W, H, B, CH = 640, 480, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
frames = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
for f0 in range(0, B, 128):
    frames[f0:f0 + 128] = synth.synth_batch_torch(128, W, H, 0x5EED0001 + f0, device=dev)
for nh in (1, 2, 3, 4):
    exs = [ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=CH) for _ in range(nh)]
    cap = exs[0].max_keypoints()
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    for ov in (False, True):
        for e in exs:
            e.set_overlap(ov)

        def step():
            for c, f0 in enumerate(range(0, B, CH)):
                e = exs[c % nh]
                e.extract_batch_device(frames[f0:f0 + CH], d_kps[f0 * cap * 28:(f0 + CH) * cap * 28],
                                       d_desc[f0 * cap * 32:(f0 + CH) * cap * 32], d_n[f0:f0 + CH], cap)
        step()
        for e in exs:
            e.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        for e in exs:
            e.synchronize()
        el = time.perf_counter() - t0
        print("handles %d chunk %d blur-overlap %d: %.0f frames/s" % (nh, CH, ov, 3 * B / el), flush=True)
    ref = d_n.clone()
    for e in exs:
        e.close()
    del d_kps, d_desc, d_n
