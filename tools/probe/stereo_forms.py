#!/usr/bin/env python3
"""Kernel time per pair of the stereo matcher's two forms (a table of pairs / one pair per call) on the same
pairs: run under `rocprofv3 --kernel-trace --stats` and compare k_stereo_match_jobs / njobs with k_stereo_match_one."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd._lib import KP_DTYPE  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402
from slam_amd.matcher import ORBmatcher  # noqa: E402

W, H, NF, NP = 1241, 376, 2000, 64
MBF = 386.1448
MB = MBF / 718.856
seq = synth.synth_stereo_sequence(W, H, 77, 6, 13, [(i % 5, -(i % 3)) for i in range(NP)], noise=2)
lefts = np.stack([s[0] for s in seq])
rights = np.stack([s[1] for s in seq])
el = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=NP)
er = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=NP)
m = ORBmatcher(0.9, True)
dev = torch.device("cuda", 0)
cap = el.max_keypoints()
bufs = {}
for side, ex, imgs in (("l", el, lefts), ("r", er, rights)):
    d_img = torch.from_numpy(imgs).to(dev)
    k = torch.zeros(NP * cap * 28, dtype=torch.uint8, device=dev)
    d = torch.zeros(NP * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(NP, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img, k, d, n, cap)
    ex.synchronize()
    bufs[side] = (d_img, k.view(NP, cap, 28), d.view(NP, cap, 32), n.cpu().numpy())
d_ur = torch.zeros((NP, cap), dtype=torch.float32, device=dev)
d_dp = torch.zeros((NP, cap), dtype=torch.float32, device=dev)
jobs = []
for f in range(NP):
    nl, nr = int(bufs["l"][3][f]), int(bufs["r"][3][f])
    jobs.append(dict(kps_left=bufs["l"][1][f, :nl], desc_left=bufs["l"][2][f, :nl], kps_right=bufs["r"][1][f, :nr],
                     desc_right=bufs["r"][2][f, :nr], pyr_left=m.pyramid_of(el, f), pyr_right=m.pyramid_of(er, f),
                     u_right=d_ur[f, :nl], depth=d_dp[f, :nl]))
for _ in range(3):
    counts = m.ComputeStereoMatchesBatch(jobs, el, MB, MBF)
torch.cuda.synchronize()
for _ in range(3):
    for f in range(NP):
        nl, nr = int(bufs["l"][3][f]), int(bufs["r"][3][f])
        kl = bufs["l"][1][f, :nl].cpu().numpy().view(KP_DTYPE).reshape(-1)
        kr = bufs["r"][1][f, :nr].cpu().numpy().view(KP_DTYPE).reshape(-1)
        m.ComputeStereoMatches(el, er, kl, bufs["l"][2][f, :nl].cpu().numpy(), kr, bufs["r"][2][f, :nr].cpu().numpy(), MB, MBF,
                               frame_left=f, frame_right=f)
print("pairs", NP, "matches per pair", float(np.mean(counts)))
