#!/usr/bin/env python3
"""Frames/s of the VGA benchmark shape under the library's concurrent schedule:
    python3 tools/probe/pipes.py [library] [frames] [frames per pass] [width height features]
Run with ORBX_NPIPES / GPU_MAX_HW_QUEUES in the environment to sweep the passes in flight and
the hardware queues they map to."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slam_amd import _lib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 512
W, H, NF = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (640, 480, 1000)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=chunk)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ex.set_stream(st.cuda_stream)
cap = ex.max_keypoints()
fr = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
step = 128 if W * H <= 640 * 480 else 8
for f0 in range(0, B, step):
    nb = min(step, B - f0)
    fr[f0:f0 + nb] = synth.synth_batch_torch(nb, W, H, 0x5EED0001 + f0, device="cuda")
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(2):
    ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("lib=%s %dx%d/%d frames=%d chunk=%d npipes=%s hwq=%s: %.0f frames/s" % (
    os.path.basename(_lib.LIB_PATH), W, H, NF, B, chunk, os.environ.get("ORBX_NPIPES", "-"),
    os.environ.get("GPU_MAX_HW_QUEUES", "-"), steps * B / dt), flush=True)
