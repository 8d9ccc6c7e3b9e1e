#!/usr/bin/env python3
"""Serialised per-stage times of the VGA benchmark shape with another build of the library:
    python3 tools/probe/lib_variant.py tools/probe/bin/liborbx_xyz.so [frames per pass]
(for compile-time constants that have no run-time knob)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slam_amd import _lib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 512   # frames per pass
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 8     # pyramid levels
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402

B = 2048
ex = ORBextractor(1000, 1.2, NL, 20, 7, max_width=640, max_height=480, max_batch=MB)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ex.set_stream(st.cuda_stream)
cap = ex.max_keypoints()
fr = torch.empty((B, 480, 640), dtype=torch.uint8, device="cuda")
for f0 in range(0, B, 128):
    fr[f0:f0 + 128] = synth.synth_batch_torch(128, 640, 480, 0x5EED0001 + f0, device="cuda")
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
ex.set_overlap(False)
for _ in range(2):
    ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
ex.profile_enable(True)
ex.profile_reset()
for _ in range(3):
    ex.extract_batch_device(fr, k, d, n, cap)
ex.synchronize()
prof = ex.profile_read()
print(os.path.basename(_lib.LIB_PATH), "frames per pass", MB, {name: round(1000.0 * ms / (3 * B), 4) for name, (ms, _) in prof.items()},
      "launches", {name: n for name, (_, n) in prof.items()})
