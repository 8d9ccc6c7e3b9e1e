#!/usr/bin/env python3
"""Reads the constant data the reference itself holds for this path out of its source text
(as text: nothing is compiled or imported) and writes tests/golden/reference_tables.npz:

  bit_pattern_31   the 256 x 4 rBRIEF test coordinates   src/ORBextractor.cc:135-393
  patch            PATCH_SIZE, HALF_PATCH_SIZE, EDGE_THRESHOLD   src/ORBextractor.cc:51-53
  umax             the circular-patch row ends, computed from HALF_PATCH_SIZE by the reference's
                   own formula (src/ORBextractor.cc:450-465)
  matcher          TH_HIGH, TH_LOW, HISTO_LENGTH         src/ORBmatcher.cc:38-40
  grid             FRAME_GRID_ROWS, FRAME_GRID_COLS      include/Frame.h:25-26

Run in the build container (the reference tree is not present on the GPU box):
    python3 tests/golden/make_reference_tables.py [/root/reference]
"""
import math
import os
import re
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ex = open(os.path.join(ref, "src", "ORBextractor.cc"), errors="replace").read()
ma = open(os.path.join(ref, "src", "ORBmatcher.cc"), errors="replace").read()
fr = open(os.path.join(ref, "include", "Frame.h"), errors="replace").read()

m = re.search(r"static\s+int\s+bit_pattern_31_\s*\[\s*256\s*\*\s*4\s*\]\s*=\s*\{(.*?)\};", ex, re.S)
body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
body = re.sub(r"//[^\n]*", "", body)
pattern = np.array([int(t) for t in re.findall(r"-?\d+", body)], np.int32)
assert pattern.size == 1024, pattern.size


def const(text, name):
    return int(re.search(r"\b%s\s*=\s*(-?\d+)\s*;" % re.escape(name), text).group(1))


def define(text, name):
    return int(re.search(r"#define\s+%s\s+(\d+)" % re.escape(name), text).group(1))


patch = np.array([const(ex, "PATCH_SIZE"), const(ex, "HALF_PATCH_SIZE"), const(ex, "EDGE_THRESHOLD")], np.int32)
half = int(patch[1])
# src/ORBextractor.cc:450-465 (cvFloor / cvCeil / cvRound of float expressions)
s2 = np.float32(math.sqrt(2.0))
vmax = int(math.floor(np.float32(np.float32(half) * s2) / np.float32(2) + np.float32(1)))
vmin = int(math.ceil(np.float32(np.float32(half) * s2) / np.float32(2)))
umax = np.zeros(half + 1, np.int32)
hp2 = float(half * half)
for v in range(vmax + 1):
    umax[v] = int(np.rint(math.sqrt(hp2 - v * v)))   # cvRound = round half to even
v0 = 0
for v in range(half, vmin - 1, -1):
    while umax[v0] == umax[v0 + 1]:
        v0 += 1
    umax[v] = v0
    v0 += 1
matcher = np.array([const(ma, "ORBmatcher::TH_HIGH"), const(ma, "ORBmatcher::TH_LOW"), const(ma, "ORBmatcher::HISTO_LENGTH")],
                   np.int32)
grid = np.array([define(fr, "FRAME_GRID_ROWS"), define(fr, "FRAME_GRID_COLS")], np.int32)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tables.npz")
np.savez(out, bit_pattern_31=pattern, patch=patch, umax=umax, matcher=matcher, grid=grid)
print(out, pattern[:8], patch, umax, matcher, grid)
