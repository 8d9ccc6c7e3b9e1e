#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 kernel trace: calls, average duration, registers, LDS, launch shape.
    python3 tools/probe/parse_kernels.py <kernel_trace.csv> [substring ...]"""
import collections
import csv
import re
import sys

per, meta = collections.defaultdict(list), {}
want = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "at::native" in n or (want and not any(w in n for w in want)):
        continue
    m = re.search(r"\bk_[A-Za-z0-9_]+(<[^>]*>)?", n)
    k = (m.group(0) if m else n)[:44]
    per[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size_X"], r["Grid_Size_X"], r["Grid_Size_Y"])
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s calls %4d avg %9.1f us  vgpr %s sgpr %s lds %s wg %s grid %s x %s" % ((k, len(v), sum(v) / len(v) / 1000) + meta[k]))
