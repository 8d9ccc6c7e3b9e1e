#!/usr/bin/env python3
"""Condenses rocprofv3 output (tools/collect_profiles.sh) into small CSV/JSON summaries:
only this library's kernels, per-kernel call count / average / total duration, and the
HBM traffic per launch from the FETCH_SIZE and WRITE_SIZE passes.

FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced read stream, so the
read side is doubled; WRITE_SIZE is taken as is.  Both are in KiB."""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", "prof_" + tag)
dst = sys.argv[2] if len(sys.argv) > 2 else src


def short(name):
    m = re.search(r"\bk_[a-z0-9_]+", name)
    return m.group(0) if m else name


def mine(name):
    return "k_" in name and "at::native" not in name


def kernel_table(sub, out_name):
    """Per-kernel durations of one rocprofv3 kernel trace -> a small csv; returns {kernel: stats}."""
    res = collections.OrderedDict()
    p = os.path.join(src, sub, "s_kernel_trace.csv")
    if not os.path.exists(p):
        return res
    per = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(p)):
        if not mine(r["Kernel_Name"]):
            continue
        k = short(r["Kernel_Name"])
        per[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size_X"])
    # a pass of the extractor launches most kernels once; k_fast goes once per group of levels, so the
    # durations are also given per pass (passes = launches of the descriptor kernel)
    passes = len(per.get("k_describe", [])) or min(len(v) for v in per.values())
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,vgpr,sgpr,lds_bytes,workgroup,launches_per_pass,ns_per_pass\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%d,%d,%.1f,%d,%d,%s,%s,%s,%s,%.2f,%.1f\n" % ((k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) + meta[k] +
                                                                (len(v) / passes, sum(v) / passes)))
            res[k] = {"calls": len(v), "avg_ns": sum(v) / len(v), "launches_per_pass": len(v) / passes, "ns_per_pass": sum(v) / passes}
    return res


stats = kernel_table("stats", "kernel_stats_%s.csv" % tag)                 # the library's default (concurrent) schedule
stats_serial = kernel_table("serial", "kernel_stats_serialized_%s.csv" % tag)  # one pass at a time on one stream

traffic = {}
for kind, sub, pre in (("FETCH_SIZE", "fetch", "f"), ("WRITE_SIZE", "write", "w")):
    p = os.path.join(src, sub, pre + "_counter_collection.csv")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r.get("Counter_Name") != kind or not mine(r["Kernel_Name"]):
            continue
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    passes = len(acc.get("k_describe", [])) or 1
    for k, v in acc.items():
        # the first half of the launches are the warm-up step; the passes are identical in shape.  Counters
        # are summed per pass: k_fast is launched once per group of levels
        traffic.setdefault(k, {})[kind + "_KiB_per_launch"] = sum(v) / passes
        traffic[k]["launches_" + kind] = len(v)
        traffic[k]["launches_per_pass"] = len(v) / passes
for k, t in traffic.items():
    rd = t.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024.0 * 2.0
    wr = t.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024.0
    t["hbm_bytes_per_launch"] = rd + wr
    t["read_bytes_corrected"] = rd
    t["write_bytes"] = wr
    t["frames_per_launch"] = 512
json.dump({"tag": tag, "kernels": stats, "kernels_serialized": stats_serial, "traffic": traffic}, open(os.path.join(dst, "summary_%s.json" % tag), "w"), indent=1)
print(json.dumps(traffic, indent=1))
