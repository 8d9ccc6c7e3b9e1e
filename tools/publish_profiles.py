#!/usr/bin/env python3
"""Copies the condensed rocprofv3 results of tools/collect_profiles.sh (gpurun_out/prof_<tag>/)
into profiles/: <tag>_kernel_stats.csv, <tag>_summary.json, <tag>_rocprofv3_kernel_stats_full.csv
and pmc_traffic.json (HBM bytes per launch by stage, the file bench.py reads)."""
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "kernel_stats_%s.csv" % tag), os.path.join("profiles", "%s_kernel_stats.csv" % tag))
shutil.copy(os.path.join(src, "summary_%s.json" % tag), os.path.join("profiles", "%s_summary.json" % tag))
ser = os.path.join(src, "kernel_stats_serialized_%s.csv" % tag)
if os.path.exists(ser):
    shutil.copy(ser, os.path.join("profiles", "%s_kernel_stats_serialized.csv" % tag))
fullser = os.path.join(src, "serial", "s_kernel_stats.csv")
if os.path.exists(fullser):
    shutil.copy(fullser, os.path.join("profiles", "%s_rocprofv3_kernel_stats_serialized_full.csv" % tag))
full = os.path.join(src, "stats", "s_kernel_stats.csv")
if os.path.exists(full):
    shutil.copy(full, os.path.join("profiles", "%s_rocprofv3_kernel_stats_full.csv" % tag))
summ = json.load(open(os.path.join(src, "summary_%s.json" % tag)))
stage_of = {"k_fast": "fast", "k_quadtree": "quadtree", "k_blur_stream": "blur", "k_blur_rows": "blur", "k_describe": "describe",
            "k_resize_stream": "pyramid", "k_pyramid_frame": "pyramid"}
out = {}
for k, t in summ["traffic"].items():
    if k not in stage_of:
        continue
    if k == "k_resize_stream" and "k_pyramid_frame" in summ["traffic"]:
        continue                                 # the benchmark batch goes through the one-launch pyramid
    lpp = t.get("launches_per_pass", 1)         # the counters are already summed over the launches of a pass
    out[stage_of[k]] = {"kernel": k + (" (%g launches per pass, summed)" % lpp if lpp != 1 else ""),
                        "hbm_bytes_per_launch": t["hbm_bytes_per_launch"],
                        "launches_per_pass": lpp,
                        "frames_per_launch": t["frames_per_launch"],
                        "read_bytes_corrected": t["read_bytes_corrected"],
                        "write_bytes": t["write_bytes"]}
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import kernel_source_hash  # noqa: E402
out["source_hash"] = kernel_source_hash()
out["tag"] = tag
json.dump(out, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
sq = os.path.join("gpurun_out", "sq_" + tag, "sq_summary.csv")
if os.path.exists(sq):
    shutil.copy(sq, os.path.join("profiles", "%s_sq_counters.csv" % tag))
    json.dump({"tag": tag, "source_hash": out["source_hash"], "file": "%s_sq_counters.csv" % tag},
              open(os.path.join("profiles", "sq_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
