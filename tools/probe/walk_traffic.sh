#!/bin/bash
# Runs on the GPU box: HBM traffic (separate FETCH_SIZE and WRITE_SIZE passes) and serialised durations of
# the one-walk-per-level kernel (ORBX_WALK=1), summed over the eight launches of a 512-frame pass.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/walk_pmc
rm -rf $OUT && mkdir -p $OUT
export ORBX_WALK=1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 1 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra --schedule serial > $OUT/f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 1 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra --schedule serial > $OUT/w.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python3 bench.py --steps 3 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra --schedule serial > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = {}
for kind, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    tot, n = 0.0, 0
    for p in glob.glob(os.path.join(out, sub, "*_counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            if r.get("Counter_Name") == kind and "k_level_walk" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    passes = max(n // 8, 1)
    res[kind + "_KiB_per_pass"] = tot / passes
    res["launches_" + kind] = n
dur = []
for p in glob.glob(os.path.join(out, "kt", "*kernel_trace.csv")):
    for r in csv.DictReader(open(p)):
        if "k_level_walk" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res["read_bytes_corrected_per_pass"] = res["FETCH_SIZE_KiB_per_pass"] * 1024 * 2
res["write_bytes_per_pass"] = res["WRITE_SIZE_KiB_per_pass"] * 1024
res["hbm_bytes_per_pass"] = res["read_bytes_corrected_per_pass"] + res["write_bytes_per_pass"]
res["ns_per_pass_serialised"] = sum(dur) / max(len(dur) // 8, 1)
res["frames_per_pass"] = 512
res["note"] = "k_level_walk (ORBX_WALK=1): eight launches per pass, one per pyramid level; FETCH_SIZE doubled (gfx950 reports half the bytes of a coalesced read stream)"
json.dump(res, open(os.path.join(out, "walk_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/fetch/*_counter_collection.csv $OUT/write/*_counter_collection.csv $OUT/kt/*kernel_trace.csv
