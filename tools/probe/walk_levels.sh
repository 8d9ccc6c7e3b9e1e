#!/bin/bash
# Runs on the GPU box: per-launch durations (grouped by grid shape) of the level-walk kernel, or of the
# separate pyramid / blur kernels with ORBX_WALK=0, under bench.py --schedule serial.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-walk}
OUT=gpurun_out/kt_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 3 --warmup 1 --frames 512 --schedule serial --no-cpu --no-hamming --no-extra > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os, re
out = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        m = re.search(r"\bk_[A-Za-z0-9_]+", n)
        if not m: continue
        key = (m.group(0), r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("VGPR_Count"))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
tot = collections.defaultdict(float)
for k, v in sorted(acc.items(), key=lambda kv: (kv[0][0], -sum(kv[1]) / len(kv[1]))):
    v = sorted(v)
    med = v[len(v) // 2]
    tot[k[0]] += med
    print("%-18s grid %6s x %5s vgpr %4s launches %3d  median %8.1f us  min %8.1f" % (k[0], k[1], k[2], k[3], len(v), med, v[0]))
for k, t in tot.items(): print("sum of medians %-18s %8.1f us" % (k, t))
PY
rm -rf $OUT/*/*kernel_trace.csv
