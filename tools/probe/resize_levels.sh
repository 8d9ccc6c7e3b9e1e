#!/bin/bash
# Runs on the GPU box: per-level durations of the pyramid kernels, for the wave-uniform and the per-lane form.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in ${MODES:-1 0}; do
  export ORBX_RESIZE_ROWS=$(( mode > 0 ? 1 : 0 ))
  export ORBX_RESIZE_LOGW=$mode
  OUT=gpurun_out/kt_resize_$mode
  echo "mode $mode (0 = per-lane form, 1 = wave-uniform with the chosen split, 3..6 = forced log2 of the groups per wave)"
  rm -rf $OUT && mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 3 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra > $OUT/log.txt 2>&1
  python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_resize" in r["Kernel_Name"]:
            key = (r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("VGPR_Count"))
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    v = sorted(v)
    med = v[len(v) // 2]
    tot += med
    print("grid %s  launches %d  median %.1f us  min %.1f" % (k, len(v), med, v[0]))
print("sum of medians %.1f us" % tot)
PY
  rm -rf $OUT/*/*kernel_trace.csv
done
