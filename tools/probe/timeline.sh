#!/bin/bash
# Runs on the GPU box: kernel timeline of the default schedule (how much of the wall time has two or more kernels running).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/timeline
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 2 --warmup 1 --frames 4096 --no-cpu --no-hamming --no-extra --schedule default > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
ev = []
for p in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "k_" not in n or "at::native" in n: continue
        import re
        m = re.search(r"\bk_[a-z0-9_]+", n)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0), r.get("Queue_Id"), r.get("Stream_Id", "")))
ev.sort()
# take the last 60% of events (timed region)
ev = ev[len(ev) * 4 // 10:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n, q, st in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[depth] += t - last
    last = t
    depth += d
wall = t1 - t0
print("wall %.2f ms, kernels %d, sum of durations %.2f ms" % (wall / 1e6, len(ev), sum(e[1] - e[0] for e in ev) / 1e6))
for k in sorted(hist): print("  %d kernels running: %.1f%% of wall" % (k, 100.0 * hist[k] / wall))
per = collections.defaultdict(list)
for s, e, n, q, st in ev: per[n].append((e - s) / 1e3)
for n, v in per.items(): print("  %-18s n=%d mean %.1f us" % (n, len(v), sum(v) / len(v)))
print("queues:", collections.Counter((e[3]) for e in ev))
PY
rm -rf $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
