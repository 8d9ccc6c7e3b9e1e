#!/bin/bash
# Runs on the GPU box (via gpurun): one SQ counter pass over a 512-frame benchmark step,
# averaged per kernel into gpurun_out/sq_<tag>.csv.
set -e
TAG=${1:-r1}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/a -o a -- python3 bench.py --steps 1 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra > $OUT/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM \
    --output-format csv -d $OUT/b -o b -- python3 bench.py --steps 1 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import collections, csv, glob, os, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(out, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
        if not m or "at::native" in r["Kernel_Name"]:
            continue
        acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
with open(os.path.join(out, "sq_summary.csv"), "w") as f:
    f.write("kernel,launches," + ",".join(names) + "\n")  # launches: in the run; counters: per pass
    # counters are summed per pass of the extractor (k_fast is launched once per group of levels)
    passes = max(len(v) for v in acc.get("k_describe", {"x": [0]}).values()) or 1
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        f.write(k + "," + str(n) + "," + ",".join("%.0f" % (sum(acc[k][c]) / passes) for c in names) + "\n")
print(open(os.path.join(out, "sq_summary.csv")).read())
PY
rm -rf $OUT/a/*_counter_collection.csv $OUT/b/*_counter_collection.csv
