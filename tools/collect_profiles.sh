#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the benchmark and two
# separate counter passes (FETCH_SIZE, WRITE_SIZE) for HBM traffic.  Output goes to
# gpurun_out/prof_<tag>/; tools/summarize_profiles.py condenses it into profiles/.
set -e
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- \
    python3 bench.py --steps 5 --warmup 2 --no-cpu --no-hamming --no-extra --schedule default --roofline-kernel fast > $OUT/bench_stats.log 2>&1
# the same steps one pass at a time on one stream: every duration in this trace is the kernel's own
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- \
    python3 bench.py --steps 5 --warmup 2 --no-cpu --no-hamming --no-extra --schedule serial --roofline-kernel fast > $OUT/bench_serial.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- \
    python3 bench.py --steps 1 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra --schedule default > $OUT/bench_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- \
    python3 bench.py --steps 1 --warmup 1 --frames 512 --no-cpu --no-hamming --no-extra --schedule default > $OUT/bench_write.log 2>&1
python3 tools/summarize_profiles.py $TAG
rm -rf $OUT/stats/s_kernel_trace.csv $OUT/stats/s_domain_stats.csv $OUT/serial/s_kernel_trace.csv $OUT/serial/s_domain_stats.csv $OUT/fetch/f_counter_collection.csv $OUT/write/w_counter_collection.csv
head -3 $OUT/stats/s_kernel_stats.csv | cut -c1-200; cat $OUT/kernel_stats_$TAG.csv
