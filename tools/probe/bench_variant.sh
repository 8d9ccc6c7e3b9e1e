#!/bin/bash
# Runs on the GPU box: default-schedule and serialised frames/s of bench.py with other builds of the library
# (the box's copy of slam_amd/liborbx.so is replaced; nothing of that travels back).
cd $GRAFT_REPO_ROOT
cp slam_amd/liborbx.so /tmp/liborbx_orig.so
for v in "$@"; do
  if [ "$v" = "-" ]; then cp /tmp/liborbx_orig.so slam_amd/liborbx.so; else cp "$v" slam_amd/liborbx.so; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-hamming --no-extra --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v', d['value'], d['serialized']['value'], {k: round(x['us_per_frame'], 3) for k, x in d['serialized']['stages'].items()})" || exit 1
done
cp /tmp/liborbx_orig.so slam_amd/liborbx.so
