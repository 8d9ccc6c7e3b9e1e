#!/bin/bash
# Runs on the GPU box: default-schedule frames/s under caps on the FAST / blur kernels' share of a compute unit.
cd $GRAFT_REPO_ROOT
run() {
  v=$(env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --schedule default --no-hamming --no-extra --no-cpu 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
  echo "$* -> $v"
}
run ORBX_NPIPES=3
run ORBX_FAST_LDS_PAD=1280
run ORBX_FAST_LDS_PAD=2560
run ORBX_FAST_LDS_PAD=5120
run ORBX_BLUR_LDS=8192
run ORBX_BLUR_LDS=16384
run ORBX_BLUR_LDS=8192 ORBX_FAST_LDS_PAD=1280
run ORBX_NPIPES=4
run ORBX_NPIPES=4 ORBX_FAST_LDS_PAD=1280
run ORBX_NPIPES=2
run ORBX_FORK_BLUR=2
run ORBX_FORK_BLUR=0
