set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_extract_gpu.py tests/test_golden.py tests/test_stereo_gpu.py tests/test_kitti_pipeline_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
