# two ranks of tests/dist_worker.py on one GPU, gloo, with a stack dump if they stall
export ORBX_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2
N=$1
mkdir -p gpurun_out/tr
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 70 python3 -X faulthandler tests/dist_worker.py gpurun_out/tr $N $N 8 > gpurun_out/tr/r$r.log 2>&1 &
done
wait
for r in 0 1; do echo "rank $r:"; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/tr/r$r.log | grep "File \"/\|Abort\|Error\|rror" | head -8 | cut -c1-150; done
ls gpurun_out/tr
