set -e
P=tools/probe/pipes.py
run() { timeout -k 10 160 python3 $P "$@" 2>&1 | tail -1; }
for ch in 1024 2048 4096; do for np in 1 3; do ORBX_NPIPES=$np run - 8192 $ch; done; done
ORBX_NPIPES=1 run - 8192 512
