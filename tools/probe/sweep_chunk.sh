set -e
P=tools/probe/pipes.py
run() { timeout -k 10 120 python3 $P "$@" 2>&1 | tail -1; }
for ch in 64 128 256 512; do for fb in 0 1; do for np in 1 3; do echo -n "fork=$fb "; ORBX_FORK_BLUR=$fb ORBX_NPIPES=$np run - 4096 $ch; done; done; done
