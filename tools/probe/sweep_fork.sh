set -e
P=tools/probe/pipes.py
run() { timeout -k 10 120 python3 $P "$@" 2>&1 | tail -1; }
for fb in 1 2 0; do for np in 1 2 3 4; do echo "fork=$fb"; ORBX_FORK_BLUR=$fb ORBX_NPIPES=$np run - 4096 512; done; done
