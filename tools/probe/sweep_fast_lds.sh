# FAST kernel against its LDS allocation: row pitch, padding (waves per compute unit), level groups
set -e
V=tools/probe/lib_variant.py
run() { timeout -k 10 120 python3 $V "$@" 2>&1 | tail -1 | cut -c1-135; }
echo "# serialised stage times, us per frame, 640x480, 512 frames per pass"
for g in 0 1; do echo -n "ORBX_FAST_GROUPS=$g "; ORBX_FAST_GROUPS=$g run; done
for p in 52 56 60 64; do echo -n "ORBX_FAST_PITCH=$p "; ORBX_FAST_PITCH=$p run; done
for pad in 16 200 1024 2048 4096 6400; do echo -n "ORBX_FAST_GROUPS=1 ORBX_FAST_LDS_PAD=$pad "; ORBX_FAST_GROUPS=1 ORBX_FAST_LDS_PAD=$pad run; done
echo "# one level (cells of 31 x 32 pixels, 5040 bytes of LDS)"
for pad in 0 160 640 1330; do echo -n "ORBX_FAST_LDS_PAD=$pad "; ORBX_FAST_LDS_PAD=$pad run - 512 1; done
echo "# pass size"
for mb in 64 128 256 512 1024; do run - $mb; done
