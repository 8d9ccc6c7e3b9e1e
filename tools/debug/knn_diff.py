"""Print where the matrix-core 2-NN differs from the CPU checker on one ragged case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle as O
from slam_amd import synth
from slam_amd.matcher import ORBmatcher

nq, nt = int(sys.argv[1]), int(sys.argv[2])
q = synth.synth_descriptors(nq, 100 + nq)
t = synth.synth_descriptors(nt, 200 + nt)
m = ORBmatcher(0.6, True)
m.debug_knn_path(2)
bi, bd, sd = m.knn2(q, t)
ri, rd, rs = O.knn2(q, t, threads=4)
bad = np.nonzero((bd != rd) | (sd != rs) | (bi != ri))[0]
print("differing queries:", len(bad))
for i in bad[:40]:
    d = np.unpackbits(q[i][None, :] ^ t, axis=1).sum(1)
    print("q %4d  got (i %4d d %3d s %3d)  want (i %4d d %3d s %3d)   d[got]=%d" % (i, bi[i], bd[i], sd[i], ri[i], rd[i], rs[i], d[bi[i]] if bi[i] >= 0 else -1))
