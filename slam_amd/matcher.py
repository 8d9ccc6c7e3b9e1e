"""Host-side mirror of ORB_SLAM2::ORBmatcher (and the Frame grid) over the orbx C ABI.

Method names and argument meaning follow the reference class (reference
include/ORBmatcher.h:25-93); frames are passed as FrameView objects holding
what the reference's Frame members hold (mvKeysUn, mDescriptors, image bounds,
mvScaleFactors, mvuRight).  All searches run as HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, MEM_HOST, MFrame, check


class FrameView:
    """The matcher's view of a Frame: undistorted keypoints, descriptors, image bounds."""

    def __init__(self, kps, desc, bounds, scale_factors, u_right=None):
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        assert len(self.kps) == len(self.desc)
        self.bounds = tuple(float(b) for b in bounds)  # mnMinX, mnMinY, mnMaxX, mnMaxY
        self.scale_factors = np.ascontiguousarray(scale_factors, np.float32)
        self.u_right = (np.full(len(self.kps), -1.0, np.float32) if u_right is None
                        else np.ascontiguousarray(u_right, np.float32))
        self.c = MFrame(self.kps.ctypes.data, self.desc.ctypes.data, len(self.kps), *self.bounds)

    @property
    def N(self):
        return len(self.kps)


class ORBmatcher:
    TH_HIGH = 100  # reference src/ORBmatcher.cc:38
    TH_LOW = 50    # :39
    HISTO_LENGTH = 30  # :40

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._lib = _lib.load()
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._m = C.c_void_p()
        check(self._lib.orbm_create(device, C.byref(self._m)))

    def close(self):
        if getattr(self, "_m", None) is not None and self._m.value:
            self._lib.orbm_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return _lib.load().orbm_hamming256(a.ctypes.data, b.ctypes.data)

    # ---- Frame grid (reference src/Frame.cc:302-317, 437-502)
    def AssignFeaturesToGrid(self, frame):
        start = np.zeros(64 * 48 + 1, np.int32)
        items = np.zeros(max(frame.N, 1), np.int32)
        check(self._lib.orbm_grid_build(self._m, C.byref(frame.c), start.ctypes.data,
                                        items.ctypes.data, MEM_HOST))
        return start, items[:start[-1]].copy()

    def GetFeaturesInArea(self, frame, x, y, r, minLevel=None, maxLevel=None, qdesc=None,
                          max_cand=None):
        """Batched Frame::GetFeaturesInArea: x, y, r (and optional levels) are arrays of
        queries.  Returns (lists of candidate indices, lists of distances or None)."""
        x = np.ascontiguousarray(np.atleast_1d(x), np.float32)
        nq = len(x)
        y = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(y), nq), np.float32)
        r = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(r), nq), np.float32)
        mn = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(-1 if minLevel is None else minLevel), nq), np.int32)
        mx = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(-1 if maxLevel is None else maxLevel), nq), np.int32)
        max_cand = max_cand or max(frame.N, 1)
        idx = np.zeros((nq, max_cand), np.int32)
        dist = np.zeros((nq, max_cand), np.int32) if qdesc is not None else None
        cnt = np.zeros(nq, np.int32)
        qd = np.ascontiguousarray(qdesc, np.uint8) if qdesc is not None else None
        check(self._lib.orbm_window_candidates(
            self._m, C.byref(frame.c), x.ctypes.data, y.ctypes.data, r.ctypes.data, mn.ctypes.data,
            mx.ctypes.data, qd.ctypes.data if qd is not None else None, nq, max_cand,
            idx.ctypes.data, dist.ctypes.data if dist is not None else None, cnt.ctypes.data,
            MEM_HOST))
        assert (cnt <= max_cand).all()
        out_i = [idx[q, :cnt[q]].copy() for q in range(nq)]
        out_d = [dist[q, :cnt[q]].copy() for q in range(nq)] if dist is not None else None
        return out_i, out_d

    # ---- reference src/ORBmatcher.cc:501-628
    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        pm = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        assert len(pm) == F1.N
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._lib.orbm_search_for_initialization(
            self._m, C.byref(F1.c), C.byref(F2.c), pm.ctypes.data, m12.ctypes.data,
            int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), C.byref(n), MEM_HOST))
        return n.value, m12[:F1.N].copy(), pm

    # ---- reference src/ORBmatcher.cc:1533-1678
    def SearchByProjection(self, CurrentFrame, LastFrame, th, valid, u, v, invz, has_obs, bf,
                           direction=0, claimed=None, claimed_obs=None):
        """Frame-to-frame projection search.  The caller supplies, per last-frame keypoint,
        the projection into the current frame (u, v, invz), whether it carries a usable map
        point (valid) and whether that point has observations (has_obs); `direction` is 0
        (neither), 1 (forward) or 2 (backward).  Returns (nmatches, claimed, claimed_obs):
        claimed[i2] is the last-frame index assigned to current keypoint i2, or -1."""
        nl, nc = LastFrame.N, CurrentFrame.N
        valid = np.ascontiguousarray(valid, np.uint8)
        u = np.ascontiguousarray(u, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        invz = np.ascontiguousarray(invz, np.float32)
        has_obs = np.ascontiguousarray(has_obs, np.uint8)
        assert len(valid) == len(u) == len(v) == len(invz) == len(has_obs) == nl
        cl = np.full(max(nc, 1), -1, np.int32) if claimed is None else np.array(claimed, np.int32)
        co = np.zeros(max(nc, 1), np.uint8) if claimed_obs is None else np.array(claimed_obs, np.uint8)
        n = C.c_int(0)
        sf = CurrentFrame.scale_factors
        check(self._lib.orbm_search_by_projection(
            self._m, C.byref(CurrentFrame.c), C.byref(LastFrame.c), valid.ctypes.data,
            u.ctypes.data, v.ctypes.data, invz.ctypes.data, has_obs.ctypes.data,
            CurrentFrame.u_right.ctypes.data, sf.ctypes.data, len(sf), float(bf), float(th),
            int(direction), int(self.mbCheckOrientation), cl.ctypes.data, co.ctypes.data,
            C.byref(n), MEM_HOST))
        return n.value, cl[:nc].copy(), co[:nc].copy()

    # ---- brute force
    def knn2(self, q, t, index_base=0):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq = len(q)
        bi, bd, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
        check(self._lib.orbm_knn2(self._m, q.ctypes.data, nq, t.ctypes.data, len(t), index_base,
                                  bi.ctypes.data, bd.ctypes.data, sd.ctypes.data, MEM_HOST))
        return bi[:nq], bd[:nq], sd[:nq]

    def knn2_device(self, q, t, best_idx, best_d, second_d, index_base=0):
        """torch tensors on the device: q (nq, 32) u8, t (nt, 32) u8, int32 outputs (nq,)."""
        check(self._lib.orbm_knn2(self._m, q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0],
                                  index_base, best_idx.data_ptr(), best_d.data_ptr(),
                                  second_d.data_ptr(), _lib.MEM_DEVICE))

    def knn2_merge_device(self, ia, da, sa, ib, db, sb):
        check(self._lib.orbm_knn2_merge(self._m, ia.shape[0], ia.data_ptr(), da.data_ptr(),
                                        sa.data_ptr(), ib.data_ptr(), db.data_ptr(), sb.data_ptr(),
                                        _lib.MEM_DEVICE))

    def knn2_merge(self, a, b):
        """Merge host partials a=(idx,d,second) (earlier shard) and b; returns merged triple."""
        ia, da, sa = (np.array(x, np.int32) for x in a)
        ib, db, sb = (np.ascontiguousarray(x, np.int32) for x in b)
        check(self._lib.orbm_knn2_merge(self._m, len(ia), ia.ctypes.data, da.ctypes.data,
                                        sa.ctypes.data, ib.ctypes.data, db.ctypes.data,
                                        sb.ctypes.data, MEM_HOST))
        return ia, da, sa

    def ratio_test(self, best_d, second_d, th=None):
        """Acceptance rule of the reference's descriptor loops (src/ORBmatcher.cc:284-288)."""
        th = self.TH_LOW if th is None else th
        b = np.asarray(best_d)
        s = np.asarray(second_d)
        return (b <= th) & (b.astype(np.float32) < np.float32(self.mfNNratio) * s.astype(np.float32))

    def set_stream(self, stream_ptr):
        check(self._lib.orbm_set_stream(self._m, C.c_void_p(stream_ptr)))

    def synchronize(self):
        check(self._lib.orbm_synchronize(self._m))
