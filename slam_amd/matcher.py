"""Host-side mirror of ORB_SLAM2::ORBmatcher (and the Frame grid) over the orbx C ABI.

Method names and argument meaning follow the reference class (reference
include/ORBmatcher.h:25-93); frames are passed as FrameView objects holding
what the reference's Frame members hold (mvKeysUn, mDescriptors, image bounds,
mvScaleFactors, mvuRight).  All searches run as HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, MEM_HOST, MFrame, ProjPolicy, ProjQueries, check



def _nmatches(arr):
    """The nmatches field of every record of a ctypes job array, read in one strided view instead of
    one attribute access per record (the tables of the batched searches hold a thousand records)."""
    if len(arr) == 0:
        return []
    rec = type(arr[0])
    raw = np.frombuffer(arr, dtype=np.uint8).reshape(len(arr), C.sizeof(rec))
    off = rec.nmatches.offset
    return raw[:, off:off + 4].copy().view(np.int32).reshape(-1).tolist()

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

    def SearchForInitializationBatch(self, jobs, windowSize=10):
        """SearchForInitialization for many independent frame pairs in one set of launches (device-resident).
        `jobs`: list of dicts of torch DEVICE tensors: kps1 (n1, 28) u8, desc1 (n1, 32) u8, kps2, desc2,
        bounds (4 floats), prev_matched (n1, 2) f32 (in/out), matches12 i32 (n1, out).  Returns the list
        of match counts."""
        arr = (_lib.InitJob * len(jobs))()
        for i, p in enumerate(jobs):
            arr[i].f1 = MFrame(p["kps1"].data_ptr(), p["desc1"].data_ptr(), p["kps1"].shape[0], *p["bounds"])
            arr[i].f2 = MFrame(p["kps2"].data_ptr(), p["desc2"].data_ptr(), p["kps2"].shape[0], *p["bounds"])
            arr[i].prev_matched_xy = p["prev_matched"].data_ptr()
            arr[i].matches12 = p["matches12"].data_ptr()
        check(self._lib.orbm_search_for_initialization_batch(self._m, arr, len(arr), int(windowSize), float(self.mfNNratio),
                                                             int(self.mbCheckOrientation)))
        return _nmatches(arr)

    # ---- projected searches
    def _claims(self, n, claimed, claimed_obs):
        cl = np.full(max(n, 1), -1, np.int32) if claimed is None else np.array(claimed, np.int32)
        co = np.zeros(max(n, 1), np.uint8) if claimed_obs is None else np.array(claimed_obs, np.uint8)
        return cl, co

    def search_projected(self, train, valid, u, v, radius, min_level, max_level, desc, policy,
                         angle=None, has_obs=None, u_right=None, right_gate=None, claimed=None,
                         claimed_obs=None):
        """Generic form (orbm_search_projected).  policy = (claim_rule, max_dist, ratio_test,
        nn_ratio, check_orientation).  Returns (nmatches, claimed, claimed_obs, query_best)."""
        n = len(u)
        keep = []

        def arr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data

        q = ProjQueries(n, arr(valid, np.uint8), arr(u, np.float32), arr(v, np.float32),
                        arr(radius, np.float32), arr(min_level, np.int32), arr(max_level, np.int32),
                        arr(np.asarray(desc).reshape(-1, 32), np.uint8), arr(angle, np.float32),
                        arr(has_obs, np.uint8), arr(u_right, np.float32), arr(right_gate, np.float32))
        pol = ProjPolicy(int(policy[0]), int(policy[1]), int(policy[2]), float(policy[3]), int(policy[4]))
        cl, co = self._claims(train.N, claimed, claimed_obs)
        qb = np.full(max(n, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self._lib.orbm_search_projected(self._m, C.byref(train.c), train.u_right.ctypes.data,
                                              C.byref(q), C.byref(pol), cl.ctypes.data,
                                              co.ctypes.data, qb.ctypes.data, C.byref(nm), MEM_HOST))
        return nm.value, cl[:train.N].copy(), co[:train.N].copy(), qb[:n].copy()

    # reference src/ORBmatcher.cc:62-155 (and RadiusByViewingCos :158-164)
    def SearchByProjectionMapPoints(self, F, valid, proj_x, proj_y, proj_xr, level, view_cos, desc,
                                    has_obs, th=3.0, claimed=None, claimed_obs=None):
        """SearchByProjection(Frame&, const vector<MapPoint*>&, th): per local map point the
        fields set by Frame::isInFrustum (mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos);
        valid = mbTrackInView && !isBad()."""
        level = np.ascontiguousarray(level, np.int32)
        # `viewCos > 0.998` compares the float against a double literal (reference :160)
        r = np.where(np.asarray(view_cos, np.float32).astype(np.float64) > 0.998, np.float32(2.5), np.float32(4.0))
        r = r.astype(np.float32)
        if float(th) != 1.0:
            r = (r * np.float32(th)).astype(np.float32)
        radius = (r * F.scale_factors[level]).astype(np.float32)
        pol = (0, self.TH_HIGH, 1, self.mfNNratio, 0)
        return self.search_projected(F, valid, proj_x, proj_y, radius, level - 1, level, desc, pol,
                                     has_obs=has_obs, u_right=proj_xr, right_gate=radius,
                                     claimed=claimed, claimed_obs=claimed_obs)

    # reference src/ORBmatcher.cc:1533-1678 (perfect/src/ORBmatcher.cc:1726 adds the point lists)
    def SearchByProjection(self, CurrentFrame, LastFrame, th, valid, u, v, invz, has_obs, bf,
                           direction=0, claimed=None, claimed_obs=None, mp_desc=None,
                           return_query_best=False):
        """Frame-to-frame projection search.  The caller supplies, per last-frame keypoint,
        the projection into the current frame (u, v, invz), whether it carries a usable map
        point (valid), whether that point has observations (has_obs) and the map points'
        descriptors (mp_desc; default: the last frame's keypoint descriptors); `direction` is
        0 (neither), 1 (forward) or 2 (backward).  Returns (nmatches, claimed, claimed_obs):
        claimed[i2] is the last-frame index assigned to current keypoint i2, or -1."""
        nl, nc = LastFrame.N, CurrentFrame.N
        valid = np.ascontiguousarray(valid, np.uint8)
        u = np.ascontiguousarray(u, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        invz = np.ascontiguousarray(invz, np.float32)
        has_obs = np.ascontiguousarray(has_obs, np.uint8)
        assert len(valid) == len(u) == len(v) == len(invz) == len(has_obs) == nl
        md = None if mp_desc is None else np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        cl, co = self._claims(nc, claimed, claimed_obs)
        qb = np.full(max(nl, 1), -1, np.int32)
        n = C.c_int(0)
        sf = CurrentFrame.scale_factors
        check(self._lib.orbm_search_by_projection(
            self._m, C.byref(CurrentFrame.c), C.byref(LastFrame.c),
            md.ctypes.data if md is not None else None, valid.ctypes.data,
            u.ctypes.data, v.ctypes.data, invz.ctypes.data, has_obs.ctypes.data,
            CurrentFrame.u_right.ctypes.data, sf.ctypes.data, len(sf), float(bf), float(th),
            int(direction), int(self.mbCheckOrientation), cl.ctypes.data, co.ctypes.data,
            qb.ctypes.data, C.byref(n), MEM_HOST))
        if return_query_best:
            return n.value, cl[:nc].copy(), co[:nc].copy(), qb[:nl].copy()
        return n.value, cl[:nc].copy(), co[:nc].copy()

    # reference src/ORBmatcher.cc:1694-1836
    def SearchByProjectionKeyFrame(self, CurrentFrame, valid, u, v, level, desc, kf_angle, th, ORBdist,
                                   claimed=None):
        """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) from the point
        where each keyframe map point is projected (u, v) and its level predicted; valid
        folds pMP != NULL, !isBad(), !sAlreadyFound.count(pMP) and the distance check."""
        level = np.ascontiguousarray(level, np.int32)
        u = np.ascontiguousarray(u, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        b = CurrentFrame.bounds
        ok = np.asarray(valid, bool) & ~((u < np.float32(b[0])) | (u > np.float32(b[2]))) & \
            ~((v < np.float32(b[1])) | (v > np.float32(b[3])))
        radius = (np.float32(th) * CurrentFrame.scale_factors[level]).astype(np.float32)
        pol = (1, int(ORBdist), 0, 0.0, int(self.mbCheckOrientation))
        return self.search_projected(CurrentFrame, ok.astype(np.uint8), u, v, radius, level - 1, level + 1,
                                     desc, pol, angle=kf_angle, claimed=claimed)

    # reference src/ORBmatcher.cc:370-485
    def SearchByProjectionSim3(self, KF, valid, u, v, level, desc, th, matched=None):
        """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) from the point where each
        candidate map point is projected with the Sim3 (u, v) and its level predicted; valid
        folds the bad/already-found/depth/distance/viewing-angle checks."""
        level = np.ascontiguousarray(level, np.int32)
        u = np.ascontiguousarray(u, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        b = KF.bounds
        inimg = (u >= np.float32(b[0])) & (u < np.float32(b[2])) & (v >= np.float32(b[1])) & (v < np.float32(b[3]))
        radius = (np.float32(int(th)) * KF.scale_factors[level]).astype(np.float32)
        pol = (1, self.TH_LOW, 0, 0.0, 0)
        return self.search_projected(KF, (np.asarray(valid, bool) & inimg).astype(np.uint8), u, v, radius,
                                     level - 1, level, desc, pol, claimed=matched)

    @staticmethod
    def projection_pair_table(pairs):
        """The C table of orbm_search_by_projection_batch for a list of pair dicts (see
        SearchByProjectionBatch); reusable across calls while the tensors stay alive."""
        arr = (_lib.ProjectionPair * len(pairs))()
        for i, p in enumerate(pairs):
            b = p["bounds"]
            arr[i].cur = MFrame(p["cur_kps"].data_ptr(), p["cur_desc"].data_ptr(), p["cur_kps"].shape[0], *b)
            arr[i].last = MFrame(p["last_kps"].data_ptr(), p["last_desc"].data_ptr(), p["last_kps"].shape[0], *b)
            arr[i].mp_desc = p["mp_desc"].data_ptr() if p.get("mp_desc") is not None else None
            for k in ("valid", "u", "v", "invz", "has_obs", "cur_u_right"):
                setattr(arr[i], k, p[k].data_ptr())
            arr[i].direction = int(p.get("direction", 0))
            arr[i].cur_claimed = p["claimed"].data_ptr()
            arr[i].cur_claimed_obs = p["claimed_obs"].data_ptr()
            arr[i].query_best = p["query_best"].data_ptr() if p.get("query_best") is not None else None
        return arr

    def SearchByProjectionBatch(self, pairs, scale_factors, bf, th):
        """Frame-to-frame SearchByProjection (reference src/ORBmatcher.cc:1533-1678) for many
        independent pairs in one call.  `pairs`: a table from projection_pair_table, or a list of
        dicts of torch DEVICE tensors with keys cur_kps (n, 28) u8, cur_desc (n, 32) u8, last_kps,
        last_desc, valid u8, u, v, invz f32, has_obs u8, cur_u_right f32, claimed i32, claimed_obs u8
        (both in/out), optional mp_desc and query_best i32, direction int, bounds (4 floats).
        Returns the list of match counts."""
        arr = self.projection_pair_table(pairs) if isinstance(pairs, list) else pairs
        sf = np.ascontiguousarray(scale_factors, np.float32)
        check(self._lib.orbm_search_by_projection_batch(self._m, arr, len(arr), sf.ctypes.data, len(sf), float(bf),
                                                        float(th), int(self.mbCheckOrientation)))
        return _nmatches(arr)

    def SearchProjectedBatch(self, jobs, policy, window_hint=0.0):
        """orbm_search_projected for many independent (frame, projected point set) jobs in one call: the batched
        form of the SearchByProjection overloads that project map points (reference src/ORBmatcher.cc:62-164,
        :370-485, :1694-1836).  `jobs`: list of dicts of torch DEVICE tensors: train_kps (n, 28) u8, train_desc
        (n, 32) u8, bounds (4 floats), valid u8, u, v, radius f32, min_level, max_level i32, desc (nq, 32) u8,
        claimed i32 and claimed_obs u8 (both in/out), optional train_u_right, angle, has_obs, u_right,
        right_gate, query_best i32.  `policy` = (claim_rule, max_dist, ratio_test, nn_ratio, check_orientation).
        Returns the list of match counts."""
        arr = (_lib.ProjectedJob * len(jobs))()
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        for i, p in enumerate(jobs):
            arr[i].train = MFrame(p["train_kps"].data_ptr(), p["train_desc"].data_ptr(), p["train_kps"].shape[0], *p["bounds"])
            arr[i].train_u_right = ptr(p.get("train_u_right"))
            arr[i].q = _lib.ProjQueries(p["valid"].shape[0], ptr(p["valid"]), ptr(p["u"]), ptr(p["v"]), ptr(p["radius"]),
                                        ptr(p["min_level"]), ptr(p["max_level"]), ptr(p["desc"]), ptr(p.get("angle")),
                                        ptr(p.get("has_obs")), ptr(p.get("u_right")), ptr(p.get("right_gate")))
            arr[i].claimed = p["claimed"].data_ptr()
            arr[i].claimed_obs = p["claimed_obs"].data_ptr()
            arr[i].query_best = ptr(p.get("query_best"))
        pol = _lib.ProjPolicy(int(policy[0]), int(policy[1]), int(policy[2]), float(policy[3]), int(policy[4]))
        check(self._lib.orbm_search_projected_batch(self._m, arr, len(arr), C.byref(pol), float(window_hint)))
        return _nmatches(arr)

    # ---- vocabulary-guided searches
    @staticmethod
    def feature_vector(fv):
        """DBoW2::FeatureVector given as {node id: [keypoint indices]} (or an ordered list of
        (node id, indices) pairs) -> (ctypes struct, arrays that must stay alive)."""
        pairs = sorted(fv.items()) if isinstance(fv, dict) else list(fv)
        ids = np.array([p[0] for p in pairs], np.uint32)
        start = np.zeros(len(pairs) + 1, np.int32)
        for i, p in enumerate(pairs):
            start[i + 1] = start[i] + len(p[1])
        items = (np.concatenate([np.asarray(p[1], np.uint32) for p in pairs]) if pairs and start[-1] > 0
                 else np.zeros(0, np.uint32)).astype(np.uint32)
        items = np.ascontiguousarray(items)
        s = _lib.FeatureVector(len(pairs), int(start[-1]), ids.ctypes.data, start.ctypes.data, items.ctypes.data)
        return s, (ids, start, items)

    def SearchByBoWBatch(self, jobs, keyframe_pair=False):
        """SearchByBoW for many independent pairs in one set of launches (device-resident).  `jobs`: list of dicts
        of torch DEVICE tensors: kps1 (n1, 28) u8, desc1 (n1, 32) u8, kps2, desc2, bounds (4 floats), valid1 u8,
        optional valid2 u8, match12 i32 (n1) and match21 i32 (n2) outputs, and the two feature vectors as
        fv1_ids / fv1_start / fv1_items, fv2_ids / fv2_start / fv2_items (u32 / i32 / u32, CSR, ids ascending).
        Returns the list of match counts."""
        arr = (_lib.BowJob * len(jobs))()
        for i, p in enumerate(jobs):
            arr[i].f1 = MFrame(p["kps1"].data_ptr(), p["desc1"].data_ptr(), p["kps1"].shape[0], *p["bounds"])
            arr[i].f2 = MFrame(p["kps2"].data_ptr(), p["desc2"].data_ptr(), p["kps2"].shape[0], *p["bounds"])
            for side in ("1", "2"):
                ids, st, it = p["fv%s_ids" % side], p["fv%s_start" % side], p["fv%s_items" % side]
                fv = _lib.FeatureVector(ids.shape[0], it.shape[0], ids.data_ptr() if ids.shape[0] else None, st.data_ptr(),
                                        it.data_ptr() if it.shape[0] else None)
                setattr(arr[i], "fv" + side, fv)
            arr[i].valid1 = p["valid1"].data_ptr()
            arr[i].valid2 = p["valid2"].data_ptr() if p.get("valid2") is not None else None
            arr[i].match12 = p["match12"].data_ptr()
            arr[i].match21 = p["match21"].data_ptr()
        check(self._lib.orbm_search_by_bow_batch(self._m, arr, len(arr), self.TH_LOW - 1 if keyframe_pair else self.TH_LOW,
                                                 float(self.mfNNratio), int(self.mbCheckOrientation)))
        return _nmatches(arr)

    def SearchByBoW(self, F1, fv1, valid1, F2, fv2, valid2=None, keyframe_pair=False):
        """ORBmatcher::SearchByBoW.  keyframe_pair=False: (KeyFrame*, Frame&) overload (reference
        src/ORBmatcher.cc:209-355), accepts distances <= TH_LOW; True: (KeyFrame*, KeyFrame*)
        (:642-788), accepts < TH_LOW and needs valid2.  Returns (nmatches, match12, match21)."""
        s1, k1 = self.feature_vector(fv1)
        s2, k2 = self.feature_vector(fv2)
        v1 = np.ascontiguousarray(valid1, np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        m12 = np.full(max(len(F1.kps), 1), -1, np.int32)
        m21 = np.full(max(len(F2.kps), 1), -1, np.int32)
        n = C.c_int(0)
        check(self._lib.orbm_search_by_bow(self._m, C.byref(F1.c), C.byref(s1), v1.ctypes.data, C.byref(F2.c), C.byref(s2),
                                           None if v2 is None else v2.ctypes.data,
                                           self.TH_LOW - 1 if keyframe_pair else self.TH_LOW, self.mfNNratio,
                                           int(self.mbCheckOrientation), m12.ctypes.data, m21.ctypes.data, C.addressof(n),
                                           MEM_HOST))
        return n.value, m12[:len(F1.kps)], m21[:len(F2.kps)]

    def SearchForTriangulation(self, F1, fv1, has_mp1, u_right1, F2, fv2, has_mp2, u_right2, F12, ex, ey,
                               level_sigma2_2, only_stereo=False):
        """ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:803-992) with the epipole
        (ex, ey) and F12 (3x3) computed by the caller.  Returns (nmatches, vMatches12)."""
        s1, k1 = self.feature_vector(fv1)
        s2, k2 = self.feature_vector(fv2)
        h1 = np.ascontiguousarray(has_mp1, np.uint8)
        h2 = np.ascontiguousarray(has_mp2, np.uint8)
        r1 = np.ascontiguousarray(u_right1, np.float32)
        r2 = np.ascontiguousarray(u_right2, np.float32)
        f12 = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf2 = np.ascontiguousarray(F2.scale_factors, np.float32)
        sg2 = np.ascontiguousarray(level_sigma2_2, np.float32)
        m12 = np.full(max(len(F1.kps), 1), -1, np.int32)
        n = C.c_int(0)
        check(self._lib.orbm_search_for_triangulation(self._m, C.byref(F1.c), C.byref(s1), h1.ctypes.data, r1.ctypes.data,
                                                      C.byref(F2.c), C.byref(s2), h2.ctypes.data, r2.ctypes.data,
                                                      f12.ctypes.data, float(ex), float(ey), sf2.ctypes.data,
                                                      sg2.ctypes.data, len(sf2), int(only_stereo),
                                                      int(self.mbCheckOrientation), m12.ctypes.data, C.addressof(n),
                                                      MEM_HOST))
        return n.value, m12[:len(F1.kps)]

    def window_best(self, KF, valid, u, v, radius, level, desc, chi2_gate=False, kf_u_right=None, q_u_right=None,
                    inv_level_sigma2=None):
        """Candidate loop of Fuse / SearchBySim3 (reference src/ORBmatcher.cc:1066-1120, :1391-1416):
        per projected map point the closest keypoint of level-1..level in the window.
        Returns (best_idx, best_dist)."""
        n = len(u)
        a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(u, np.float32),
             np.ascontiguousarray(v, np.float32), np.ascontiguousarray(radius, np.float32),
             np.ascontiguousarray(level, np.int32), np.ascontiguousarray(desc, np.uint8)]
        kur = None if kf_u_right is None else np.ascontiguousarray(kf_u_right, np.float32)
        qur = None if q_u_right is None else np.ascontiguousarray(q_u_right, np.float32)
        isg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        bi = np.full(max(n, 1), -1, np.int32)
        bd = np.full(max(n, 1), 256, np.int32)
        check(self._lib.orbm_window_best(self._m, C.byref(KF.c), n, *[x.ctypes.data for x in a], int(chi2_gate),
                                         None if kur is None else kur.ctypes.data,
                                         None if qur is None else qur.ctypes.data,
                                         None if isg is None else isg.ctypes.data, 0 if isg is None else len(isg),
                                         bi.ctypes.data, bd.ctypes.data, MEM_HOST))
        return bi[:n], bd[:n]

    def SearchBySim3(self, KF1, KF2, q12, q21):
        """ORBmatcher::SearchBySim3 (reference src/ORBmatcher.cc:1298-1517) from the projections on.
        q12 = (valid, u, v, radius, level, desc) per keypoint of KF1 (its map point projected into
        KF2), q21 the reverse.  Returns (nfound, match12)."""
        def prep(q):
            return [np.ascontiguousarray(q[0], np.uint8), np.ascontiguousarray(q[1], np.float32),
                    np.ascontiguousarray(q[2], np.float32), np.ascontiguousarray(q[3], np.float32),
                    np.ascontiguousarray(q[4], np.int32), np.ascontiguousarray(q[5], np.uint8)]
        a, b = prep(q12), prep(q21)
        assert len(a[1]) == len(KF1.kps) and len(b[1]) == len(KF2.kps)
        m12 = np.full(max(len(KF1.kps), 1), -1, np.int32)
        n = C.c_int(0)
        check(self._lib.orbm_search_by_sim3(self._m, C.byref(KF1.c), C.byref(KF2.c), *[x.ctypes.data for x in a],
                                            *[x.ctypes.data for x in b], m12.ctypes.data, C.addressof(n), MEM_HOST))
        return n.value, m12[:len(KF1.kps)]

    def ComputeDistinctiveDescriptors(self, desc, group_start):
        """MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:279-341) for a batch:
        returns (best index per group, best median per group, chosen descriptors (G, 32))."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        gs = np.ascontiguousarray(group_start, np.int32)
        g = len(gs) - 1
        best = np.full(max(g, 1), -1, np.int32)
        med = np.zeros(max(g, 1), np.int32)
        out = np.zeros((max(g, 1), 32), np.uint8)
        check(self._lib.orbm_distinctive_descriptors(self._m, d.ctypes.data, gs.ctypes.data, g, best.ctypes.data,
                                                     med.ctypes.data, out.ctypes.data, len(d), MEM_HOST))
        return best[:g], med[:g], out[:g]

    # ---- keypoint set utilities
    def filter_by_mask(self, kps, desc, mask):
        """Dynamic-object filter of reference perfect/src/Frame.cc:357-374: returns the kept
        (keypoints, descriptors)."""
        k = np.ascontiguousarray(kps, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        mk = np.ascontiguousarray(mask, np.uint8)
        ok = np.zeros(max(len(k), 1), KP_DTYPE)
        od = np.zeros((max(len(k), 1), 32), np.uint8)
        n = C.c_int(0)
        check(self._lib.orbm_filter_by_mask(self._m, k.ctypes.data, d.ctypes.data, len(k), mk.ctypes.data, mk.shape[1],
                                            mk.shape[0], mk.strides[0], ok.ctypes.data, od.ctypes.data, C.addressof(n),
                                            MEM_HOST))
        return ok[:n.value].copy(), od[:n.value].copy()

    def pack_keyframe_records(self, kps, desc, mp_index=None):
        """Keypoint records of the binary map file (reference perfect/src/Map.cc:361-380): (n*64,) uint8."""
        k = np.ascontiguousarray(kps, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        mi = None if mp_index is None else np.ascontiguousarray(mp_index, np.uint64)
        rec = np.zeros(max(len(k), 1) * 64, np.uint8)
        check(self._lib.orbm_pack_keyframe_records(self._m, k.ctypes.data, d.ctypes.data,
                                                   None if mi is None else mi.ctypes.data, len(k), rec.ctypes.data,
                                                   MEM_HOST))
        return rec[:len(k) * 64]

    def unpack_keyframe_records(self, records):
        """Inverse of pack_keyframe_records (reference perfect/src/Map.cc:169-187)."""
        r = np.ascontiguousarray(records, np.uint8)
        n = len(r) // 64
        k = np.zeros(max(n, 1), KP_DTYPE)
        d = np.zeros((max(n, 1), 32), np.uint8)
        mi = np.zeros(max(n, 1), np.uint64)
        check(self._lib.orbm_unpack_keyframe_records(self._m, r.ctypes.data, n, k.ctypes.data, d.ctypes.data,
                                                     mi.ctypes.data, MEM_HOST))
        return k[:n], d[:n], mi[:n]

    # ---- stereo
    @staticmethod
    def pyramid_of(extractor, frame=0):
        """mvImagePyramid of `frame` of the extractor's last call as device images."""
        p = _lib.Pyramid()
        p.nlevels = extractor.GetLevels()
        for l in range(p.nlevels):
            ptr, w, h, st = extractor.level_info(frame, l)
            p.level[l], p.width[l], p.height[l], p.pitch[l] = ptr, w, h, st
        return p

    def ComputeStereoMatches(self, ext_left, ext_right, keys, desc, keys_right, desc_right, mb, mbf,
                             frame_left=0, frame_right=0):
        """Frame::ComputeStereoMatches (reference src/Frame.cc:603-774): returns (mvuRight, mvDepth,
        number of stereo points).  ext_left / ext_right are the two extractors whose last call
        produced the keypoints (their pyramids are read in place on the device)."""
        kl = np.ascontiguousarray(keys, KP_DTYPE)
        kr = np.ascontiguousarray(keys_right, KP_DTYPE)
        dl = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(desc_right, np.uint8).reshape(-1, 32)
        fl = MFrame(kl.ctypes.data, dl.ctypes.data, len(kl), 0, 0, 1, 1)
        fr = MFrame(kr.ctypes.data, dr.ctypes.data, len(kr), 0, 0, 1, 1)
        pl = self.pyramid_of(ext_left, frame_left)
        pr = self.pyramid_of(ext_right, frame_right)
        sf = np.ascontiguousarray(ext_left.GetScaleFactors(), np.float32)
        isf = np.ascontiguousarray(ext_left.GetInverseScaleFactors(), np.float32)
        ur = np.full(max(len(kl), 1), -1, np.float32)
        dp = np.full(max(len(kl), 1), -1, np.float32)
        n = C.c_int(0)
        check(self._lib.orbm_compute_stereo_matches(self._m, C.byref(fl), C.byref(fr), C.byref(pl), C.byref(pr),
                                                    sf.ctypes.data, isf.ctypes.data, mb, mbf, ur.ctypes.data,
                                                    dp.ctypes.data, C.addressof(n), MEM_HOST))
        return ur[:len(kl)], dp[:len(kl)], n.value

    @staticmethod
    def stereo_job_table(jobs):
        """The C table of orbm_compute_stereo_matches_batch for a list of job dicts (see
        ComputeStereoMatchesBatch); reusable across calls while the tensors and pyramids stay alive."""
        arr = (_lib.StereoJob * len(jobs))()
        for i, p in enumerate(jobs):
            arr[i].left = MFrame(p["kps_left"].data_ptr(), p["desc_left"].data_ptr(), p["kps_left"].shape[0], 0, 0, 1, 1)
            arr[i].right = MFrame(p["kps_right"].data_ptr(), p["desc_right"].data_ptr(), p["kps_right"].shape[0], 0, 0, 1, 1)
            arr[i].pyr_left = p["pyr_left"]
            arr[i].pyr_right = p["pyr_right"]
            arr[i].u_right = p["u_right"].data_ptr()
            arr[i].depth = p["depth"].data_ptr()
        return arr

    def ComputeStereoMatchesBatch(self, jobs, ext_left, mb, mbf):
        """Frame::ComputeStereoMatches (reference src/Frame.cc:603-774) for many rectified pairs in one set
        of launches (device-resident).  `jobs`: a table from stereo_job_table, or a list of dicts: kps_left
        (n, 28) u8, desc_left (n, 32) u8, kps_right, desc_right, u_right f32 (n, out), depth f32 (n, out) as
        torch DEVICE tensors, and pyr_left / pyr_right from pyramid_of(extractor, frame).  ext_left supplies
        the scale tables.  Returns the list of stereo point counts."""
        arr = self.stereo_job_table(jobs) if isinstance(jobs, list) else jobs
        sf = np.ascontiguousarray(ext_left.GetScaleFactors(), np.float32)
        isf = np.ascontiguousarray(ext_left.GetInverseScaleFactors(), np.float32)
        check(self._lib.orbm_compute_stereo_matches_batch(self._m, arr, len(arr), sf.ctypes.data, isf.ctypes.data, len(sf),
                                                          float(mb), float(mbf)))
        if len(arr) == 0:
            return []
        raw = np.frombuffer(arr, dtype=np.uint8).reshape(len(arr), C.sizeof(_lib.StereoJob))
        off = _lib.StereoJob.n_stereo.offset
        return raw[:, off:off + 4].copy().view(np.int32).reshape(-1).tolist()

    # ---- brute force
    def knn2(self, q, t, index_base=0):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq = len(q)
        bi, bd, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
        check(self._lib.orbm_knn2(self._m, q.ctypes.data, nq, t.ctypes.data, len(t), index_base,
                                  bi.ctypes.data, bd.ctypes.data, sd.ctypes.data, MEM_HOST))
        return bi[:nq], bd[:nq], sd[:nq]

    def debug_knn_path(self, path):
        """0 = scan chosen by size, 1 = vector ALU, 2 = matrix cores (same results either way)."""
        check(self._lib.orbm_debug_knn_path(self._m, int(path)))

    def knn2_device(self, q, t, best_idx, best_d, second_d, index_base=0):
        """torch tensors on the device: q (nq, 32) u8, t (nt, 32) u8, int32 outputs (nq,)."""
        check(self._lib.orbm_knn2(self._m, q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0],
                                  index_base, best_idx.data_ptr(), best_d.data_ptr(),
                                  second_d.data_ptr(), _lib.MEM_DEVICE))

    def knn2_noalloc_device(self, q, t, best_idx, best_d, second_d, index_base=0):
        """knn2_device through the entry point that allocates nothing (static device scratch)."""
        check(self._lib.orbm_knn2_noalloc(self._m, q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0],
                                          index_base, best_idx.data_ptr(), best_d.data_ptr(),
                                          second_d.data_ptr()))

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

    # ---- DBoW2 transform (the call of Frame::ComputeBoW, reference src/Frame.cc:513-522)
    def BowTransform(self, voc, descriptors, levelsup=4):
        """TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) with the
        vocabulary given as arrays (dict: levels, node_desc, child_start, children, word_id, weight).
        The tree descent runs on the GPU; the two std::map containers are filled on the host with
        DBoW2's arithmetic (TF-IDF accumulation in feature order, L1 normalisation in word order).
        Returns (bow_ids, bow_values, {node id: [feature indices]})."""
        nd = np.ascontiguousarray(voc["node_desc"], np.uint8)
        cs = np.ascontiguousarray(voc["child_start"], np.int32)
        ch = np.ascontiguousarray(voc["children"], np.int32)
        wi = np.ascontiguousarray(voc["word_id"], np.int32)
        wt = np.ascontiguousarray(voc["weight"], np.float64)
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        cv = _lib.Vocabulary(len(nd), len(ch), int(voc["levels"]), nd.ctypes.data, cs.ctypes.data, ch.ctypes.data,
                             wi.ctypes.data, wt.ctypes.data)
        word, node = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        weight = np.zeros(max(n, 1), np.float64)
        check(self._lib.orbm_bow_transform(self._m, C.byref(cv), d.ctypes.data, n, int(levelsup), word.ctypes.data,
                                           weight.ctypes.data, node.ctypes.data, MEM_HOST))
        word, weight, node = word[:n], weight[:n], node[:n]
        bow = {}
        fv = {}
        for i in range(n):
            w = float(weight[i])
            if w > 0:                      # not a stopped word
                k = int(word[i])
                bow[k] = bow[k] + w if k in bow else w
                fv.setdefault(int(node[i]), []).append(i)
        ids = np.array(sorted(bow), np.int32)
        vals = np.array([bow[int(k)] for k in ids], np.float64)
        norm = 0.0
        for v in vals:                     # sequential, ascending word id, as BowVector::normalize does
            norm += abs(float(v))
        if norm > 0.0:
            vals = vals / norm
        return ids, vals, fv

    def ratio_test_device(self, best_d, second_d, accept, th=None):
        """Device form of ratio_test on torch tensors: int32 (nq,), int32 (nq,), uint8 (nq,) out."""
        check(self._lib.orbm_ratio_test(self._m, best_d.shape[0], best_d.data_ptr(), second_d.data_ptr(),
                                        float(self.mfNNratio), self.TH_LOW if th is None else th,
                                        accept.data_ptr(), _lib.MEM_DEVICE))

    def set_stream(self, stream_ptr):
        """Run on the hipStream_t with this handle value (0 / None = the legacy default stream)."""
        check(self._lib.orbm_set_stream(self._m, C.c_void_p(stream_ptr or None)))

    def reset_stream(self):
        check(self._lib.orbm_reset_stream(self._m))

    def synchronize(self):
        check(self._lib.orbm_synchronize(self._m))
