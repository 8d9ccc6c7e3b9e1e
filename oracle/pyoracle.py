"""ctypes access to the CPU parity checker (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The product package (slam_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("blur_variant", C.c_int),
                ("fma_variant", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float),
                ("max_y", C.c_float), ("scale_factors", C.c_void_p), ("nlevels", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_fast_atan2.restype = C.c_float
        _lib.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.oracle_ic_angle.restype = C.c_float
        _lib.oracle_cv_round.argtypes = [C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def params(nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, blur_variant=0, fma_variant=0):
    return Params(nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant, fma_variant)


def tables(p):
    n = p.nlevels
    sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
    q = np.zeros(n, np.int32)
    um = np.zeros(16, np.int32)
    lib().oracle_tables(C.byref(p), _p(sc), _p(inv), _p(s2), _p(is2), _p(q), _p(um))
    return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quotas=q, umax=um)


def level_size(p, w, h, level):
    lw, lh = C.c_int(), C.c_int()
    lib().oracle_level_size(C.byref(p), w, h, level, C.byref(lw), C.byref(lh))
    return lw.value, lh.value


def cv_round(v):
    return lib().oracle_cv_round(float(v))


def sincosf(angles):
    a = np.ascontiguousarray(angles, np.float32)
    s, c = np.zeros_like(a), np.zeros_like(a)
    lib().oracle_sincosf(_p(a), len(a), _p(s), _p(c))
    return s, c


def fast_atan2(y, x):
    return lib().oracle_fast_atan2(float(y), float(x))


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear(_p(src), src.shape[1], src.shape[0], C.c_size_t(src.strides[0]),
                               _p(dst), dw, dh, C.c_size_t(dw))
    return dst


def gaussian7(src, variant=0):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().oracle_gaussian7(_p(src), src.shape[1], src.shape[0], C.c_size_t(src.strides[0]),
                           _p(dst), C.c_size_t(dst.strides[0]), variant)
    return dst


def fast9_nms(roi, threshold):
    roi = np.ascontiguousarray(roi, np.uint8)
    cap = roi.size
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().oracle_fast9_nms(_p(roi), roi.shape[1], roi.shape[0], C.c_size_t(roi.strides[0]),
                               threshold, _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def pyramid_level(p, img, level):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    lw, lh = level_size(p, w, h, level)
    dst = np.zeros((lh, lw), np.uint8)
    lib().oracle_pyramid_level(C.byref(p), _p(img), w, h, C.c_size_t(img.strides[0]), level, _p(dst))
    return dst


def detect_cells(lvl, ini_th=20, min_th=7):
    lvl = np.ascontiguousarray(lvl, np.uint8)
    cap = max(16, lvl.size // 4 + 1024)
    xs, ys, rs = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().oracle_detect_cells(_p(lvl), lvl.shape[1], lvl.shape[0], C.c_size_t(lvl.strides[0]),
                                  ini_th, min_th, _p(xs), _p(ys), _p(rs), cap)
    assert n <= cap
    return xs[:n].copy(), ys[:n].copy(), rs[:n].copy()


def distribute(xs, ys, rs, min_x, max_x, min_y, max_y, n_target):
    xs = np.ascontiguousarray(xs, np.int32)
    ys = np.ascontiguousarray(ys, np.int32)
    rs = np.ascontiguousarray(rs, np.int32)
    cap = len(xs) + 16
    out = np.zeros(cap, np.int32)
    n = lib().oracle_distribute(_p(xs), _p(ys), _p(rs), len(xs), min_x, max_x, min_y, max_y,
                                n_target, _p(out), cap)
    return out[:n].copy()


def ic_angle(lvl, x, y):
    lvl = np.ascontiguousarray(lvl, np.uint8)
    return lib().oracle_ic_angle(_p(lvl), C.c_size_t(lvl.strides[0]), int(x), int(y))


def descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().oracle_descriptor(_p(blurred), C.c_size_t(blurred.strides[0]), int(x), int(y),
                            C.c_float(angle_deg), _p(d))
    return d


def extract(p, img, cap=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = cap or p.nfeatures + 64
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int()
    rc = lib().oracle_extract(C.byref(p), _p(img), w, h, C.c_size_t(img.strides[0]), _p(kps),
                              _p(desc), cap, C.byref(n))
    if rc:
        raise RuntimeError("oracle_extract failed rc=%d" % rc)
    return kps[:n.value].copy(), desc[:n.value].copy()


def extract_batch(p, imgs, cap=None, threads=1):
    imgs = np.ascontiguousarray(imgs, np.uint8)
    b, h, w = imgs.shape
    cap = cap or p.nfeatures + 64
    kps = np.zeros((b, cap), KP_DTYPE)
    desc = np.zeros((b, cap, 32), np.uint8)
    n = np.zeros(b, np.int32)
    rc = lib().oracle_extract_batch(C.byref(p), _p(imgs), b, w, h, C.c_size_t(imgs.strides[1]),
                                    C.c_size_t(imgs.strides[0]), _p(kps), _p(desc), cap, _p(n),
                                    threads)
    if rc:
        raise RuntimeError("oracle_extract_batch failed rc=%d" % rc)
    return kps, desc, n


def extract_stages(p, imgs):
    """Single-threaded extraction of a batch; seconds per stage as a dict."""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    b, h, w = imgs.shape
    t = np.zeros(6, np.float64)
    rc = lib().oracle_extract_stages(C.byref(p), _p(imgs), b, w, h, C.c_size_t(imgs.strides[1]),
                                     C.c_size_t(imgs.strides[0]), _p(t))
    if rc:
        raise RuntimeError("oracle_extract_stages failed rc=%d" % rc)
    return dict(zip(("pyramid", "fast", "quadtree", "orientation", "blur", "describe"), t.tolist()))


def compute_stereo_matches(p, img_left, img_right, kl, dl, kr, dr, mb, mbf):
    il = np.ascontiguousarray(img_left, np.uint8)
    ir = np.ascontiguousarray(img_right, np.uint8)
    assert il.shape == ir.shape
    kl = np.ascontiguousarray(kl, KP_DTYPE)
    kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    ur = np.zeros(max(len(kl), 1), np.float32)
    dp = np.zeros(max(len(kl), 1), np.float32)
    n = lib().oracle_compute_stereo_matches(C.byref(p), _p(il), _p(ir), il.shape[1], il.shape[0],
                                            C.c_size_t(il.strides[0]), _p(kl), _p(dl), len(kl), _p(kr), _p(dr),
                                            len(kr), C.c_float(mb), C.c_float(mbf), _p(ur), _p(dp))
    return n, ur[:len(kl)].copy(), dp[:len(kl)].copy()


# ---------------------------------------------------------------- matching
class FrameData:
    """Keeps the numpy buffers alive behind an oracle_frame struct."""

    def __init__(self, kps, desc, bounds, scale_factors):
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        self.bounds = tuple(float(b) for b in bounds)
        self.c = Frame(self.kps.ctypes.data, self.desc.ctypes.data, len(self.kps), self.bounds[0],
                       self.bounds[1], self.bounds[2], self.bounds[3], self.sf.ctypes.data,
                       len(self.sf))

    @property
    def n(self):
        return len(self.kps)


def hamming256(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_hamming256(_p(a), _p(b))


def grid_build(fr):
    start = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(max(fr.n, 1), np.int32)
    lib().oracle_grid_build(C.byref(fr.c), _p(start), _p(items))
    return start, items[:start[-1]].copy()


def features_in_area(fr, start, items, x, y, r, min_level=-1, max_level=-1):
    out = np.zeros(max(fr.n, 1), np.int32)
    items = np.ascontiguousarray(items, np.int32)
    n = lib().oracle_features_in_area(C.byref(fr.c), _p(start), _p(items), C.c_float(x),
                                      C.c_float(y), C.c_float(r), min_level, max_level, _p(out),
                                      len(out))
    return out[:n].copy()


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().oracle_three_maxima(_p(sizes), len(sizes), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def search_for_initialization(f1, f2, prev_matched, window_size=100, nn_ratio=0.9, check_ori=True):
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(f1.n, 1), np.int32)
    n = lib().oracle_search_for_initialization(C.byref(f1.c), C.byref(f2.c), _p(pm), _p(m12),
                                               int(window_size), C.c_float(nn_ratio),
                                               int(bool(check_ori)))
    return n, m12[:f1.n].copy(), pm


def search_by_projection_frames(cur, last, valid, u, v, invz, has_obs, u_right, bf, th, direction,
                                check_ori=True, claimed=None, claimed_obs=None, mp_desc=None,
                                return_query_best=False):
    valid = np.ascontiguousarray(valid, np.uint8)
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    invz = np.ascontiguousarray(invz, np.float32)
    has_obs = np.ascontiguousarray(has_obs, np.uint8)
    u_right = np.ascontiguousarray(u_right, np.float32)
    md = None if mp_desc is None else np.ascontiguousarray(mp_desc, np.uint8)
    cl = np.full(max(cur.n, 1), -1, np.int32) if claimed is None else np.array(claimed, np.int32)
    co = np.zeros(max(cur.n, 1), np.uint8) if claimed_obs is None else np.array(claimed_obs, np.uint8)
    qb = np.full(max(last.n, 1), -1, np.int32)
    n = lib().oracle_search_by_projection_frames(
        C.byref(cur.c), C.byref(last.c), _p(md) if md is not None else None, _p(valid), _p(u), _p(v),
        _p(invz), _p(has_obs), _p(u_right), C.c_float(bf), C.c_float(th), int(direction),
        C.c_float(0.9), int(bool(check_ori)), _p(cl), _p(co), _p(qb))
    if return_query_best:
        return n, cl[:cur.n].copy(), co[:cur.n].copy(), qb[:last.n].copy()
    return n, cl[:cur.n].copy(), co[:cur.n].copy()


def project_last_frame(Tcw, Tlw, x3dw, fx, fy, cx, cy, mb, mono):
    """Pose algebra of reference src/ORBmatcher.cc:1545-1556, :1570-1578 -> (u, v, invz, direction)."""
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(16)
    X = np.ascontiguousarray(x3dw, np.float32).reshape(-1, 3)
    n = len(X)
    u, v, iz = (np.zeros(max(n, 1), np.float32) for _ in range(3))
    d = C.c_int(0)
    lib().oracle_project_last_frame(_p(Tcw), _p(Tlw), _p(X), n, C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                    C.c_float(cy), C.c_float(mb), int(bool(mono)), _p(u), _p(v), _p(iz), C.byref(d))
    return u[:n], v[:n], iz[:n], d.value


def search_by_projection_mappoints(f, f_u_right, valid, px, py, pxr, level, view_cos, desc, has_obs, th,
                                   nn_ratio, claimed=None, claimed_obs=None):
    valid = np.ascontiguousarray(valid, np.uint8)
    px = np.ascontiguousarray(px, np.float32)
    py = np.ascontiguousarray(py, np.float32)
    pxr = np.ascontiguousarray(pxr, np.float32)
    level = np.ascontiguousarray(level, np.int32)
    view_cos = np.ascontiguousarray(view_cos, np.float32)
    desc = np.ascontiguousarray(desc, np.uint8)
    has_obs = np.ascontiguousarray(has_obs, np.uint8)
    ur = np.ascontiguousarray(f_u_right, np.float32)
    cl = np.full(max(f.n, 1), -1, np.int32) if claimed is None else np.array(claimed, np.int32)
    co = np.zeros(max(f.n, 1), np.uint8) if claimed_obs is None else np.array(claimed_obs, np.uint8)
    n = lib().oracle_search_by_projection_mappoints(
        C.byref(f.c), _p(ur), len(px), _p(valid), _p(px), _p(py), _p(pxr), _p(level), _p(view_cos),
        _p(desc), _p(has_obs), C.c_float(th), C.c_float(nn_ratio), _p(cl), _p(co))
    return n, cl[:f.n].copy(), co[:f.n].copy()


def search_by_projection_keyframe(cur, valid, u, v, level, desc, kf_angle, th, orb_dist, check_ori=True,
                                  claimed=None):
    valid = np.ascontiguousarray(valid, np.uint8)
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    level = np.ascontiguousarray(level, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    kf_angle = np.ascontiguousarray(kf_angle, np.float32)
    cl = np.full(max(cur.n, 1), -1, np.int32) if claimed is None else np.array(claimed, np.int32)
    n = lib().oracle_search_by_projection_keyframe(C.byref(cur.c), len(u), _p(valid), _p(u), _p(v), _p(level),
                                                   _p(desc), _p(kf_angle), C.c_float(th), int(orb_dist),
                                                   int(bool(check_ori)), _p(cl))
    return n, cl[:cur.n].copy()


def search_by_projection_sim3(kf, valid, u, v, level, desc, th, matched=None):
    valid = np.ascontiguousarray(valid, np.uint8)
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    level = np.ascontiguousarray(level, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    m = np.full(max(kf.n, 1), -1, np.int32) if matched is None else np.array(matched, np.int32)
    n = lib().oracle_search_by_projection_sim3(C.byref(kf.c), len(u), _p(valid), _p(u), _p(v), _p(level),
                                               _p(desc), int(th), _p(m))
    return n, m[:kf.n].copy()


def knn2(q, t, threads=1):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    nq = len(q)
    bi, bd, sd = (np.zeros(nq, np.int32) for _ in range(3))
    if threads > 1:
        lib().oracle_knn2_mt(_p(q), nq, _p(t), len(t), _p(bi), _p(bd), _p(sd), int(threads))
    else:
        lib().oracle_knn2(_p(q), nq, _p(t), len(t), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


# ---------------------------------------------------------------- DBoW2 transform
def bow_transform(voc, desc, levelsup=4):
    """voc: dict with levels, node_desc (n, 32) u8, child_start (n+1) i32, children i32, word_id i32,
    weight f64.  Returns (word, weight, node) per descriptor."""
    nd = np.ascontiguousarray(voc["node_desc"], np.uint8)
    cs = np.ascontiguousarray(voc["child_start"], np.int32)
    ch = np.ascontiguousarray(voc["children"], np.int32)
    wi = np.ascontiguousarray(voc["word_id"], np.int32)
    wt = np.ascontiguousarray(voc["weight"], np.float64)
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(d)
    word, node = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    weight = np.zeros(max(n, 1), np.float64)
    f = lib().oracle_bow_transform
    f.restype = None
    f(len(nd), int(voc["levels"]), _p(nd), _p(cs), _p(ch), _p(wi), _p(wt), _p(d), n, int(levelsup), _p(word), _p(weight),
      _p(node))
    return word[:n], weight[:n], node[:n]


def bow_vector(word, weight):
    word = np.ascontiguousarray(word, np.int32)
    weight = np.ascontiguousarray(weight, np.float64)
    ids = np.zeros(max(len(word), 1), np.int32)
    vals = np.zeros(max(len(word), 1), np.float64)
    k = lib().oracle_bow_vector(_p(word), _p(weight), len(word), _p(ids), _p(vals))
    return ids[:k].copy(), vals[:k].copy()


# ---------------------------------------------------------------- vocabulary-guided searches
class FeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("node_ids", C.c_void_p), ("start", C.c_void_p), ("items", C.c_void_p)]


def feature_vector(fv):
    pairs = sorted(fv.items()) if isinstance(fv, dict) else list(fv)
    ids = np.array([p[0] for p in pairs], np.uint32)
    start = np.zeros(len(pairs) + 1, np.int32)
    for i, p in enumerate(pairs):
        start[i + 1] = start[i] + len(p[1])
    items = np.ascontiguousarray(np.concatenate([np.asarray(p[1], np.uint32) for p in pairs]).astype(np.uint32)
                                 if pairs and start[-1] > 0 else np.zeros(0, np.uint32))
    return FeatVec(len(pairs), ids.ctypes.data, start.ctypes.data, items.ctypes.data), (ids, start, items)


def search_by_bow(f1, fv1, valid1, f2, fv2, valid2, max_dist, nn_ratio, check_ori):
    s1, k1 = feature_vector(fv1)
    s2, k2 = feature_vector(fv2)
    v1 = np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    m12 = np.full(max(f1.n, 1), -1, np.int32)
    m21 = np.full(max(f2.n, 1), -1, np.int32)
    fn = lib().oracle_search_by_bow
    fn.restype = C.c_int
    n = fn(C.byref(f1.c), C.byref(s1), _p(v1), C.byref(f2.c), C.byref(s2), None if v2 is None else _p(v2),
           C.c_int(max_dist), C.c_float(nn_ratio), C.c_int(int(check_ori)), _p(m12), _p(m21))
    return n, m12[:f1.n], m21[:f2.n]


def search_for_triangulation(f1, fv1, has_mp1, ur1, f2, fv2, has_mp2, ur2, F12, ex, ey, level_sigma2, only_stereo,
                             check_ori):
    s1, k1 = feature_vector(fv1)
    s2, k2 = feature_vector(fv2)
    a = [np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(ur1, np.float32),
         np.ascontiguousarray(has_mp2, np.uint8), np.ascontiguousarray(ur2, np.float32),
         np.ascontiguousarray(F12, np.float32).reshape(9), np.ascontiguousarray(level_sigma2, np.float32)]
    m12 = np.full(max(f1.n, 1), -1, np.int32)
    fn = lib().oracle_search_for_triangulation
    fn.restype = C.c_int
    n = fn(C.byref(f1.c), C.byref(s1), _p(a[0]), _p(a[1]), C.byref(f2.c), C.byref(s2), _p(a[2]), _p(a[3]), _p(a[4]),
           C.c_float(ex), C.c_float(ey), _p(a[5]), C.c_int(int(only_stereo)), C.c_int(int(check_ori)), _p(m12))
    return n, m12[:f1.n]


def window_best(kf, valid, u, v, radius, level, desc, chi2_gate=False, kf_u_right=None, q_u_right=None,
                inv_level_sigma2=None):
    n = len(u)
    a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32),
         np.ascontiguousarray(radius, np.float32), np.ascontiguousarray(level, np.int32),
         np.ascontiguousarray(desc, np.uint8)]
    kur = None if kf_u_right is None else np.ascontiguousarray(kf_u_right, np.float32)
    qur = None if q_u_right is None else np.ascontiguousarray(q_u_right, np.float32)
    isg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    bi = np.full(max(n, 1), -1, np.int32)
    bd = np.full(max(n, 1), 256, np.int32)
    fn = lib().oracle_window_best
    fn.restype = None
    fn(C.byref(kf.c), C.c_int(n), *[_p(x) for x in a], C.c_int(int(chi2_gate)), None if kur is None else _p(kur),
       None if qur is None else _p(qur), None if isg is None else _p(isg), _p(bi), _p(bd))
    return bi[:n], bd[:n]


def search_by_sim3(kf1, kf2, q12, q21):
    def prep(q):
        return [np.ascontiguousarray(q[0], np.uint8), np.ascontiguousarray(q[1], np.float32),
                np.ascontiguousarray(q[2], np.float32), np.ascontiguousarray(q[3], np.float32),
                np.ascontiguousarray(q[4], np.int32), np.ascontiguousarray(q[5], np.uint8)]
    a, b = prep(q12), prep(q21)
    m12 = np.full(max(kf1.n, 1), -1, np.int32)
    fn = lib().oracle_search_by_sim3
    fn.restype = C.c_int
    n = fn(C.byref(kf1.c), C.byref(kf2.c), *[_p(x) for x in a], *[_p(x) for x in b], _p(m12))
    return n, m12[:kf1.n]


def distinctive_descriptors(desc, group_start):
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    gs = np.ascontiguousarray(group_start, np.int32)
    g = len(gs) - 1
    best = np.full(max(g, 1), -1, np.int32)
    med = np.zeros(max(g, 1), np.int32)
    fn = lib().oracle_distinctive_descriptors
    fn.restype = None
    fn(_p(d), _p(gs), C.c_int(g), _p(best), _p(med))
    return best[:g], med[:g]


def filter_by_mask(kps, desc, mask):
    k = np.ascontiguousarray(kps, KP_DTYPE)
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    m = np.ascontiguousarray(mask, np.uint8)
    ok = np.zeros(max(len(k), 1), KP_DTYPE)
    od = np.zeros((max(len(k), 1), 32), np.uint8)
    fn = lib().oracle_filter_by_mask
    fn.restype = C.c_int
    n = fn(_p(k), _p(d), C.c_int(len(k)), _p(m), C.c_int(m.shape[1]), C.c_int(m.shape[0]), C.c_size_t(m.strides[0]),
           _p(ok), _p(od))
    return ok[:n].copy(), od[:n].copy()


def pack_keyframe_records(kps, desc, mp_index):
    k = np.ascontiguousarray(kps, KP_DTYPE)
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    mi = None if mp_index is None else np.ascontiguousarray(mp_index, np.uint64)
    rec = np.zeros(max(len(k), 1) * 64, np.uint8)
    fn = lib().oracle_pack_keyframe_records
    fn.restype = None
    fn(_p(k), _p(d), None if mi is None else _p(mi), C.c_int(len(k)), _p(rec))
    return rec[:len(k) * 64]


def unpack_keyframe_records(rec):
    r = np.ascontiguousarray(rec, np.uint8)
    n = len(r) // 64
    k = np.zeros(max(n, 1), KP_DTYPE)
    d = np.zeros((max(n, 1), 32), np.uint8)
    mi = np.zeros(max(n, 1), np.uint64)
    fn = lib().oracle_unpack_keyframe_records
    fn.restype = None
    fn(_p(r), C.c_int(n), _p(k), _p(d), _p(mi))
    return k[:n], d[:n], mi[:n]
