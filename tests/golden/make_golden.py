#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference ships no fixtures for this path and cannot be built here (it
needs OpenCV), so these vectors are outputs of the oracle (oracle/*.cpp) on
seeded synthetic inputs; "parity unpinned" with respect to a run of the
reference itself.  They pin the oracle against accidental change and give the
HIP path a fixed target that does not depend on the oracle being present.

Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from slam_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
BOUNDS = (0.0, 0.0, 640.0, 480.0)

EXTRACT_CASES = [  # name, w, h, seed, nfeatures
    ("vga_s11", 640, 480, 11, 1000),
    ("vga_s12", 640, 480, 12, 1000),
    ("vga_s13", 640, 480, 13, 1000),
    ("kitti_s21", 1241, 376, 21, 2000),
    ("qvga_s31", 320, 240, 31, 500),
]


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    for name, w, h, seed, nf in EXTRACT_CASES:
        img = synth.synth_image(w, h, seed)
        p = O.params(nfeatures=nf)
        k, d = O.extract(p, img, cap=nf + 256)
        np.savez_compressed(os.path.join(OUT, "extract_%s.npz" % name), w=w, h=h, seed=seed, nfeatures=nf,
                            image_sha1=sha(img), kps=k, desc=d)
        print(name, len(k))
    # edge cases with the image stored (small)
    small = synth.synth_image(200, 150, 41)
    k, d = O.extract(O.params(nfeatures=300), small, cap=600)
    np.savez_compressed(os.path.join(OUT, "extract_small_stored.npz"), image=small, nfeatures=300, kps=k, desc=d)
    low = (synth.synth_image(640, 480, 21).astype(np.int32) // 12 + 100).astype(np.uint8)
    k, d = O.extract(O.params(), low)
    np.savez_compressed(os.path.join(OUT, "extract_low_contrast.npz"), seed=21, image_sha1=sha(low), kps=k, desc=d)

    # matching: two shifted views of one scene
    p = O.params()
    sf = O.tables(p)["scale"]
    scene = synth.synth_image(700, 520, 77)
    a = np.ascontiguousarray(scene[20:500, 30:670])
    b = np.ascontiguousarray(scene[26:506, 21:661])
    ka, da = O.extract(p, a)
    kb, db = O.extract(p, b)
    fa, fb = O.FrameData(ka, da, BOUNDS, sf), O.FrameData(kb, db, BOUNDS, sf)
    prev = np.stack([ka["x"], ka["y"]], axis=1)
    n_init, m12, pm = O.search_for_initialization(fa, fb, prev, 100, 0.9, True)
    rng = np.random.RandomState(5)
    nl = len(ka)
    valid = (rng.uniform(size=nl) < 0.8).astype(np.uint8)
    u = (ka["x"] + 9.0 + rng.normal(0, 1.0, nl)).astype(np.float32)
    v = (ka["y"] - 6.0 + rng.normal(0, 1.0, nl)).astype(np.float32)
    invz = rng.uniform(0.05, 1.0, nl).astype(np.float32)
    has_obs = (rng.uniform(size=nl) > 0.3).astype(np.uint8)
    ur = np.full(len(kb), -1.0, np.float32)
    pick = rng.uniform(size=len(kb)) < 0.5
    ur[pick] = (kb["x"][pick] - rng.uniform(0, 40, pick.sum())).astype(np.float32)
    n_proj, cl, co = O.search_by_projection_frames(fb, fa, valid, u, v, invz, has_obs, ur, 40.0, 15.0, 0, True)
    start, items = O.grid_build(fb)
    q = synth.synth_descriptors(400, 1)
    t = synth.synth_descriptors(3000, 2)
    t[500:700] = synth.perturb_descriptors(q[:200], 4)
    bi, bd, sd = O.knn2(q, t)
    np.savez_compressed(os.path.join(OUT, "match_pair.npz"), ka=ka, da=da, kb=kb, db=db, sf=sf, prev=prev,
                        n_init=n_init, m12=m12, pm=pm, valid=valid, u=u, v=v, invz=invz, has_obs=has_obs, ur=ur,
                        n_proj=n_proj, claimed=cl, claimed_obs=co, grid_start=start, grid_items=items,
                        knn_q=q, knn_t=t, knn_idx=bi, knn_d=bd, knn_s=sd)
    print("match", n_init, n_proj)

    # stereo: KITTI-sized rectified pair, two disparity planes, noisy right view
    p = O.params(nfeatures=2000)
    left, right = synth.synth_stereo_pair(1241, 376, 61, 9, 33, 2)
    kl, dl = O.extract(p, left, cap=2300)
    kr, dr = O.extract(p, right, cap=2300)
    n, ur, dp = O.compute_stereo_matches(p, left, right, kl, dl, kr, dr, 0.54, 386.1448)
    np.savez_compressed(os.path.join(OUT, "stereo_kitti.npz"), w=1241, h=376, seed=61, dtop=9, dbot=33, noise=2,
                        nfeatures=2000, left_sha1=sha(left), right_sha1=sha(right), mb=np.float32(0.54),
                        mbf=np.float32(386.1448), n_left=len(kl), n_right=len(kr), n_stereo=n, u_right=ur, depth=dp)
    print("stereo", len(kl), len(kr), n)

    make_bow_golden()


def make_bow_golden():
    """Vocabulary-guided searches, candidate loop, Sim3 agreement, distinctive descriptors, mask
    filter and map records on one pair of views; inputs are stored with the expected outputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_bow_gpu as T
    p = O.params()
    sf = O.tables(p)["scale"]
    scene = synth.synth_image(700, 520, 91)
    a = np.ascontiguousarray(scene[20:500, 30:670])
    b = np.ascontiguousarray(scene[24:504, 18:658])
    ka, da = O.extract(p, a)
    kb, db = O.extract(p, b)
    pair = dict(sf=sf, ka=ka, da=da, kb=kb, db=db, shift=(12.0, -4.0))
    fa, fb = O.FrameData(ka, da, BOUNDS, sf), O.FrameData(kb, db, BOUNDS, sf)
    fva = T.feature_vector(T.node_of(ka, 0, 0, 97, 1), 3)
    fvb = T.feature_vector(T.node_of(kb, -12.0, 4.0, 97, 2), 4)

    def csr(fv):
        s, keep = O.feature_vector(fv)
        return keep

    r = np.random.default_rng(42)
    valid1 = (r.uniform(size=len(ka)) < 0.8).astype(np.uint8)
    valid2 = (r.uniform(size=len(kb)) < 0.7).astype(np.uint8)
    bow_a = O.search_by_bow(fa, fva, valid1, fb, fvb, None, 50, 0.75, True)
    bow_b = O.search_by_bow(fa, fva, valid1, fb, fvb, valid2, 49, 0.75, True)
    has1 = (r.uniform(size=len(ka)) < 0.3).astype(np.uint8)
    has2 = (r.uniform(size=len(kb)) < 0.3).astype(np.uint8)
    ur1 = np.where(r.uniform(size=len(ka)) < 0.5, ka["x"] - 10, -1).astype(np.float32)
    ur2 = np.where(r.uniform(size=len(kb)) < 0.5, kb["x"] - 10, -1).astype(np.float32)
    F12 = np.array([[0, 0, -4.0], [0, 0, -12.0], [4.0, 12.0, 0]], np.float32).T.copy()
    sigma2 = (sf ** 2).astype(np.float32)
    tri = O.search_for_triangulation(fa, fva, has1, ur1, fb, fvb, has2, ur2, F12, 4320.0, -1093.0, sigma2, False, True)
    v12, u12, w12, l12 = T.projected_queries(ka, (12.0, -4.0), sf, 11)
    v21, u21, w21, l21 = T.projected_queries(kb, (-12.0, 4.0), sf, 12)
    q12 = (v12, u12, w12, (7.5 * sf[l12]).astype(np.float32), l12, synth.perturb_descriptors(da, 13, 20))
    q21 = (v21, u21, w21, (7.5 * sf[l21]).astype(np.float32), l21, synth.perturb_descriptors(db, 14, 20))
    kur = np.where(r.uniform(size=len(kb)) < 0.5, kb["x"] - 8, -1).astype(np.float32)
    qur = (u12 - 8).astype(np.float32)
    isg = (1.0 / sf ** 2).astype(np.float32)
    wb = O.window_best(fb, q12[0], q12[1], q12[2], q12[3], q12[4], q12[5], True, kur, qur, isg)
    sim3 = O.search_by_sim3(fa, fb, q12, q21)
    gs = np.concatenate([[0], np.cumsum(r.integers(0, 12, 80))]).astype(np.int32)
    dd = synth.perturb_descriptors(np.repeat(synth.synth_descriptors(80, 5), np.diff(gs), axis=0), 6, 60)
    dbest, dmed = O.distinctive_descriptors(dd, gs)
    mask = np.ones((480, 640), np.uint8)
    mask[120:330, 180:400] = 0
    mk, md = O.filter_by_mask(ka, da, mask)
    mi = r.integers(0, 9000, len(ka)).astype(np.uint64)
    rec = O.pack_keyframe_records(ka, da, mi)
    ia, sa, ta = csr(fva)
    ib, sb, tb = csr(fvb)
    np.savez_compressed(os.path.join(OUT, "bow_pair.npz"), ka=ka, da=da, kb=kb, db=db, sf=sf, fva_ids=ia, fva_start=sa,
                        fva_items=ta, fvb_ids=ib, fvb_start=sb, fvb_items=tb, valid1=valid1, valid2=valid2,
                        bow_a_n=bow_a[0], bow_a_12=bow_a[1], bow_a_21=bow_a[2], bow_b_n=bow_b[0], bow_b_12=bow_b[1],
                        bow_b_21=bow_b[2], has1=has1, has2=has2, ur1=ur1, ur2=ur2, F12=F12, sigma2=sigma2, tri_n=tri[0],
                        tri_12=tri[1], q12_valid=q12[0], q12_u=q12[1], q12_v=q12[2], q12_r=q12[3], q12_level=q12[4],
                        q12_desc=q12[5], q21_valid=q21[0], q21_u=q21[1], q21_v=q21[2], q21_r=q21[3], q21_level=q21[4],
                        q21_desc=q21[5], kur=kur, qur=qur, isg=isg, wb_idx=wb[0], wb_dist=wb[1], sim3_n=sim3[0],
                        sim3_12=sim3[1], dd_start=gs, dd_desc=dd, dd_best=dbest, dd_med=dmed, mask_box=[120, 330, 180, 400],
                        mask_kept=len(mk), mask_sha1=sha(np.concatenate([mk.view(np.uint8).ravel(), md.ravel()])),
                        mp_index=mi, rec_sha1=sha(rec))
    print("bow", bow_a[0], bow_b[0], tri[0], sim3[0])


if __name__ == "__main__":
    main()
