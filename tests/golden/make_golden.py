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


if __name__ == "__main__":
    main()
