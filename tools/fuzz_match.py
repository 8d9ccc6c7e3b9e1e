#!/usr/bin/env python3
"""Randomised parity sweep of the HIP matchers against the CPU oracle.

Frames are synthetic keypoint sets (no images): two views of one random point set with a
shift, noise, dropped points and points outside the image bounds.  Descriptors are noisy copies
of a small number of prototypes, so equal distances - where the candidate order decides - are
common.  Every search is run with randomly drawn parameters and compared bit for bit.

    python3 tools/fuzz_match.py --seed 1 --seconds 120
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as oracle  # noqa: E402
from slam_amd._lib import KP_DTYPE  # noqa: E402
from slam_amd.matcher import FrameView, ORBmatcher  # noqa: E402

NLEVELS = 8
SF = (np.float32(1.2) ** np.arange(NLEVELS)).astype(np.float32)


def flip_bits(desc, rng, max_flips):
    d = desc.copy()
    if max_flips <= 0 or len(d) == 0:
        return d
    for i in range(len(d)):
        k = int(rng.integers(0, max_flips + 1))
        if k:
            bits = rng.integers(0, 256, k)
            np.bitwise_xor.at(d[i], bits >> 3, (1 << (bits & 7)).astype(np.uint8))
    return d


def draw_case(rng, npts=None):
    w = float(rng.integers(200, 1300))
    h = float(rng.integers(150, 800))
    off = rng.uniform(-9, 9, 4) if rng.uniform() < 0.5 else np.zeros(4)
    bounds = (float(np.float32(off[0])), float(np.float32(off[1])), float(np.float32(w + off[2])), float(np.float32(h + off[3])))
    npts = int(rng.choice([0, 1, 3, 40, 300, 1000, 2200])) if npts is None else npts
    nproto = int(rng.choice([1, 4, 32, 100000]))
    max_flips = int(rng.choice([0, 2, 12, 40]))
    shift = (float(rng.integers(-20, 21)), float(rng.integers(-12, 13)))
    integer_xy = rng.uniform() < 0.3
    protos = rng.integers(0, 256, (min(nproto, max(npts, 1)), 32), dtype=np.uint8)
    px = rng.uniform(bounds[0] - 4, bounds[2] + 4, npts)
    py = rng.uniform(bounds[1] - 4, bounds[3] + 4, npts)
    if npts and rng.uniform() < 0.3:  # clusters: many keypoints in few grid cells
        c = rng.integers(0, npts, npts // 2)
        px[: len(c)] = px[c] + rng.normal(0, 3, len(c))
        py[: len(c)] = py[c] + rng.normal(0, 3, len(c))
    octv = np.minimum(rng.geometric(0.45, npts) - 1, NLEVELS - 1).astype(np.int32)
    ang = rng.uniform(0, 360, npts)
    pid = rng.integers(0, len(protos), npts)
    rot = float(rng.uniform(-40, 40))

    def view(dx, dy, drot, keep_p):
        keep = rng.uniform(size=npts) < keep_p
        n = int(keep.sum())
        k = np.zeros(n, KP_DTYPE)
        x = px[keep] + dx + rng.normal(0, 0.7, n)
        y = py[keep] + dy + rng.normal(0, 0.7, n)
        if integer_xy:
            x, y = np.round(x), np.round(y)
        k["x"], k["y"] = x.astype(np.float32), y.astype(np.float32)
        k["octave"] = np.clip(octv[keep] + (rng.uniform(size=n) < 0.15) * rng.integers(-1, 2, n), 0, NLEVELS - 1)
        k["angle"] = np.mod(ang[keep] + drot + rng.normal(0, 4, n), 360).astype(np.float32)
        k["size"] = 31 * SF[k["octave"]]
        k["response"] = rng.integers(7, 120, n)
        k["class_id"] = -1
        d = flip_bits(protos[pid[keep]], rng, max_flips)
        perm = rng.permutation(n)
        return k[perm], np.ascontiguousarray(d[perm]), np.nonzero(keep)[0][perm]

    ka, da, ida = view(0, 0, 0, float(rng.choice([1.0, 0.8, 0.3])))
    kb, db, idb = view(shift[0], shift[1], rot, float(rng.choice([1.0, 0.8, 0.3])))
    return dict(bounds=bounds, shift=shift, ka=ka, da=da, kb=kb, db=db, ida=ida, idb=idb, max_flips=max_flips, npts=npts)


STATS = {}


def eq(name, a, b):
    if isinstance(a, (int, np.integer)):
        STATS[name] = STATS.get(name, 0) + int(b)
        if a != b:
            return "%s: %s vs oracle %s" % (name, a, b)
        return None
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b):
        bad = np.nonzero(a.reshape(-1) != b.reshape(-1))[0] if a.shape == b.shape else []
        return "%s differs (shape %s vs %s, first at %s)" % (name, a.shape, b.shape, bad[:5])
    return None


def first(*errs):
    for e in errs:
        if e:
            return e
    return None


def queries_from(rng, src_k, src_d, shift, bounds, max_flips):
    """Map points seen from the other view, projected with noise; a few far off or outside."""
    n = len(src_k)
    u = (src_k["x"] + shift[0] + rng.normal(0, 1.5, n)).astype(np.float32)
    v = (src_k["y"] + shift[1] + rng.normal(0, 1.5, n)).astype(np.float32)
    far = rng.uniform(size=n) < 0.1
    u[far] = rng.uniform(bounds[0] - 30, bounds[2] + 30, int(far.sum())).astype(np.float32)
    v[far] = rng.uniform(bounds[1] - 30, bounds[3] + 30, int(far.sum())).astype(np.float32)
    if n and rng.uniform() < 0.3:  # exactly on the borders
        u[: min(n, 4)] = [bounds[0], bounds[2], bounds[0], bounds[2]][: min(n, 4)]
    level = np.clip(src_k["octave"] + rng.integers(-1, 2, n), 0, NLEVELS - 1).astype(np.int32)
    desc = flip_bits(src_d, rng, max_flips)
    valid = (rng.uniform(size=n) < 0.85).astype(np.uint8)
    return valid, u, v, level, desc


def node_lists(rng, scene_id, nodes):
    fv = {}
    node = (scene_id * 2654435761 % 1000003) % nodes
    wrong = rng.uniform(size=len(scene_id)) < 0.05
    node = np.where(wrong, rng.integers(0, nodes, len(scene_id)), node)
    for i, nd in enumerate(node):
        fv.setdefault(int(nd) * 3 + 1, []).append(i)
    for k in list(fv):
        if rng.uniform() < 0.1:
            fv[k] = [int(t) for t in rng.permutation(fv[k])]
    return fv


def run_case(rng, c):
    b = c["bounds"]
    ka, da, kb, db = c["ka"], c["da"], c["kb"], c["db"]
    na, nb = len(ka), len(kb)
    ur_a = np.where(rng.uniform(size=na) < 0.5, ka["x"] - rng.uniform(0, 40, na), -1).astype(np.float32)
    ur_b = np.where(rng.uniform(size=nb) < 0.5, kb["x"] - rng.uniform(0, 40, nb), -1).astype(np.float32)
    ga, gb = FrameView(ka, da, b, SF, u_right=ur_a), FrameView(kb, db, b, SF, u_right=ur_b)
    oa, ob = oracle.FrameData(ka, da, b, SF), oracle.FrameData(kb, db, b, SF)
    ratio = float(rng.choice([0.6, 0.75, 0.9]))
    ori = bool(rng.integers(0, 2))
    m = ORBmatcher(ratio, ori)
    sx, sy = c["shift"]
    mf = c["max_flips"]
    try:
        # grid
        s, it = m.AssignFeaturesToGrid(gb)
        rs, rit = oracle.grid_build(ob)
        e = first(eq("grid start", s, rs), eq("grid items", it, rit))
        if e:
            return e
        # initialization
        if na:
            win = int(rng.choice([10, 30, 100]))
            prev = np.stack([ka["x"] + sx, ka["y"] + sy], axis=1).astype(np.float32)
            n, m12, pm = m.SearchForInitialization(ga, gb, prev, win)
            rn, rm12, rpm = oracle.search_for_initialization(oa, ob, prev, win, ratio, ori)
            e = first(eq("init n", n, rn), eq("init m12", m12, rm12), eq("init prev", pm, rpm.reshape(-1, 2)))
            if e:
                return e
        # frame-to-frame projection
        valid, u, v, level, desc = queries_from(rng, ka, da, (sx, sy), b, mf)
        invz = rng.uniform(0.05, 1.0, na).astype(np.float32)
        invz[rng.uniform(size=na) < 0.03] = -0.1
        has_obs = (rng.uniform(size=na) < 0.7).astype(np.uint8)
        claimed = np.full(max(nb, 1), -1, np.int32)
        cobs = np.zeros(max(nb, 1), np.uint8)
        if rng.uniform() < 0.4 and nb:
            pick = rng.uniform(size=nb) < 0.25
            claimed[:nb][pick] = 5
            cobs[:nb][pick] = rng.uniform(size=int(pick.sum())) < 0.5
        direction = int(rng.integers(0, 3))
        th = float(rng.choice([7.0, 15.0, 30.0]))
        mpd = desc if rng.uniform() < 0.5 else None
        n, cl, co, qb = m.SearchByProjection(gb, ga, th, valid, u, v, invz, has_obs, 40.0, direction, claimed=claimed,
                                             claimed_obs=cobs, mp_desc=mpd, return_query_best=True)
        rn, rcl, rco, rqb = oracle.search_by_projection_frames(ob, oa, valid, u, v, invz, has_obs, ur_b, 40.0, th,
                                                               direction, ori, claimed, cobs, mp_desc=mpd,
                                                               return_query_best=True)
        e = first(eq("proj n", n, rn), eq("proj claimed", cl, rcl), eq("proj obs", co, rco), eq("proj query_best", qb, rqb))
        if e:
            return e
        # local map points
        view_cos = rng.choice(np.array([0.9, 0.998, 0.99800003, 0.9995], np.float32), na)
        pxr = (u - rng.uniform(0, 30, na)).astype(np.float32)
        th2 = float(rng.choice([1.0, 3.0, 5.0]))
        n, cl, co, _ = m.SearchByProjectionMapPoints(gb, valid, u, v, pxr, level, view_cos, desc, has_obs, th2, claimed, cobs)
        rn, rcl, rco = oracle.search_by_projection_mappoints(ob, ur_b, valid, u, v, pxr, level, view_cos, desc, has_obs,
                                                             th2, ratio, claimed, cobs)
        e = first(eq("mappoints n", n, rn), eq("mappoints claimed", cl, rcl), eq("mappoints obs", co, rco))
        if e:
            return e
        # keyframe (relocalisation) and Sim3 (loop closing) projections
        kfang = rng.uniform(0, 360, na).astype(np.float32)
        th3, od = float(rng.choice([3.0, 10.0])), int(rng.choice([64, 100]))
        n, cl, _, _ = m.SearchByProjectionKeyFrame(gb, valid, u, v, level, desc, kfang, th3, od, claimed)
        rn, rcl = oracle.search_by_projection_keyframe(ob, valid, u, v, level, desc, kfang, th3, od, ori, claimed)
        e = first(eq("keyframe n", n, rn), eq("keyframe claimed", cl, rcl))
        if e:
            return e
        th4 = int(rng.choice([3, 10]))
        n, cl, _, _ = m.SearchByProjectionSim3(gb, valid, u, v, level, desc, th4, claimed)
        rn, rcl = oracle.search_by_projection_sim3(ob, valid, u, v, level, desc, th4, claimed)
        e = first(eq("sim3proj n", n, rn), eq("sim3proj claimed", cl, rcl))
        if e:
            return e
        # window best (Fuse) with and without the chi-square gate
        radius = (np.float32(rng.choice([3.0, 7.5])) * SF[level]).astype(np.float32)
        chi2 = bool(rng.integers(0, 2))
        isg = (1.0 / SF ** 2).astype(np.float32)
        qur = (u - 8 + rng.normal(0, 1, na)).astype(np.float32)
        if rng.uniform() < 0.3:
            qur[rng.uniform(size=na) < 0.5] = -1
        bi, bd = m.window_best(gb, valid, u, v, radius, level, desc, chi2, ur_b, qur, isg)
        oi, odist = oracle.window_best(ob, valid, u, v, radius, level, desc, chi2, ur_b, qur, isg)
        e = first(eq("window idx", bi, oi), eq("window dist", bd, odist))
        if e:
            return e
        # SearchBySim3
        v21, u21, w21, l21, d21 = queries_from(rng, kb, db, (-sx, -sy), b, mf)
        q12 = (valid, u, v, (7.5 * SF[level]).astype(np.float32), level, desc)
        q21 = (v21, u21, w21, (7.5 * SF[l21]).astype(np.float32), l21, d21)
        n, m12 = m.SearchBySim3(ga, gb, q12, q21)
        rn, rm12 = oracle.search_by_sim3(oa, ob, q12, q21)
        e = first(eq("sim3 n", n, rn), eq("sim3 m12", m12, rm12))
        if e:
            return e
        # vocabulary-guided
        nodes = int(rng.choice([1, 7, 97, 1000]))
        fva, fvb = node_lists(rng, c["ida"], nodes), node_lists(rng, c["idb"], nodes)
        kfpair = bool(rng.integers(0, 2))
        v1 = (rng.uniform(size=na) < 0.8).astype(np.uint8)
        v2 = (rng.uniform(size=nb) < 0.8).astype(np.uint8) if kfpair else None
        n, m12, m21 = m.SearchByBoW(ga, fva, v1, gb, fvb, v2, keyframe_pair=kfpair)
        rn, r12, r21 = oracle.search_by_bow(oa, fva, v1, ob, fvb, v2, 49 if kfpair else 50, ratio, ori)
        e = first(eq("bow n", n, rn), eq("bow m12", m12, r12), eq("bow m21", m21, r21))
        if e:
            return e
        has1 = (rng.uniform(size=na) < 0.3).astype(np.uint8)
        has2 = (rng.uniform(size=nb) < 0.3).astype(np.uint8)
        tx, ty = (sx, sy) if (sx or sy) else (3.0, 1.0)
        F12 = np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float32).T.copy()
        ex = float(rng.choice([b[2] / 2 + 4000.0, b[2] / 2]))
        ey = float(rng.choice([b[3] / 2 - 1333.0, b[3] / 2]))
        sigma2 = (SF ** 2).astype(np.float32)
        only_stereo = bool(rng.integers(0, 2))
        n, m12 = m.SearchForTriangulation(ga, fva, has1, ur_a, gb, fvb, has2, ur_b, F12, ex, ey, sigma2, only_stereo)
        rn, r12 = oracle.search_for_triangulation(oa, fva, has1, ur_a, ob, fvb, has2, ur_b, F12, ex, ey, sigma2,
                                                  only_stereo, ori)
        e = first(eq("tri n", n, rn), eq("tri m12", m12, r12))
        if e:
            return e
        # brute force 2-NN
        if na and nb:
            bi, bd, sd = m.knn2(da, db)
            ri, rd, rs2 = oracle.knn2(da, db)
            e = first(eq("knn idx", bi, ri), eq("knn best", bd, rd), eq("knn second", sd, rs2))
            if e:
                return e
        # distinctive descriptors over random groups of the second frame's descriptors
        if nb:
            cuts = np.unique(np.concatenate([[0, nb], rng.integers(0, nb + 1, int(rng.integers(0, 40)))]))
            best, med, chosen = m.ComputeDistinctiveDescriptors(db, cuts)
            obest, omed = oracle.distinctive_descriptors(db, cuts)
            e = first(eq("distinctive best", best, obest), eq("distinctive median", med[obest >= 0], omed[obest >= 0]))
            if e:
                return e
        return None
    finally:
        m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--max-cases", type=int, default=1000000)
    a = ap.parse_args()
    t0 = time.time()
    cases = 0
    while time.time() - t0 < a.seconds and cases < a.max_cases:
        case_seed = a.seed * 1000003 + cases
        rng = np.random.default_rng(case_seed)
        c = draw_case(rng)
        err = run_case(rng, c)
        if err:
            print("MISMATCH in case seed %d (n1 %d, n2 %d, bounds %r): %s" % (case_seed, len(c["ka"]), len(c["kb"]),
                                                                              c["bounds"], err), flush=True)
            return 1
        cases += 1
        if cases % 20 == 0:
            print("%d cases, %.0f s" % (cases, time.time() - t0), flush=True)
    print("OK: %d cases, every search bit-exact; matches compared: %s" % (
        cases, ", ".join("%s %d" % (k[:-2], v) for k, v in sorted(STATS.items()))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
