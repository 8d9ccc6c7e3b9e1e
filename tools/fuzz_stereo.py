#!/usr/bin/env python3
"""Randomised parity sweep of the HIP stereo matcher against the CPU oracle: random image
sizes, pyramid shapes, disparity ranges, noise and baselines; mvuRight / mvDepth compared as
raw 32-bit patterns.

    python3 tools/fuzz_stereo.py --seed 1 --seconds 120
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as oracle  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402
from slam_amd.matcher import ORBmatcher  # noqa: E402

STATS = {"points": 0}


def draw(rng):
    scale = float(rng.choice([1.2, 1.2, 1.25, 1.5]))
    w = int(rng.integers(200, 1300))
    h = int(rng.integers(150, min(720, int(1.4 * w))))
    nl_max = 1
    while nl_max < 8 and min(w, h) / scale ** nl_max >= 72:
        nl_max += 1
    return dict(w=w, h=h, scale=scale, nlevels=int(rng.integers(1, nl_max + 1)), nf=int(rng.choice([300, 1000, 2000])),
                dt=int(rng.integers(0, 40)), db=int(rng.integers(0, 60)), noise=int(rng.integers(0, 6)),
                mb=float(rng.choice([0.1, 0.54, 1.0])), mbf=float(rng.choice([10.0, 40.0, 386.1448])),
                seed=int(rng.integers(1, 100000)))


def run_one(c):
    left, right = synth.synth_stereo_pair(c["w"], c["h"], c["seed"], c["dt"], c["db"], c["noise"])
    p = oracle.params(nfeatures=c["nf"], scale_factor=c["scale"], nlevels=c["nlevels"])
    el = ORBextractor(c["nf"], c["scale"], c["nlevels"], 20, 7, max_width=c["w"], max_height=c["h"])
    er = ORBextractor(c["nf"], c["scale"], c["nlevels"], 20, 7, max_width=c["w"], max_height=c["h"])
    m = ORBmatcher(0.6, True)
    try:
        kl, dl = el(left, None)
        kr, dr = er(right, None)
        ur, dp, n = m.ComputeStereoMatches(el, er, kl, dl, kr, dr, c["mb"], c["mbf"])
        on, our, odp = oracle.compute_stereo_matches(p, left, right, kl, dl, kr, dr, c["mb"], c["mbf"])
    finally:
        m.close()
        el.close()
        er.close()
    if n != on:
        return "%d stereo points vs oracle %d" % (n, on)
    if not np.array_equal(ur.view(np.uint32), our.view(np.uint32)):
        return "mvuRight differs at %s" % np.nonzero(ur.view(np.uint32) != our.view(np.uint32))[0][:5]
    if not np.array_equal(dp.view(np.uint32), odp.view(np.uint32)):
        return "mvDepth differs"
    STATS["points"] += int(on)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=60)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    cases = skipped = 0
    while time.time() - t0 < a.seconds:
        c = draw(rng)
        try:
            err = run_one(c)
        except Exception as e:  # noqa: BLE001
            if "LDS budget" in str(e):
                skipped += 1
                continue
            err = "exception %r" % (e,)
        if err:
            print("MISMATCH after %d cases: %s\n  config: %r" % (cases, err, c), flush=True)
            return 1
        cases += 1
        if cases % 10 == 0:
            print("%d cases, %d stereo points, %.0f s" % (cases, STATS["points"], time.time() - t0), flush=True)
    print("OK: %d cases, %d stereo points compared, all bit-exact (%d skipped)" % (cases, STATS["points"], skipped))
    return 0


if __name__ == "__main__":
    sys.exit(main())
