"""The stereo tracking front-end end to end at the KITTI shape (1241x376 pairs, 2000 features per
image), every stage on the device and every stage compared with the CPU oracle:

  Frame::Frame(left, right, ...)                       reference src/Frame.cc:118-152
    two ORBextractor calls                             src/ORBextractor.cc:1052-1114
    ComputeStereoMatches                               src/Frame.cc:603-774
    AssignFeaturesToGrid                               src/Frame.cc:302-317
  Tracking::TrackWithMotionModel, STEREO branch        src/Tracking.cc:1325-1340
    SearchByProjection(cur, last, th = 7, bMono=false) src/ORBmatcher.cc:1533-1678

The last frame's map points are its stereo points back-projected with the depth the stereo step
produced (Frame::UnprojectStereo, src/Frame.cc:799-813), the motion is a small translation along
the optical axis (forward, backward or below the baseline: the three level windows of
src/ORBmatcher.cc:1554-1556, :1593-1598), and the mvuRight gate (:1615-1621) reads the current
frame's stereo output.  Bar: bit-exact (keypoints, descriptors, mvuRight / mvDepth bit patterns,
grid, claims, match counts).  The reference holds no fixture for any of these stages.
"""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu

W, H, NF = 1241, 376, 2000
FX = FY = 718.856
CX, CY = 607.1928, 185.2157
MBF = 386.1448
MB = MBF / FX
BOUNDS = (0.0, 0.0, float(W), float(H))


def unproject(kps, depth):
    """Frame::UnprojectStereo with an identity pose (float arithmetic as in the reference)."""
    z = depth.astype(np.float32)
    x = ((kps["x"] - np.float32(CX)) * z * np.float32(1.0 / FX)).astype(np.float32)
    y = ((kps["y"] - np.float32(CY)) * z * np.float32(1.0 / FY)).astype(np.float32)
    return np.stack([x, y, z], axis=1).astype(np.float32)


@pytest.fixture(scope="module")
def stereo_frames(oracle):
    """Two consecutive stereo frames: device extraction + batched device stereo, both checked
    against the oracle; yields everything the tracking step needs, for both sides."""
    import torch
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    from slam_amd.matcher import ORBmatcher
    seq = synth.synth_stereo_sequence(W, H, 0x5EED7001, 6, 13, [(0, 0), (2, -1)], noise=2)
    lefts = np.stack([s[0] for s in seq])
    rights = np.stack([s[1] for s in seq])
    p = oracle.params(nfeatures=NF)
    el = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    er = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    m = ORBmatcher(0.9, True)
    dev = torch.device("cuda", 0)
    cap = el.max_keypoints()
    out = {}
    try:
        bufs = {}
        for side, ex, imgs in (("l", el, lefts), ("r", er, rights)):
            d_img = torch.from_numpy(imgs).to(dev)
            k = torch.zeros(2 * cap * 28, dtype=torch.uint8, device=dev)
            d = torch.zeros(2 * cap * 32, dtype=torch.uint8, device=dev)
            n = torch.zeros(2, dtype=torch.int32, device=dev)
            ex.extract_batch_device(d_img, k, d, n, cap)
            ex.synchronize()
            bufs[side] = (d_img, k.view(2, cap, 28), d.view(2, cap, 32), n.cpu().numpy())
        frames = []
        jobs = []
        d_ur = torch.full((2, cap), -7.0, dtype=torch.float32, device=dev)
        d_dp = torch.full((2, cap), -7.0, dtype=torch.float32, device=dev)
        for f in range(2):
            fr = {}
            for side, imgs in (("l", lefts), ("r", rights)):
                _, k, d, n = bufs[side]
                nk = int(n[f])
                kps = k[f, :nk].cpu().numpy().view(KP_DTYPE).reshape(-1)
                desc = d[f, :nk].cpu().numpy()
                ok, od = oracle.extract(p, imgs[f], cap=cap)
                assert kps.tobytes() == ok.tobytes(), "frame %d %s: keypoints differ" % (f, side)
                assert np.array_equal(desc, od), "frame %d %s: descriptors differ" % (f, side)
                assert nk > 1500
                fr[side] = (kps, desc)
            frames.append(fr)
            nl, nr = len(fr["l"][0]), len(fr["r"][0])
            jobs.append(dict(kps_left=bufs["l"][1][f, :nl], desc_left=bufs["l"][2][f, :nl],
                             kps_right=bufs["r"][1][f, :nr], desc_right=bufs["r"][2][f, :nr],
                             pyr_left=m.pyramid_of(el, f), pyr_right=m.pyramid_of(er, f),
                             u_right=d_ur[f, :nl], depth=d_dp[f, :nl]))
        counts = m.ComputeStereoMatchesBatch(jobs, el, MB, MBF)
        for f in range(2):
            (kl, dl), (kr, dr) = frames[f]["l"], frames[f]["r"]
            on, our, odp = oracle.compute_stereo_matches(p, lefts[f], rights[f], kl, dl, kr, dr, MB, MBF)
            ur = d_ur[f, :len(kl)].cpu().numpy()
            dp = d_dp[f, :len(kl)].cpu().numpy()
            assert counts[f] == on and on > 400, (counts[f], on)
            assert np.array_equal(ur.view(np.uint32), our.view(np.uint32)), "frame %d: mvuRight differs" % f
            assert np.array_equal(dp.view(np.uint32), odp.view(np.uint32)), "frame %d: mvDepth differs" % f
            assert (d_ur[f, len(kl):] == -7.0).all()          # nothing written past the frame's keypoints
            # the single-pair entry point on the same device arrays gives the same bits
            ur1, dp1, n1 = m.ComputeStereoMatches(el, er, kl, dl, kr, dr, MB, MBF, frame_left=f, frame_right=f)
            assert n1 == on and np.array_equal(ur1.view(np.uint32), our.view(np.uint32))
            assert np.array_equal(dp1.view(np.uint32), odp.view(np.uint32))
            frames[f]["ur"], frames[f]["depth"] = ur, dp
            frames[f]["d_ur"] = d_ur[f, :len(kl)]
        out = dict(frames=frames, bufs=bufs, sf=el.GetScaleFactors(), keep=(d_ur, d_dp))
        yield out
    finally:
        m.close()
        el.close()
        er.close()


@pytest.mark.parametrize("tz,want_direction,prior", [(-0.65, 1, False), (0.65, 2, False), (0.1, 0, True)])
def test_track_with_motion_model_stereo(oracle, stereo_frames, tz, want_direction, prior):
    import torch
    from slam_amd.matcher import FrameView, ORBmatcher
    fr_last, fr_cur = stereo_frames["frames"]
    sf = stereo_frames["sf"]
    (ka, da), (kb, db) = fr_last["l"], fr_cur["l"]
    nl, nc = len(ka), len(kb)
    rng = np.random.RandomState(11 + want_direction)
    depth = fr_last["depth"]
    # mvpMapPoints of the last frame: its stereo points (UpdateLastFrame, src/Tracking.cc:1245-1307); a
    # tenth of them flagged as outliers, a fifth temporal points without observations
    valid = ((depth > 0) & (rng.uniform(size=nl) < 0.9)).astype(np.uint8)
    has_obs = (rng.uniform(size=nl) < 0.8).astype(np.uint8)
    x3d = unproject(ka, np.where(depth > 0, depth, np.float32(1.0)))
    Tlw = np.eye(4, dtype=np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    # the camera moved by (2, -1) pixels' worth sideways (depth 30..64 m) and tz along the optical axis
    Tcw[:3, 3] = (-0.12, 0.06, tz)
    u, v, invz, direction = oracle.project_last_frame(Tcw, Tlw, x3d, FX, FY, CX, CY, MB, False)
    assert direction == want_direction
    claimed = np.full(nc, -1, np.int32)
    cobs = np.zeros(nc, np.uint8)
    if prior:
        pick = rng.uniform(size=nc) < 0.25
        claimed[pick] = 3
        cobs[pick] = (rng.uniform(size=pick.sum()) < 0.5).astype(np.uint8)
    ur_cur = fr_cur["ur"]
    ocur = oracle.FrameData(kb, db, BOUNDS, sf)
    olast = oracle.FrameData(ka, da, BOUNDS, sf)
    rn, rcl, rco, rqb = oracle.search_by_projection_frames(ocur, olast, valid, u, v, invz, has_obs, ur_cur, MBF, 7.0,
                                                           direction, True, claimed, cobs, return_query_best=True)
    assert rn > 150, rn
    if want_direction:
        # the gate does decide something on these pairs: without mvuRight the result differs
        rn2, rcl2, _ = oracle.search_by_projection_frames(ocur, olast, valid, u, v, invz, has_obs,
                                                          np.full(nc, -1.0, np.float32), MBF, 7.0, direction, True,
                                                          claimed, cobs)
        assert not np.array_equal(rcl, rcl2)

    m = ORBmatcher(0.9, True)
    try:
        cur = FrameView(kb, db, BOUNDS, sf, u_right=ur_cur)
        last = FrameView(ka, da, BOUNDS, sf)
        # AssignFeaturesToGrid of the current frame
        start, items = m.AssignFeaturesToGrid(cur)
        ostart, oitems = oracle.grid_build(ocur)
        assert np.array_equal(start, ostart) and np.array_equal(items, oitems)
        # host arrays in and out (the shape of the reference's call)
        n, cl, co, qb = m.SearchByProjection(cur, last, 7.0, valid, u, v, invz, has_obs, MBF, direction,
                                             claimed=claimed, claimed_obs=cobs, return_query_best=True)
        assert n == rn and np.array_equal(cl, rcl) and np.array_equal(co, rco) and np.array_equal(qb, rqb)
        # device-resident: keypoints, descriptors and mvuRight exactly where extraction and stereo left them
        dev = torch.device("cuda", 0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        bl = stereo_frames["bufs"]["l"]
        d_cl, d_co = t(claimed), t(cobs)
        d_qb = torch.full((nl,), -5, dtype=torch.int32, device=dev)
        pair = dict(cur_kps=bl[1][1, :nc], cur_desc=bl[2][1, :nc], last_kps=bl[1][0, :nl], last_desc=bl[2][0, :nl],
                    valid=t(valid), u=t(u), v=t(v), invz=t(invz), has_obs=t(has_obs), cur_u_right=fr_cur["d_ur"],
                    claimed=d_cl, claimed_obs=d_co, query_best=d_qb, direction=direction, bounds=BOUNDS)
        counts = m.SearchByProjectionBatch([pair], sf, MBF, 7.0)
        torch.cuda.synchronize()
        assert counts == [rn]
        assert np.array_equal(d_cl.cpu().numpy(), rcl) and np.array_equal(d_co.cpu().numpy(), rco)
        assert np.array_equal(d_qb.cpu().numpy(), rqb)
    finally:
        m.close()
