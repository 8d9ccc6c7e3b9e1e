#!/usr/bin/env python3
"""Headline benchmark: ORB frames/s (640x480, 1000 kp, 8 levels) on N MI355X.

One step = one pass of the extractor over this rank's batch of synthetic frames
(4096 per GPU, resident in HBM when the timed region starts, outputs left in
HBM).  Frames shard across ranks with no collective (weak scaling).  The JSON
line also carries: per-stage HIP-event timings and the roofline of the
dominant kernel, a CPU baseline (the oracle, on a bounded sample), a parity
check of that sample, and Hamming pairs/s of the brute-force 2-NN search.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

W, H, NFEAT, NLEVELS = 640, 480, 1000, 8
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# Algorithmic bytes per 640x480 frame of each stage (level sizes 640x480 ... 179x134):
LEVEL_PX = [640 * 480, 533 * 400, 444 * 333, 370 * 278, 309 * 231, 257 * 193, 214 * 161, 179 * 134]
ALG_BYTES = {
    "pyramid": sum(LEVEL_PX[:-1]) + sum(LEVEL_PX[1:]),  # read levels 0..6, write levels 1..7
    "fast": sum(LEVEL_PX),                              # every level read once
    "blur": 2 * sum(LEVEL_PX),                          # read + write every level
    "describe": NFEAT * (749 + 512 + 60),               # patch + test gathers + outputs per keypoint
    "quadtree": None,                                   # latency-bound list logic; bytes are negligible
}
KERNEL_OF = {"pyramid": "k_pyramid_frame", "fast": "k_fast", "quadtree": "k_quadtree",
             "blur": "k_blur_rows", "describe": "k_describe"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--chunk", type=int, default=512, help="frames per internal pass")
    ap.add_argument("--cpu-frames", type=int, default=4096, help="sample size of the CPU baseline")
    ap.add_argument("--schedule", choices=["both", "default", "serial"], default="both",
                    help="default: only the timed region (the library's default schedule); both: also a serialised run; "
                         "serial: the timed region itself runs one pass at a time on one stream (for a kernel "
                         "trace in which every duration is the kernel's own)")
    ap.add_argument("--roofline-kernel", default="auto", choices=["auto", "pyramid", "fast", "quadtree", "blur", "describe"],
                    help="stage reported in `roofline` (auto: largest share of the serialised time)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other frame shapes (KITTI stereo shape, 4K)")
    ap.add_argument("--knn", type=int, default=1 << 20, help="descriptors per side of the 2-NN search")
    return ap.parse_args()


def host_cores():
    """(cores this process may run on, threads used by the CPU legs).  The legs use every core of the
    process's affinity mask, bounded by ORBX_BENCH_CPU_THREADS (default 256) so that a very wide host does
    not spawn thousands of threads for a bounded sample."""
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    return avail, max(1, min(avail, int(os.environ.get("ORBX_BENCH_CPU_THREADS", "256"))))


def kitti_pipeline(dev, stream, barrier, synth, ORBextractor, ORBmatcher, npairs=1024, chunk=256, cpu=True):
    """BASELINE config 3 as a device-resident pipeline: 1241x376 stereo frames, 2000 features per image.
    Per frame: extraction of both images, ComputeStereoMatches; per consecutive frame pair: frame-to-frame
    SearchByProjection (th 7, bMono false) with the mvuRight gate fed by the stereo step.  The frames are
    cut two per synthetic scene (the second displaced by (2, -1) pixels), so npairs stereo frames give
    npairs projection searches (each scene's pair in both orders)."""
    import torch
    w2, h2, nf2 = 1241, 376, 2000
    FX, CX, CY, MBF = 718.856, 607.1928, 185.2157, 386.1448
    MB = MBF / FX
    out = {}
    el = ORBextractor(nf2, 1.2, NLEVELS, 20, 7, max_width=w2, max_height=h2, max_batch=chunk, device=dev.index)
    er = ORBextractor(nf2, 1.2, NLEVELS, 20, 7, max_width=w2, max_height=h2, max_batch=chunk, device=dev.index)
    mm = ORBmatcher(0.9, True, device=dev.index)
    for o in (el, er, mm):
        o.set_stream(stream.cuda_stream)
    cap2 = el.max_keypoints()
    pitch = (w2 + 3) // 4 * 4
    margin, vm, dt, db = 80, 16, 6, 13
    bl = torch.zeros((npairs, h2, pitch), dtype=torch.uint8, device=dev)
    br = torch.zeros((npairs, h2, pitch), dtype=torch.uint8, device=dev)
    for s0 in range(0, npairs // 2, 8):
        n = min(8, npairs // 2 - s0)
        scene = synth.synth_batch_torch(n, w2 + 2 * margin, h2 + 2 * vm, 0x5EED7001 + s0, device=dev)
        for k, (dx, dy) in enumerate(((0, 0), (2, -1))):
            x0, y0 = margin + dx, vm + dy
            split = vm + h2 // 2 - y0
            f = slice(2 * s0 + k, 2 * (s0 + n) + k, 2)
            bl[f, :, :w2] = scene[:, y0:y0 + h2, x0:x0 + w2]
            br[f, :split, :w2] = scene[:, y0:y0 + split, x0 + dt:x0 + dt + w2]
            br[f, split:, :w2] = scene[:, y0 + split:y0 + h2, x0 + db:x0 + db + w2]
        del scene
    kL = torch.zeros(npairs * cap2 * 28, dtype=torch.uint8, device=dev)
    dL = torch.zeros(npairs * cap2 * 32, dtype=torch.uint8, device=dev)
    nL = torch.zeros(npairs, dtype=torch.int32, device=dev)
    kR, dR, nR = torch.zeros_like(kL), torch.zeros_like(dL), torch.zeros_like(nL)
    uR = torch.full((npairs, cap2), -1.0, dtype=torch.float32, device=dev)
    dp = torch.full((npairs, cap2), -1.0, dtype=torch.float32, device=dev)
    kLv, dLv, kRv, dRv = kL.view(npairs, cap2, 28), dL.view(npairs, cap2, 32), kR.view(npairs, cap2, 28), dR.view(npairs, cap2, 32)

    def extract_chunk(c0, c1):
        el.extract_batch_device(bl[c0:c1, :, :w2], kLv[c0:c1].reshape(-1), dLv[c0:c1].reshape(-1), nL[c0:c1], cap2)
        er.extract_batch_device(br[c0:c1, :, :w2], kRv[c0:c1].reshape(-1), dRv[c0:c1].reshape(-1), nR[c0:c1], cap2)

    # warm-up pass: also builds the stereo job table of every chunk (the pyramids of a chunk are resident
    # after its extraction; later passes reuse the same buffers, so the tables stay valid)
    tables = []
    for c0 in range(0, npairs, chunk):
        c1 = min(c0 + chunk, npairs)
        extract_chunk(c0, c1)
        el.synchronize()
        er.synchronize()
        nl, nr = nL[c0:c1].cpu().numpy(), nR[c0:c1].cpu().numpy()
        jobs = [dict(kps_left=kLv[c0 + i, :int(nl[i])], desc_left=dLv[c0 + i, :int(nl[i])],
                     kps_right=kRv[c0 + i, :int(nr[i])], desc_right=dRv[c0 + i, :int(nr[i])],
                     pyr_left=mm.pyramid_of(el, i), pyr_right=mm.pyramid_of(er, i),
                     u_right=uR[c0 + i, :int(nl[i])], depth=dp[c0 + i, :int(nl[i])]) for i in range(c1 - c0)]
        tables.append(mm.stereo_job_table(jobs))
        mm.ComputeStereoMatchesBatch(tables[-1], el, MB, MBF)
    nl_all = nL.cpu().numpy()
    # projection inputs on the device from the stereo depth: the last frame's stereo points moved by a small
    # translation (float32 torch arithmetic; an input of the search, the same for the CPU leg)
    kf = kLv.contiguous().view(torch.float32).view(npairs, cap2, 7)
    z = torch.where(dp > 0, dp, torch.ones_like(dp))
    X = (kf[:, :, 0] - CX) * z / FX
    Y = (kf[:, :, 1] - CY) * z / FX
    sign = torch.ones((npairs, 1), device=dev)
    sign[1::2] = -1.0  # frame 2k -> 2k+1 sees the scene move by (-2, +1) pixels' worth; 2k+1 -> 2k the opposite
    tz = 0.65
    zc = z + sign * tz
    uu = (FX * (X - sign * 0.12) / zc + CX).contiguous()
    vv = (FX * (Y + sign * 0.06) / zc + CY).contiguous()
    invz = (1.0 / zc).contiguous()
    valid = (dp > 0).to(torch.uint8).contiguous()
    ones8 = torch.ones((npairs, cap2), dtype=torch.uint8, device=dev)
    claimed = torch.full((npairs, cap2), -1, dtype=torch.int32, device=dev)
    cobs = torch.zeros((npairs, cap2), dtype=torch.uint8, device=dev)
    pairs = []
    for i in range(npairs):
        j = i ^ 1  # the other frame of the scene is the current frame
        n, nc = int(nl_all[i]), int(nl_all[j])
        pairs.append(dict(cur_kps=kLv[j, :nc], cur_desc=dLv[j, :nc], last_kps=kLv[i, :n], last_desc=dLv[i, :n],
                          valid=valid[i, :n], u=uu[i, :n], v=vv[i, :n], invz=invz[i, :n], has_obs=ones8[i, :n],
                          cur_u_right=uR[j, :nc], claimed=claimed[j, :nc], claimed_obs=cobs[j, :nc],
                          direction=2 if i % 2 == 0 else 1, bounds=(0.0, 0.0, float(w2), float(h2))))
    sfac = el.GetScaleFactors()
    ptable = mm.projection_pair_table(pairs)
    counts = mm.SearchByProjectionBatch(ptable, sfac, MBF, 7.0)
    barrier()
    # ---- timed: the whole pipeline, with the stereo and projection calls timed on their own inside it
    reps = 2
    t_st = t_pr = 0.0
    n_st = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        for ci, c0 in enumerate(range(0, npairs, chunk)):
            c1 = min(c0 + chunk, npairs)
            extract_chunk(c0, c1)
            el.synchronize()
            er.synchronize()
            t1 = time.perf_counter()
            cnt = mm.ComputeStereoMatchesBatch(tables[ci], el, MB, MBF)
            t_st += time.perf_counter() - t1
            n_st += sum(cnt)
        claimed.fill_(-1)
        cobs.zero_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        counts = mm.SearchByProjectionBatch(ptable, sfac, MBF, 7.0)
        torch.cuda.synchronize()
        t_pr += time.perf_counter() - t1
    barrier()
    t_all = time.perf_counter() - t0
    # ---- extraction alone, both images of every frame, default schedule (all chunks in one call per side)
    t1 = time.perf_counter()
    for _ in range(reps):
        el.extract_batch_device(bl[:, :, :w2], kL, dL, nL, cap2)
        er.extract_batch_device(br[:, :, :w2], kR, dR, nR, cap2)
    barrier()
    t_ex = time.perf_counter() - t1
    nq = float(valid.sum().item())
    out = {"stereo_frames": npairs, "frames_per_pass": chunk, "features": nf2,
           "extraction_frames_per_s": round(2 * npairs * reps / t_ex, 1),
           "stereo_points_per_s": round(float(nl_all.sum()) * reps / t_st, 1),
           "stereo_frames_per_s": round(npairs * reps / t_st, 1),
           "stereo_points_found_per_frame": round(n_st / (reps * npairs), 1),
           "projection_queries_per_s": round(nq * reps / t_pr, 1),
           "projection_pairs_per_s": round(npairs * reps / t_pr, 1),
           "projection_matches_per_pair": round(float(np.mean(counts)), 1),
           "pipeline_stereo_frames_per_s": round(npairs * reps / t_all, 1),
           "mean_keypoints_left": round(float(nl_all.mean()), 1), "th": 7.0, "bMono": False,
           "note": "1241x376 rows padded to 1244 bytes; extraction of both images + orbm_compute_stereo_matches_batch "
                   "per pass of frames, orbm_search_by_projection_batch over all frame pairs (mvuRight gate fed by the "
                   "stereo output), everything resident in HBM; the stereo and projection rates are host wall time of "
                   "their synchronous calls inside the pipeline loop"}
    if cpu:
        # CPU leg on a bounded sample: the oracle's ComputeStereoMatches and SearchByProjection over the first
        # frames, single thread (the reference runs them on the tracking thread), with parity of that sample
        from oracle import pyoracle as O
        from slam_amd._lib import KP_DTYPE
        S = 8
        p = O.params(nfeatures=nf2)
        imgs_l = bl[:S, :, :w2].cpu().numpy()
        imgs_r = br[:S, :, :w2].cpu().numpy()
        n_r = nR.cpu().numpy()
        bad = 0
        t_cs = t_cp = 0.0
        q_cpu = 0
        fr = []
        for i in range(S):
            kl = kLv[i, :int(nl_all[i])].cpu().numpy().view(KP_DTYPE).reshape(-1)
            dl = dLv[i, :int(nl_all[i])].cpu().numpy()
            kr = kRv[i, :int(n_r[i])].cpu().numpy().view(KP_DTYPE).reshape(-1)
            dr = dRv[i, :int(n_r[i])].cpu().numpy()
            t1 = time.perf_counter()
            on, our, odp = O.compute_stereo_matches(p, imgs_l[i], imgs_r[i], kl, dl, kr, dr, MB, MBF)
            t_cs += time.perf_counter() - t1
            bad += int((our.view(np.uint32) != uR[i, :len(kl)].cpu().numpy().view(np.uint32)).sum())
            fr.append((kl, dl, our))
        cl_dev = claimed.cpu().numpy()
        for i in range(S):
            j = i ^ 1
            (ka, da, _), (kb, dbb, urb) = fr[i], fr[j]
            oc = O.FrameData(kb, dbb, (0.0, 0.0, float(w2), float(h2)), sfac)
            ol = O.FrameData(ka, da, (0.0, 0.0, float(w2), float(h2)), sfac)
            n = len(ka)
            a = [x[i, :n].cpu().numpy() for x in (valid, uu, vv, invz, ones8)]
            t1 = time.perf_counter()
            rn, rcl, rco = O.search_by_projection_frames(oc, ol, a[0], a[1], a[2], a[3], a[4], urb, MBF, 7.0,
                                                         2 if i % 2 == 0 else 1, True)
            t_cp += time.perf_counter() - t1
            q_cpu += int(a[0].sum())
            bad += int((rcl != cl_dev[j, :len(kb)]).sum())
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "sample": "first %d stereo frames, oracle, single thread" % S,
                               "stereo_points_per_s": round(sum(len(f[0]) for f in fr) / t_cs, 1),
                               "projection_queries_per_s": round(q_cpu / t_cp, 1), "unit": "points/s, queries/s",
                               "mismatches_vs_device": bad}
    mm.close()
    el.close()
    er.close()
    return out


def main():
    if os.environ.get("ORBX_BENCH_WATCHDOG"):
        # diagnostic: after that many seconds every thread's Python stack goes to stderr (the run continues)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ORBX_BENCH_WATCHDOG"]), repeat=False)
    args = parse()
    from slam_amd import dist as sdist
    from slam_amd import synth
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    from slam_amd.matcher import ORBmatcher

    rank, local_rank, world = sdist.init_process_group()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # more ranks than GPUs only happens in rehearsals
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # scalars exchanged between ranks live on the GPU under RCCL and on the host under gloo (a rehearsal of
    # several ranks on one GPU): gloo's handling of device tensors stalled now and then in that setting
    red_dev = dev if (world > 1 and dist.get_backend() == "nccl") else torch.device("cpu")

    B = args.frames
    ex = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, max_width=W, max_height=H,
                      max_batch=min(args.chunk, B), device=local_rank)
    # One stream for torch's work and the library's: producers (synthetic frames, fills), the
    # library's kernels and consumers are then ordered by the stream itself.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex.set_stream(stream.cuda_stream)
    cap = ex.max_keypoints()

    # synthetic frames, generated on the device (bit-identical to the numpy generator)
    seed0 = 0x5EED0001 + rank * B
    frames = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    for f0 in range(0, B, 128):
        n = min(128, B - f0)
        frames[f0:f0 + n] = synth.synth_batch_torch(n, W, H, seed0 + f0, device=dev)
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)

    def step():
        ex.extract_batch_device(frames, d_kps, d_desc, d_n, cap)

    def timed(overlap):
        """W untimed steps, then K timed steps between barriers; returns (seconds, stage profile)."""
        ex.set_overlap(overlap)
        for _ in range(max(args.warmup, 1)):
            step()
        barrier()
        ex.profile_enable(True)
        ex.profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        el = time.perf_counter() - t0
        ex.synchronize()  # raises if a kernel of the device-space passes reported a capacity error
        ex.profile_enable(False)
        prof = ex.profile_read()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, prof

    passes = args.steps * ((B + ex.max_batch - 1) // ex.max_batch)
    walk_note = None
    frames_per_launch = B * args.steps / passes

    def stage_table(prof):
        stages = {}
        # The library's default schedule makes one walk per pyramid level that writes the blurred level AND the
        # next level (k_level_walk): the blur stage then has no launches of its own and the "pyramid" stage
        # stands for both, with the bytes of both (every level read once, levels 1.. and all blurred levels written).
        walk = prof.get("blur", (0.0, 0))[1] == 0
        ALG_BYTES["pyramid"] = (3 * sum(LEVEL_PX) - LEVEL_PX[0]) if walk else (sum(LEVEL_PX[:-1]) + sum(LEVEL_PX[1:]))
        KERNEL_OF["pyramid"] = "k_level_walk" if walk else "k_pyramid_frame"
        for name, (ms, launches) in prof.items():
            if launches == 0:
                continue
            if name == "pyramid" and walk:
                launches = launches  # one timed interval per pass covers the eight walks
            per_launch_ms = ms / max(launches, 1)
            ent = {"ms_total": round(ms, 3), "launches": launches, "ms_per_launch": round(per_launch_ms, 4),
                   "us_per_frame": round(1000.0 * ms / (B * args.steps), 4)}
            if ALG_BYTES[name]:
                ent["achieved_GBs"] = round(ALG_BYTES[name] * frames_per_launch / (per_launch_ms * 1e-3) / 1e9, 1)
            stages[name] = ent
        return stages

    # The timed region is the library's default schedule: up to three passes of 512 frames in
    # flight on their own streams, the blur of a pass beside its FAST + quadtree.  Stage times
    # are HIP events around each stage on the stream it runs on, so under this schedule they
    # include the time a kernel shares the GPU with the other passes.
    elapsed, prof = timed(args.schedule != "serial")
    total_frames = world * B * args.steps
    value = total_frames / elapsed
    stages = stage_table(prof)
    # ---- the same steps with every kernel serialised on one stream: each stage's duration is its own
    if args.schedule == "both":
        el2, prof2 = timed(False)
        st2 = stage_table(prof2)
        ex.set_overlap(True)
    else:
        el2, st2 = None, stages
    # The dominant kernel is the one with the largest share of the serialised time (under the
    # concurrent schedule a stage's wall time also counts what runs beside it); its launch
    # duration for the roofline is the one measured in the timed region above.
    dominant = args.roofline_kernel if args.roofline_kernel != "auto" else max(st2, key=lambda k: st2[k]["ms_total"])
    # HBM traffic and vector-instruction counts come from separate rocprofv3 counter passes
    # (tools/collect_profiles.sh, tools/collect_sq.sh) whose summaries are committed under profiles/
    # together with a hash of the kernel sources they were taken with: a summary of other sources is
    # not quoted.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import kernel_source_hash
    src_hash = kernel_source_hash()
    traffic, traffic_note, launches_per_pass = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("source_hash") != src_hash:
                traffic_note = "profiles/pmc_traffic.json was taken with other kernel sources (stale): not quoted"
            elif dominant in tj and tj[dominant].get("frames_per_launch"):
                # the counter passes run one pass of 512 frames; bytes scale with the frames of a pass
                traffic = int(tj[dominant]["hbm_bytes_per_launch"] * frames_per_launch / tj[dominant]["frames_per_launch"])
                launches_per_pass = tj[dominant].get("launches_per_pass", 1)
                if tj[dominant]["frames_per_launch"] != frames_per_launch:
                    traffic_note = ("counter passes ran %d frames per pass; scaled to the %d of this run"
                                    % (tj[dominant]["frames_per_launch"], int(frames_per_launch)))
        except Exception:
            traffic = None
    valu = None
    spath = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(spath):
        try:
            import csv
            sj = json.load(open(spath))
            if sj.get("source_hash") == src_hash:
                for r in csv.DictReader(open(os.path.join(ROOT, "profiles", sj["file"]))):
                    if r["kernel"] == KERNEL_OF[dominant].split(" ")[0]:
                        valu = float(r["SQ_INSTS_VALU"]) * frames_per_launch / 512.0   # the counter pass ran 512 frames
        except Exception:
            valu = None
    alg = ALG_BYTES[dominant]
    ach = stages[dominant].get("achieved_GBs", 0.0)
    roofline = {"kernel": KERNEL_OF[dominant], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg * frames_per_launch) if alg else None,
                "avg_launch_ms": stages[dominant]["ms_per_launch"],
                "avg_launch_ms_serialized": st2[dominant]["ms_per_launch"],
                "achieved_serialized": st2[dominant].get("achieved_GBs"),
                "frac_serialized": round((st2[dominant].get("achieved_GBs") or 0.0) / HBM_PEAK_GBS, 5),
                "time_share": round(st2[dominant]["ms_total"] / sum(s["ms_total"] for s in st2.values()), 3),
                "kernel_source_hash": src_hash}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    if launches_per_pass and launches_per_pass != 1:
        # a "launch" of this stage is the set of launches that covers one pass of frames (k_fast runs once per
        # group of pyramid levels): durations, bytes and counters are sums over that set
        roofline["launches_per_pass"] = launches_per_pass
    if valu:
        per_s = valu / (st2[dominant]["ms_per_launch"] * 1e-3)
        roofline["valu_issue"] = {"wave_instr_per_launch": int(valu), "wave_instr_per_s_serialized": round(per_s, 1),
                                  "issue_rate_at_4_cycles": round(1024 * 2.4e9 / 4.0, 1),
                                  "frac_of_issue_rate": round(per_s / (1024 * 2.4e9 / 4.0), 3),
                                  "note": "1024 SIMDs at 2.4 GHz; measured 2.2-2.8 cycles per wave64 instruction for "
                                          "add/sub/and/or/xor/lshr/bitop3/fma, 4.1-4.5 for the rest (profiles/r2_valu_probe.txt)"}

    out = {
        "metric": "ORB frames/s (640x480, 1000kp, 8 lvl)", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "synthetic 640x480 batch of %d frames per GPU, ORB extraction "
                               "(1000 features, 8 levels, scale 1.2, FAST 20/7), frames and outputs "
                               "resident in HBM" % B,
                   "frames_per_gpu": B, "frames_per_pass": ex.max_batch,
                   "schedule": ("one pass at a time on one stream (--schedule serial)" if args.schedule == "serial"
                                else "up to 3 passes in flight, blur beside FAST + quadtree"),
                   "parallelism": "frames sharded, no collective"},
        "roofline": roofline, "stages": stages,
    }

    if el2 is not None:
        out["serialized"] = {"value": round(total_frames / el2, 1), "unit": "frames/s",
                             "note": "one pass at a time on one stream; same work, same outputs",
                             "stages": {k: {"us_per_frame": v["us_per_frame"], "ms_per_launch": v["ms_per_launch"],
                                            **({"achieved_GBs": v["achieved_GBs"]} if "achieved_GBs" in v else {})}
                                        for k, v in st2.items()}}

    # ---- CPU baseline + parity of the sample (rank 0, single-GPU run only)
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle as O
        S = min(args.cpu_frames, B)
        host = frames[:S].cpu().numpy()
        avail, threads = host_cores()
        p = O.params(NFEAT, 1.2, NLEVELS, 20, 7)
        t1 = time.perf_counter()
        rk, rd, rn = O.extract_batch(p, host, cap=cap, threads=threads)
        cpu_mt = time.perf_counter() - t1
        s1 = min(256, S)
        t1 = time.perf_counter()
        O.extract_batch(p, host[:s1], cap=cap, threads=1)
        cpu_1t = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(S / cpu_mt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
                               "cores_available": avail, "threads_used": threads,
                               "hardware_concurrency": os.cpu_count(),
                               "sample": "first %d of the benchmark's frames, oracle (C++ restatement of the "
                                         "reference path, not OpenCV), frame-parallel std::thread on every core of "
                                         "the process's affinity mask" % S,
                               "single_thread_frames_per_s": round(s1 / cpu_1t, 2), "single_thread_sample_frames": s1}
        if threads != 16 and avail >= 16:
            t1 = time.perf_counter()
            O.extract_batch(p, host[:min(S, 512)], cap=cap, threads=16)
            out["cpu_baseline"]["frames_per_s_16_threads"] = round(min(S, 512) / (time.perf_counter() - t1), 2)
        s2 = min(64, S)
        st_cpu = O.extract_stages(p, host[:s2])
        tot_cpu = sum(st_cpu.values()) or 1.0
        out["cpu_baseline"]["single_thread_stage_share"] = {k: round(v / tot_cpu, 3) for k, v in st_cpu.items()}
        out["cpu_baseline"]["single_thread_ms_per_frame"] = round(tot_cpu / s2 * 1e3, 2)
        out["cpu_baseline"]["stage_share_sample_frames"] = s2
        n = d_n[:S].cpu().numpy()
        kps = d_kps[:S * cap * 28].cpu().numpy().view(KP_DTYPE).reshape(S, cap)
        desc = d_desc[:S * cap * 32].cpu().numpy().reshape(S, cap, 32)
        bad_kp = bad_desc = tot = 0
        for f in range(S):
            if n[f] != rn[f]:
                bad_kp += max(n[f], rn[f])
                tot += max(n[f], rn[f])
                continue
            tot += int(n[f])
            same = np.ones(n[f], bool)
            for fld in KP_DTYPE.names:
                same &= kps[f, :n[f]][fld] == rk[f, :n[f]][fld]
            bad_kp += int((~same).sum())
            bad_desc += int((desc[f, :n[f]] != rd[f, :n[f]]).any(axis=1).sum())
        out["parity"] = {"frames": S, "keypoints": tot, "mismatched_keypoints": bad_kp,
                         "mismatched_descriptors": bad_desc}

    # ---- the same extraction from HOST memory (the shape of the reference's call: the caller owns the images
    # and wants keypoints back in host memory): pageable numpy arrays through the C ABI; never the headline
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            from slam_amd import _lib as L_
            lib = L_.load()
            HB = min(2048, B)
            h_img = frames[:HB].cpu().numpy()
            h_k = np.zeros((HB, cap), KP_DTYPE)
            h_d = np.zeros((HB, cap, 32), np.uint8)
            h_n = np.zeros(HB, np.int32)

            def host_call():
                L_.check(lib.orbx_extract_batch(ex._h, h_img.ctypes.data, HB, W, H, W, W * H, h_k.ctypes.data, h_d.ctypes.data,
                                                cap, h_n.ctypes.data, 0))
            host_call()          # also touches every page of the output arrays
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                host_call()
            dt = time.perf_counter() - t1
            dn = d_n[:HB].cpu().numpy()
            dk = d_kps[:HB * cap * 28].cpu().numpy().view(KP_DTYPE).reshape(HB, cap)
            dd = d_desc[:HB * cap * 32].cpu().numpy().reshape(HB, cap, 32)
            same = bool(np.array_equal(dn, h_n)) and all(
                dk[f, :dn[f]].tobytes() == h_k[f, :dn[f]].tobytes() and np.array_equal(dd[f, :dn[f]], h_d[f, :dn[f]]) for f in range(HB))
            out["host_path"] = {"frames_per_s": round(HB * reps / dt, 1), "frames_per_call": HB, "frames_per_pass": min(ex.max_batch, 128),
                                "bytes_in_per_frame": W * H, "bytes_out_per_frame": cap * 60 + 4,
                                "identical_to_device_resident_outputs": same,
                                "note": "pageable host images in, host keypoints / descriptors out (orbx_extract_batch, ORBX_MEM_HOST): "
                                        "worker threads stage passes through pinned memory while uploads, kernels and downloads of "
                                        "neighbouring passes overlap; PCIe-inclusive, not the headline"}
            del h_img, h_k, h_d, dk, dd
        except Exception as e:  # noqa: BLE001
            out["host_path"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- Hamming pairs/s: brute-force 2-NN, train rows sharded over ranks.  A failure here
    # must not lose the headline line: on more than one GPU this leg has a collective exchange, so
    # the JSON line goes out first and the Hamming result follows on stderr (and in
    # gpurun_out/hamming_N.json when that directory exists).
    headline_printed = False
    if world > 1 and rank == 0:
        print(json.dumps(out), flush=True)
        headline_printed = True
    guard = None
    if world > 1 and not args.no_hamming:
        # The headline line is out.  The leg below has a collective exchange: should it ever stall (a rank
        # lost, a fabric problem), every rank ends its process after ORBX_BENCH_LEG_TIMEOUT seconds
        # (default 300) with status 3 (a failure: the headline stands, the leg is reported as timed out)
        # instead of holding the job until something kills it.
        import threading

        def _give_up():
            print("hamming: " + json.dumps({"error": "timeout"}), file=sys.stderr, flush=True)
            print("bench: the multi-rank Hamming leg exceeded its time limit; ending without it", file=sys.stderr, flush=True)
            os._exit(3)

        guard = threading.Timer(float(os.environ.get("ORBX_BENCH_LEG_TIMEOUT", "300")), _give_up)
        guard.daemon = True
        guard.start()
    if not args.no_hamming:
        try:
            m = ORBmatcher(0.6, True, device=local_rank)
            m.set_stream(stream.cuda_stream)
            NQ = NT = args.knn
            nq_l = NQ // world
            nt0, nt1 = sdist.shard_range(NT, rank, world)
            # BASELINE config 4 data (seed 0xD35C): random train rows, half of the queries planted
            # near-duplicates (<= 30 flipped bits), exact ties at higher indices; generated once on
            # the host, every rank takes its slice of queries and its shard of train rows
            q_h, t_h, _ = synth.synth_knn_workload(NQ, NT, seed=0xD35C)
            q_l = torch.from_numpy(q_h[rank * nq_l:(rank + 1) * nq_l]).to(dev)
            t_l = torch.from_numpy(t_h[nt0:nt1]).to(dev)
            accept = torch.empty(nq_l, dtype=torch.uint8, device=dev)

            def local_knn2(q_all, t_loc, off):
                bi = torch.empty(q_all.shape[0], dtype=torch.int32, device=dev)
                bd = torch.empty_like(bi)
                sd = torch.empty_like(bi)
                m.knn2_device(q_all, t_loc, bi, bd, sd, off)
                return bi, bd, sd

            def merge(a, b):
                m.knn2_merge_device(a[0], a[1], a[2], b[0], b[1], b[2])
                return a

            def search(ql, tl):
                """One whole job: exchange, local scans, ordered merge, ratio test."""
                r = sdist.knn2_sharded(ql, tl, nt0, local_knn2, merge)
                m.ratio_test_device(r[1], r[2], accept)
                return r

            def time_search(ql, tl, reps=2):
                search(ql, tl)  # warm-up
                barrier()
                t1 = time.perf_counter()
                for _ in range(reps):
                    search(ql, tl)
                barrier()
                dt = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                return float(nq_l * world) * float(NT) * reps / dt

            rate = time_search(q_l, t_l)
            accepted = int(accept.sum().item())
            if world > 1:
                ta = torch.tensor([accepted], dtype=torch.int64, device=red_dev)
                dist.all_reduce(ta)
                accepted = int(ta.item())
            # worst case for the scan's skip of the top-2 update: distances fall along the train rows
            qa_h, ta_h = synth.synth_knn_adversarial(NQ, NT, seed=0xD35C)
            rate_adv = time_search(torch.from_numpy(qa_h[rank * nq_l:(rank + 1) * nq_l]).to(dev),
                                   torch.from_numpy(ta_h[nt0:nt1]).to(dev))
            # the same job on the vector ALU (xor + popcount), for the comparison the choice of path rests on
            m.debug_knn_path(1)
            rate_valu = time_search(q_l, t_l, reps=1)
            m.debug_knn_path(0)
            # Matrix-core roofline of the scan: a pair of 256-bit descriptors is a 256-long contraction =
            # 512 operations in the block-scaled fp4 form (dense peak 10 PFLOP/s per GPU).  The rate the chip
            # sustains on this operand mix with nothing but the matrix instruction in flight (0/1 x +-1
            # nibbles, two waves per SIMD, tools/probe/mfma_probe.hip) is power-limited well below that peak.
            PEAK_FP4 = 1.0e16
            SUSTAINED = 15.35e12
            # Integer-issue bound of the vector-ALU form: per pair 8 XOR (2.8 cycles per wave64 instruction per
            # SIMD, measured) + 8 popcount-accumulate (4.5 cycles), 1024 SIMDs at 2.4 GHz
            issue_bound = 1024 * 2.4e9 * 64 / (8 * 2.8 + 8 * 4.5)
            out["hamming"] = {"pairs_per_s": round(rate, 1), "nq": nq_l * world, "nt": NT,
                              "workload": "seed 0xD35C, 50% planted near-duplicates (<= 30 flips), ties at higher "
                                          "indices, k=2 + ratio test 0.6 / TH_LOW 50 inside the timed region",
                              "path": "matrix cores: v_mfma_scale_f32_32x32x64_f8f6f4, e2m1 operands 0/1 x +-1, exact",
                              "accepted_matches": accepted,
                              "pairs_per_s_falling_distances": round(rate_adv, 1),
                              "pairs_per_s_vector_alu": round(rate_valu, 1),
                              "roofline": {"bound": "mfma", "achieved": round(rate * 512 / world / 1e12, 1),
                                           "peak": PEAK_FP4 / 1e12, "unit": "TFLOP/s",
                                           "frac": round(rate * 512 / world / PEAK_FP4, 4), "traffic": None},
                              "sustained_matrix_rate_pairs_per_s_per_gpu": SUSTAINED,
                              "frac_of_sustained_matrix_rate": round(rate / (SUSTAINED * world), 4),
                              "sustained_note": "profiles/r2_mfma_probe.txt: fp4 32x32x64, 0/1 x +-1, 2 waves per SIMD, "
                                                "operands in registers",
                              "vector_alu_issue_bound_pairs_per_s_per_gpu": round(issue_bound, 1),
                              "exchange": ("all-gather queries + all-to-all partials (RCCL)" if torch.distributed.get_backend() == "nccl"
                                           else "all-gather queries + all-gather partials (gloo rehearsal)") if world > 1 else "none"}
            if rank == 0 and world == 1 and not args.no_cpu:
                # CPU side of the same metric: the oracle's serial scan (the reference's descriptor loop and
                # DescriptorDistance, src/ORBmatcher.cc:255-288, :1905-1921) over a slice of the queries against
                # ALL train rows, query-parallel on the host cores, extrapolated linearly to the full query set
                from oracle import pyoracle as O
                avail, threads = host_cores()
                nqs = min(1 << 12, NQ)
                t1 = time.perf_counter()
                ci, cd, cs = O.knn2(q_h[:nqs], t_h, threads=threads)
                dt_c = time.perf_counter() - t1
                if dt_c < 4.0 and NQ >= (1 << 14):
                    nqs = 1 << 14
                    t1 = time.perf_counter()
                    ci, cd, cs = O.knn2(q_h[:nqs], t_h, threads=threads)
                    dt_c = time.perf_counter() - t1
                r = search(q_l, t_l)
                torch.cuda.synchronize()
                same = (np.array_equal(r[0][:nqs].cpu().numpy(), ci) and np.array_equal(r[1][:nqs].cpu().numpy(), cd)
                        and np.array_equal(r[2][:nqs].cpu().numpy(), cs))
                out["hamming"]["cpu_baseline"] = {
                    "value": round(float(nqs) * NT / dt_c, 1), "unit": "pairs/s", "cores": threads, "cores_available": avail,
                    "kind": "port", "sample": "%d x %d slice of the same data (first queries, all train rows), oracle "
                                              "serial scan with the reference's SWAR popcount, query-parallel; "
                                              "the full job is this rate extrapolated linearly" % (nqs, NT),
                    "extrapolated_seconds_full_job": round(float(NQ) * NT / (float(nqs) * NT / dt_c), 1),
                    "slice_equals_device_result": bool(same)}
            m.close()

        except Exception as e:  # noqa: BLE001
            out["hamming"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if guard is not None:
        guard.cancel()
    # ---- windowed matcher throughput: frame-to-frame SearchByProjection over many independent pairs
    # in one call (each frame against itself, projections = keypoint positions + 1 px noise).
    if rank == 0 and world == 1 and not args.no_hamming:
        try:
            mp = ORBmatcher(0.9, True, device=local_rank)
            mp.set_stream(stream.cuda_stream)
            NP = min(1024, B)
            nk = d_n[:NP].cpu().numpy()
            kview = d_kps.view(B, cap, 28)
            dview = d_desc.view(B, cap, 32)
            fl = kview[:NP].contiguous().view(torch.float32).view(NP, cap, 7)
            g2 = torch.Generator(device=dev)
            g2.manual_seed(7)
            uu = (fl[:, :, 0] + torch.randn((NP, cap), device=dev, generator=g2)).contiguous()
            vv = (fl[:, :, 1] + torch.randn((NP, cap), device=dev, generator=g2)).contiguous()
            ones8 = torch.ones((NP, cap), dtype=torch.uint8, device=dev)
            invz = torch.full((NP, cap), 0.5, dtype=torch.float32, device=dev)
            ur = torch.full((NP, cap), -1.0, dtype=torch.float32, device=dev)
            claimed = torch.full((NP, cap), -1, dtype=torch.int32, device=dev)
            cobs = torch.zeros((NP, cap), dtype=torch.uint8, device=dev)
            pairs = []
            for i in range(NP):
                n = int(nk[i])
                pairs.append(dict(cur_kps=kview[i, :n], cur_desc=dview[i, :n], last_kps=kview[i, :n], last_desc=dview[i, :n],
                                  valid=ones8[i, :n], u=uu[i, :n], v=vv[i, :n], invz=invz[i, :n], has_obs=ones8[i, :n],
                                  cur_u_right=ur[i, :n], claimed=claimed[i, :n], claimed_obs=cobs[i, :n], direction=0,
                                  bounds=(0.0, 0.0, float(W), float(H))))
            sfac = ex.GetScaleFactors()
            pairs = mp.projection_pair_table(pairs)
            counts = mp.SearchByProjectionBatch(pairs, sfac, 40.0, 7.0)
            barrier()
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                claimed.fill_(-1)
                cobs.zero_()
                counts = mp.SearchByProjectionBatch(pairs, sfac, 40.0, 7.0)
            barrier()
            dt = time.perf_counter() - t1
            out["projection_search"] = {"queries_per_s": round(float(nk.sum()) * reps / dt, 1), "pairs": NP,
                                        "queries_per_pair": round(float(nk.mean()), 1), "th": 7.0,
                                        "matches_per_pair": round(float(np.mean(counts)), 1),
                                        "note": "orbm_search_by_projection_batch, device-resident"}
            if not args.no_cpu:
                # CPU side: the oracle's SearchByProjection (src/ORBmatcher.cc:1533-1678) over the first pairs of
                # the same table, single thread (the reference runs it on the tracking thread)
                from oracle import pyoracle as O
                SP = min(32, NP)
                cl_dev = claimed[:SP].cpu().numpy()
                kh = kview[:SP].cpu().numpy().view(KP_DTYPE).reshape(SP, cap)
                dh = dview[:SP].cpu().numpy()
                uh, vh = uu[:SP].cpu().numpy(), vv[:SP].cpu().numpy()
                t_c, q_c, bad = 0.0, 0, 0
                for i in range(SP):
                    n = int(nk[i])
                    fd = O.FrameData(kh[i, :n], dh[i, :n], (0.0, 0.0, float(W), float(H)), sfac)
                    t1 = time.perf_counter()
                    rn, rcl, rco = O.search_by_projection_frames(fd, fd, np.ones(n, np.uint8), uh[i, :n], vh[i, :n],
                                                                 np.full(n, 0.5, np.float32), np.ones(n, np.uint8),
                                                                 np.full(n, -1.0, np.float32), 40.0, 7.0, 0, True)
                    t_c += time.perf_counter() - t1
                    q_c += n
                    bad += int((rcl != cl_dev[i, :n]).sum())
                out["projection_search"]["cpu_baseline"] = {
                    "value": round(q_c / t_c, 1), "unit": "queries/s", "cores": 1, "kind": "port",
                    "sample": "first %d pairs of the same table, oracle, single thread" % SP,
                    "mismatches_vs_device": bad}
            mp.close()
            del pairs, uu, vv, ones8, invz, ur, claimed, cobs, fl
        except Exception as e:  # noqa: BLE001
            out["projection_search"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the other frame shapes of the baseline (rank 0 of a single-GPU run): same schedule, smaller batches
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}
        ex.close()
        frames = d_kps = d_desc = d_n = None
        torch.cuda.empty_cache()
        try:
            extra["kitti_1241x376_2000kp"] = kitti_pipeline(dev, stream, barrier, synth, ORBextractor, ORBmatcher,
                                                            cpu=not args.no_cpu)
        except Exception as e:  # noqa: BLE001
            extra["kitti_1241x376_2000kp"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        for name, (w2, h2, nf2, b2, ch2) in {"uhd_3840x2160_10000kp": (3840, 2160, 10000, 128, 32)}.items():
            try:
                e2 = ORBextractor(nf2, 1.2, NLEVELS, 20, 7, max_width=w2, max_height=h2, max_batch=ch2, device=local_rank)
                e2.set_stream(stream.cuda_stream)
                cap2 = e2.max_keypoints()
                pitch = (w2 + 3) // 4 * 4
                buf = torch.zeros((b2, h2, pitch), dtype=torch.uint8, device=dev)
                for f0 in range(0, b2, 16):
                    n = min(16, b2 - f0)
                    buf[f0:f0 + n, :, :w2] = synth.synth_batch_torch(n, w2, h2, 0x5EED7001 + f0, device=dev)
                k2 = torch.zeros(b2 * cap2 * 28, dtype=torch.uint8, device=dev)
                dd2 = torch.zeros(b2 * cap2 * 32, dtype=torch.uint8, device=dev)
                n2 = torch.zeros(b2, dtype=torch.int32, device=dev)
                for _ in range(2):
                    e2.extract_batch_device(buf[:, :, :w2], k2, dd2, n2, cap2)
                barrier()
                t1 = time.perf_counter()
                reps = 8
                for _ in range(reps):
                    e2.extract_batch_device(buf[:, :, :w2], k2, dd2, n2, cap2)
                barrier()
                dt = time.perf_counter() - t1
                extra[name] = {"frames_per_s": round(b2 * reps / dt, 1), "frames": b2, "frames_per_pass": ch2,
                               "mean_keypoints": round(float(n2.float().mean().item()), 1)}
                e2.close()
                del buf, k2, dd2, n2
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["other_shapes"] = extra
        ex = None
    if rank == 0:
        if not headline_printed:
            print(json.dumps(out), flush=True)
        elif "hamming" in out:
            print("hamming: " + json.dumps(out["hamming"]), file=sys.stderr, flush=True)
            try:
                d = os.path.join(ROOT, "gpurun_out")
                if os.path.isdir(d):
                    json.dump(out["hamming"], open(os.path.join(d, "hamming_%d.json" % world), "w"))
            except OSError:
                pass
    if ex is not None:
        ex.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
