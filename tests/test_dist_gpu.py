"""The product under the multi-rank choreography: two ranks, started as fresh child processes on
the one GPU of the box (gloo carries the exchange; RCCL wants one GPU per rank), each running the
HIP 2-NN scan of its train shard, the exchange of slam_amd.dist.knn2_sharded and the HIP merge,
plus its share of a frame-sharded extraction.  Checked against the oracle (2-NN) and against a
single-process pass over all frames (extraction, byte for byte)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_hip_knn2_and_frame_sharding(tmp_path, oracle):
    world, nq, nt, nframes = 2, 1 << 12, (1 << 15) + 7, 9
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ORBX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path),
                                       str(nq), str(nt), str(nframes)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for k in procs:
                k.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # ---- 2-NN: every rank finished its own query slice; together they equal the serial scan
    q, t, meta = synth.synth_knn_workload(nq, nt, seed=0xD35C)
    oi, od, os_ = oracle.knn2(q, t, threads=8)
    bi = np.concatenate([r["bi"] for r in res])
    bd = np.concatenate([r["bd"] for r in res])
    sd = np.concatenate([r["sd"] for r in res])
    acc = np.concatenate([r["acc"] for r in res])
    assert np.array_equal(bi, oi) and np.array_equal(bd, od) and np.array_equal(sd, os_)
    want = (od <= 50) & (od.astype(np.float32) < np.float32(0.6) * os_.astype(np.float32))
    assert np.array_equal(acc.astype(bool), want) and want.sum() > nq // 4
    pl = meta["planted"] >= 0
    assert np.array_equal(bi[pl], meta["planted"][pl].astype(np.int32))      # ties across the shard boundary too
    # ---- extraction: the ranks' frame ranges put together equal one pass over all frames
    from slam_amd._lib import KP_DTYPE
    from slam_amd.extractor import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nframes)
    try:
        frames = synth.synth_batch(nframes, 640, 480, 0x5EED0001)
        kps, desc, n = ex.extract_batch(frames)
        cap = ex.max_keypoints()
    finally:
        ex.close()
    covered = 0
    for r in res:
        f0, f1 = (int(v) for v in r["frames"])
        assert int(r["cap"][0]) == cap
        rk = r["kps"].view(KP_DTYPE).reshape(f1 - f0, cap)
        rd = r["desc"].reshape(f1 - f0, cap, 32)
        for f in range(f0, f1):
            assert r["n"][f - f0] == n[f]
            assert rk[f - f0, :n[f]].tobytes() == kps[f, :n[f]].tobytes()
            assert np.array_equal(rd[f - f0, :n[f]], desc[f, :n[f]])
            covered += 1
    assert covered == nframes


def test_rccl_collectives_with_one_rank(tmp_path, oracle):
    """The RCCL branch of the sharded 2-NN (device all_gather_into_tensor of the queries, all_to_all_single
    of the 12-byte partials, then the HIP merge) on the one GPU of the box: a process group of ONE rank on
    the nccl backend, the exchange forced although one rank has nothing to exchange.  Result = the serial
    scan; the frame-sharded extraction of the same worker covers all frames."""
    nq, nt, nframes = 1 << 12, (1 << 15) + 7, 3
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(free_port()), ORBX_DIST_BACKEND="nccl", ORBX_DIST_FORCE_COLLECTIVES="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(nq), str(nt),
                          str(nframes)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        o, _ = p.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode == 0, o.decode(errors="replace")[-2000:]
    r = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    q, t, meta = synth.synth_knn_workload(nq, nt, seed=0xD35C)
    oi, od, os_ = oracle.knn2(q, t, threads=8)
    assert np.array_equal(r["bi"], oi) and np.array_equal(r["bd"], od) and np.array_equal(r["sd"], os_)
    want = (od <= 50) & (od.astype(np.float32) < np.float32(0.6) * os_.astype(np.float32))
    assert np.array_equal(r["acc"].astype(bool), want)
    assert tuple(int(v) for v in r["frames"]) == (0, nframes)
