"""World-size-2 rehearsal of the multi-GPU paths on CPU (gloo).

The collective choreography of slam_amd.dist is exercised with the oracle as
the per-rank compute (tests may call the oracle; the GPU path plugs the HIP
kernels into the same two hooks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, nt, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import pyoracle as O
    from slam_amd import dist as sdist
    from slam_amd import synth
    r, _, w = sdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    q = synth.synth_descriptors(nq, 5)
    t = synth.synth_descriptors(nt, 6)
    t[10:10 + nq // 2] = synth.perturb_descriptors(q[:nq // 2], 2)
    t[nt - nq // 4:] = t[10:10 + nq // 4]          # duplicates in the last shard: lower index must win
    q0, q1 = sdist.shard_range(nq, rank, world)
    t0, t1 = sdist.shard_range(nt, rank, world)

    def local_knn2(q_all, t_loc, off):
        bi, bd, sd = O.knn2(q_all.numpy(), t_loc.numpy())
        bi = np.where(bi >= 0, bi + off, -1).astype(np.int32)
        return torch.from_numpy(bi), torch.from_numpy(bd), torch.from_numpy(sd)

    def merge(a, b):
        m = sdist.merge_partials(tuple(x.numpy() for x in a), tuple(x.numpy() for x in b))
        return tuple(torch.from_numpy(x) for x in m)

    idx, d, s = sdist.knn2_sharded(torch.from_numpy(q[q0:q1].copy()), torch.from_numpy(t[t0:t1].copy()), t0,
                                   local_knn2, merge)
    ri, rd, rs = O.knn2(q[q0:q1], t)
    ok = np.array_equal(idx.numpy(), ri) and np.array_equal(d.numpy(), rd) and np.array_equal(s.numpy(), rs)

    # frame sharding: each rank extracts its contiguous range; the gathered counts equal a single pass
    frames = synth.synth_batch(5, 320, 240, 900)
    f0, f1 = sdist.shard_range(5, rank, world)
    p = O.params(nfeatures=300)
    mine = [len(O.extract(p, frames[f])[0]) for f in range(f0, f1)]
    counts = [None] * world
    dist.all_gather_object(counts, (f0, mine))
    flat = [c for _, part in sorted(counts) for c in part]
    ok = ok and flat == [len(O.extract(p, fr)[0]) for fr in frames]
    open(os.path.join(out_dir, "rank%d.ok" % rank), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from slam_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096, 1000003):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_merge_rule_reproduces_serial_scan(oracle):
    from slam_amd import synth
    from slam_amd.dist import merge_partials
    q = synth.synth_descriptors(200, 1)
    t = synth.synth_descriptors(900, 2)
    t[50:150] = synth.perturb_descriptors(q[:100], 8)
    t[600:700] = t[50:150]
    ref = oracle.knn2(q, t)
    for cuts in ((0, 300, 900), (0, 100, 650, 900), (0, 1, 2, 900)):
        acc = None
        for a, b in zip(cuts[:-1], cuts[1:]):
            bi, bd, sd = oracle.knn2(q, t[a:b])
            part = (np.where(bi >= 0, bi + a, -1).astype(np.int32), bd, sd)
            acc = part if acc is None else merge_partials(acc, part)
        assert all(np.array_equal(x, y) for x, y in zip(acc, ref))


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), 64, 700, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.ok" % r)).read() == "1"
