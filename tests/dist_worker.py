"""One rank of the multi-process GPU test (tests/test_dist_gpu.py starts two of these as fresh
child processes; RANK / WORLD_SIZE / MASTER_* come from the environment).  Runs the product
path under the multi-rank choreography: the sharded brute-force 2-NN (HIP scan of this rank's
train rows, exchange, HIP merge) and its share of a frame-sharded extraction, and writes what it
computed to <outdir>/rank<r>.npz.  No oracle here: the parent compares."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from slam_amd import dist as sdist  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.extractor import ORBextractor  # noqa: E402
from slam_amd.matcher import ORBmatcher  # noqa: E402


def main():
    outdir, nq, nt, nframes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    force = os.environ.get("ORBX_DIST_FORCE_COLLECTIVES") == "1"   # a group of one rank, exchange still made
    rank, _, world = sdist.init_process_group(force=force)
    torch.cuda.set_device(0)   # every rank of this rehearsal shares the one GPU of the box
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    m = ORBmatcher(0.6, True)
    m.set_stream(stream.cuda_stream)
    q, t, _ = synth.synth_knn_workload(nq, nt, seed=0xD35C)
    nq_l = nq // world
    t0, t1 = sdist.shard_range(nt, rank, world)
    q_l = torch.from_numpy(q[rank * nq_l:(rank + 1) * nq_l]).to(dev)
    t_l = torch.from_numpy(t[t0:t1]).to(dev)

    def local_knn2(q_all, t_loc, off):
        bi = torch.empty(q_all.shape[0], dtype=torch.int32, device=dev)
        bd, sd = torch.empty_like(bi), torch.empty_like(bi)
        m.knn2_device(q_all, t_loc, bi, bd, sd, off)
        return bi, bd, sd

    def merge(a, b):
        m.knn2_merge_device(a[0], a[1], a[2], b[0], b[1], b[2])
        return a

    bi, bd, sd = sdist.knn2_sharded(q_l, t_l, t0, local_knn2, merge, force_collectives=force)
    acc = torch.empty(nq_l, dtype=torch.uint8, device=dev)
    m.ratio_test_device(bd, sd, acc)
    # frame-sharded extraction: contiguous frame ranges, no collective
    f0, f1 = sdist.shard_range(nframes, rank, world)
    frames = synth.synth_batch_torch(nframes, 640, 480, 0x5EED0001, device=dev)[f0:f1].contiguous()
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=4)
    ex.set_stream(stream.cuda_stream)
    cap = ex.max_keypoints()
    nf = f1 - f0
    kps = torch.zeros(nf * cap * 28, dtype=torch.uint8, device=dev)
    desc = torch.zeros(nf * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(nf, dtype=torch.int32, device=dev)
    ex.extract_batch_device(frames, kps, desc, n, cap)
    ex.synchronize()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), bi=bi.cpu().numpy(), bd=bd.cpu().numpy(), sd=sd.cpu().numpy(),
             acc=acc.cpu().numpy(), kps=kps.cpu().numpy(), desc=desc.cpu().numpy(), n=n.cpu().numpy(),
             frames=np.array([f0, f1]), cap=np.array([cap]))
    m.close()
    ex.close()
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
