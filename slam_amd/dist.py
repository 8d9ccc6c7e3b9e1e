"""One process per GPU: frame sharding for extraction and the sharded kNN-2 exchange.

Extraction and windowed matching shard by frame (or frame pair) with no
data-path collective.  Brute-force kNN-2 shards the train set by rows; the
queries are all-gathered, each rank searches its shard, and the per-query
partials (best index, best distance, second distance) are exchanged so that
each rank finishes its own slice of the queries.  Partials are merged in shard
order, which reproduces the serial scan of the reference's descriptor loops
(reference src/ORBmatcher.cc:255-282): lowest global index wins ties, and the
second best counts duplicates of the minimum.
"""
import os

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None, force=False):
    """Joins the process group of the torch.distributed.run environment.  A single process needs none and
    gets none, unless `force` asks for a group of one (the collectives of knn2_sharded then run through the
    backend with one rank: the way to exercise the RCCL calls on a one-GPU machine)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # ORBX_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (RCCL wants one GPU per rank)
            backend = os.environ.get("ORBX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n, rank, world):
    """Contiguous [begin, end) of n items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def merge_partials(a, b):
    """Host (numpy) statement of the merge rule for shard a preceding shard b; the
    device path uses the library's merge kernel, which applies the same rule."""
    ia, da, sa = a
    ib, db, sb = b
    take_b = db < da
    idx = np.where(take_b, ib, ia)
    d = np.minimum(da, db)
    s = np.minimum(np.maximum(da, db), np.minimum(sa, sb))
    return idx.astype(np.int32), d.astype(np.int32), s.astype(np.int32)


def knn2_sharded(q_local, t_local, t_offset, local_knn2, merge=merge_partials, group=None, force_collectives=False):
    """Sharded brute-force kNN-2.

    q_local: this rank's slice of the queries, torch uint8 (nq_local, 32), equal on all ranks
             in length.  t_local: this rank's rows of the train set, whose first row has
    global index t_offset.  local_knn2(q_all, t_local, t_offset) -> (idx, d, second) int32
    tensors over ALL queries.  merge((ia,da,sa),(ib,db,sb)) merges two partials, a first.
    Returns the merged (idx, d, second) for this rank's query slice.  With one rank nothing is exchanged,
    unless force_collectives asks for the all-gather and the all-to-all to run anyway (a group of one must
    then exist).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nq = q_local.shape[0]
    if world == 1 and not (force_collectives and dist.is_initialized()):
        return local_knn2(q_local, t_local, t_offset)
    rccl = dist.get_backend(group) == "nccl"

    def gather(x):
        """All ranks' x, concatenated along dimension 0.  Under gloo (rehearsals of several ranks on one
        GPU, CPU tests) the exchange goes through host tensors: gloo's handling of device tensors
        stalled now and then when two ranks shared a GPU."""
        if rccl or not x.is_cuda:
            out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(out, x.contiguous(), group=group)
            return out
        h = x.contiguous().cpu()
        out = torch.empty((world * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h, group=group)
        return out.to(x.device)

    q_all = gather(q_local)
    idx, d, s = local_knn2(q_all, t_local, t_offset)
    mine = torch.stack([idx, d, s], dim=1).to(torch.int32).reshape(world, nq, 3).contiguous()
    if rccl:
        # slice j of every rank's partials goes to rank j: one all-to-all over xGMI
        recv = torch.empty_like(mine)
        dist.all_to_all_single(recv, mine, group=group)
    else:
        # gloo has no all-to-all: gather everything and keep this rank's slice
        full = gather(mine.reshape(1, -1))
        recv = full.reshape(world, world, nq, 3)[:, rank].contiguous()
    acc = (recv[0, :, 0].contiguous(), recv[0, :, 1].contiguous(), recv[0, :, 2].contiguous())
    for r in range(1, world):
        part = (recv[r, :, 0].contiguous(), recv[r, :, 1].contiguous(), recv[r, :, 2].contiguous())
        acc = merge(acc, part)
    return acc
