"""The matrix-core form of the brute-force 2-NN scan against the oracle's serial scan
(reference src/ORBmatcher.cc:255-282): forced on ragged sizes, ties, duplicates, slices of the
train rows and both orderings of the data, and against the vector-ALU form at a larger size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    from slam_amd.matcher import ORBmatcher
    m = ORBmatcher(0.6, True)
    yield m
    m.debug_knn_path(0)
    m.close()


def _check(matcher, oracle, q, t, base=0):
    matcher.debug_knn_path(2)
    bi, bd, sd = matcher.knn2(q, t, base)
    ri, rd, rs = oracle.knn2(q, t, threads=16)
    ri = np.where(ri >= 0, ri + base, ri)
    assert np.array_equal(bd, rd)
    assert np.array_equal(sd, rs)
    assert np.array_equal(bi, ri)


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 63), (31, 64), (33, 65), (127, 129), (512, 511), (513, 1000), (700, 4097),
                                   (2049, 2047), (5000, 333)])
def test_ragged_sizes(matcher, oracle, nq, nt):
    from slam_amd import synth
    q = synth.synth_descriptors(nq, 100 + nq)
    t = synth.synth_descriptors(nt, 200 + nt)
    _check(matcher, oracle, q, t)
    _check(matcher, oracle, q, t, base=12345)


def test_ties_and_duplicates(matcher, oracle):
    from slam_amd import synth
    rng = np.random.default_rng(5)
    t = synth.synth_descriptors(3000, 9)
    t[1500:] = t[:1500]                       # every row twice: lowest index wins, second = best
    q = t[rng.integers(0, 3000, 2000)].copy()
    flip = rng.integers(0, 256, 2000)
    q[np.arange(2000), flip >> 3] ^= (1 << (flip & 7)).astype(np.uint8)
    _check(matcher, oracle, q, t)
    # all rows equal: distance 0 everywhere, index 0, second 0
    t2 = np.repeat(t[:1], 777, 0)
    _check(matcher, oracle, t2[:100], t2)
    # extreme distances: complements (256) and zeros
    z = np.zeros((130, 32), np.uint8)
    o = np.full((70, 32), 255, np.uint8)
    _check(matcher, oracle, z, o)
    _check(matcher, oracle, np.concatenate([z, o]), np.concatenate([o, z[:3]]))


def test_adversarial_orderings(matcher, oracle):
    from slam_amd import synth
    q, t = synth.synth_knn_adversarial(1024, 8192)
    _check(matcher, oracle, q, t)
    _check(matcher, oracle, q, t[::-1].copy())


def test_sliced_and_whole_scans_agree_with_the_vector_form(matcher):
    import torch
    from slam_amd import synth
    dev = torch.device("cuda:0")
    q, t, _ = synth.synth_knn_workload(1 << 15, 1 << 17)
    tq, tt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    outs = []
    for path in (1, 2):
        matcher.debug_knn_path(path)
        o = [torch.empty(len(q), dtype=torch.int32, device=dev) for _ in range(3)]
        matcher.knn2_device(tq, tt, *o)
        torch.cuda.synchronize()
        outs.append([x.cpu().numpy() for x in o])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_entry_point_without_allocations_takes_the_same_path(matcher, oracle):
    import torch
    from slam_amd import synth
    dev = torch.device("cuda:0")
    matcher.debug_knn_path(2)
    for nq, nt in ((3000, 5000), (700, 901)):
        q = synth.synth_descriptors(nq, 31 + nq)
        t = synth.synth_descriptors(nt, 32 + nt)
        tq, tt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
        o = [torch.full((nq,), -7, dtype=torch.int32, device=dev) for _ in range(3)]
        matcher.knn2_noalloc_device(tq, tt, *o, index_base=5)
        torch.cuda.synchronize()
        ri, rd, rs = oracle.knn2(q, t, threads=16)
        assert np.array_equal(o[0].cpu().numpy(), ri + 5)
        assert np.array_equal(o[1].cpu().numpy(), rd)
        assert np.array_equal(o[2].cpu().numpy(), rs)
