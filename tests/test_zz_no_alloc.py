"""The 2-NN entry point that allocates nothing (orbm_knn2_noalloc: partial results in a static
device array) against orbm_knn2 on device memory, through the C ABI.

Bar: the three output arrays bit-exact.  Two calls in a row on one context, the second with a
smaller input, so that what the first call left in the shared array is seen not to leak; then the
size limits.  Collected last (the file name sorts after every other test file).
"""
import ctypes as C

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def both(m, q, t, base):
    import torch
    dq = torch.from_numpy(q).cuda()
    dt = torch.from_numpy(t).cuda() if len(t) else torch.zeros((0, 32), dtype=torch.uint8, device="cuda")
    outs = []
    for fn in (m.knn2_device, m.knn2_noalloc_device):
        bi, bd, sd = (torch.full((len(q),), -7, dtype=torch.int32, device="cuda") for _ in range(3))
        torch.cuda.synchronize()
        fn(dq, dt, bi, bd, sd, base)
        m.synchronize()
        outs.append((bi.cpu().numpy(), bd.cpu().numpy(), sd.cpu().numpy()))
    return outs


def test_noalloc_equals_knn2_twice_second_call_smaller(oracle):
    from slam_amd.matcher import ORBmatcher
    m = ORBmatcher(0.6, True)
    try:
        # near-duplicates and exact ties so that best index and second distance are exercised
        t1 = synth.synth_descriptors(20000, 11)
        q1 = t1[::7][:2500].copy()
        q1[:, 3] ^= 0x15
        q1[::5] = t1[5::35][:len(q1[::5])]
        t1[9000:9100] = t1[100:200]
        for (q, t, base) in ((q1, t1, 1000), (q1[:333], t1[:4999], 0)):
            ref, got = both(m, q, t, base)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b)
            oi, od, os_ = oracle.knn2(q, t)
            assert np.array_equal(got[0], np.where(oi >= 0, oi + base, -1))
            assert np.array_equal(got[1], od) and np.array_equal(got[2], os_)
        # more shapes: one query, few rows, no rows, many queries (one slice)
        for nq, nt in ((1, 77), (5, 3), (64, 0), (70000, 260)):
            q = synth.synth_descriptors(nq, 100 + nq)
            t = synth.synth_descriptors(max(nt, 1), 200 + nt)[:nt]
            ref, got = both(m, q, t, 5)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b), (nq, nt)
    finally:
        m.close()


def test_noalloc_rejects_sizes_beyond_its_scratch():
    import torch
    from slam_amd import _lib
    from slam_amd.matcher import ORBmatcher
    m = ORBmatcher(0.6, True)
    try:
        lib = _lib.load()
        d = torch.zeros((64, 32), dtype=torch.uint8, device="cuda")
        o = torch.zeros(64, dtype=torch.int32, device="cuda")
        rc = lib.orbm_knn2_noalloc(m._m, d.data_ptr(), (1 << 20) + 1, d.data_ptr(), 64, 0, o.data_ptr(), o.data_ptr(), o.data_ptr())
        assert rc == -4 and b"exceed" in lib.orbx_last_error()
        rc = lib.orbm_knn2_noalloc(m._m, d.data_ptr(), 64, d.data_ptr(), (1 << 22) + 1, 0, o.data_ptr(), o.data_ptr(), o.data_ptr())
        assert rc == -4
        rc = lib.orbm_knn2_noalloc(m._m, d.data_ptr() + 1, 8, d.data_ptr(), 64, 0, o.data_ptr(), o.data_ptr(), o.data_ptr())
        assert rc == -1
    finally:
        m.close()


def test_noalloc_two_contexts_on_two_streams_do_not_race(oracle):
    """Two matchers (the reference keeps one ORBmatcher per call site and thread), each on a stream of its
    own, sharing the GPU's one scratch array: calls enqueued back to back without waiting must still
    give each call its own result (the library orders them on the device)."""
    import torch
    from slam_amd.matcher import ORBmatcher
    a, b = ORBmatcher(0.6, True), ORBmatcher(0.6, True)
    try:
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        a.set_stream(sa.cuda_stream)
        b.set_stream(sb.cuda_stream)
        cases = []
        for k, (nq, nt) in enumerate(((3000, 40000), (2500, 30000), (900, 52000), (4000, 9000))):
            q = synth.synth_descriptors(nq, 300 + k)
            t = synth.synth_descriptors(nt, 400 + k)
            t[:nq // 2] = synth.perturb_descriptors(q[:nq // 2], 17 + k)
            dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
            outs = tuple(torch.full((nq,), -7, dtype=torch.int32, device="cuda") for _ in range(3))
            cases.append((q, t, dq, dt, outs))
        torch.cuda.synchronize()
        for rep in range(3):
            for k, (q, t, dq, dt, outs) in enumerate(cases):   # alternate the contexts, never wait in between
                (a if k % 2 == 0 else b).knn2_noalloc_device(dq, dt, outs[0], outs[1], outs[2], 0)
        torch.cuda.synchronize()
        for q, t, dq, dt, outs in cases:
            oi, od, os_ = oracle.knn2(q, t, threads=8)
            assert np.array_equal(outs[0].cpu().numpy(), oi)
            assert np.array_equal(outs[1].cpu().numpy(), od) and np.array_equal(outs[2].cpu().numpy(), os_)
    finally:
        a.close()
        b.close()


def test_noalloc_allocates_nothing():
    """The call runs under stream capture (an allocation, a free or a synchronising copy would end the
    capture with an error) and the replayed graph gives orbm_knn2's result; free device memory is the same
    before and after plain calls."""
    import torch
    from slam_amd.matcher import ORBmatcher
    m = ORBmatcher(0.6, True)
    try:
        q = synth.synth_descriptors(3000, 71)
        t = synth.synth_descriptors(50000, 72)
        t[100:1600] = synth.perturb_descriptors(q[:1500], 73)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        ref = tuple(torch.full((len(q),), -7, dtype=torch.int32, device="cuda") for _ in range(3))
        got = tuple(torch.full((len(q),), -7, dtype=torch.int32, device="cuda") for _ in range(3))
        m.knn2_device(dq, dt, ref[0], ref[1], ref[2], 0)
        m.knn2_noalloc_device(dq, dt, got[0], got[1], got[2], 0)       # warm: code object loaded
        m.synchronize()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(4):
            m.knn2_noalloc_device(dq, dt, got[0], got[1], got[2], 0)
        m.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        for x in got:
            x.fill_(-9)
        s = torch.cuda.Stream()
        m.set_stream(s.cuda_stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            m.knn2_noalloc_device(dq, dt, got[0], got[1], got[2], 0)
        torch.cuda.synchronize()
        assert (got[0] == -9).all()            # captured, not run
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        m.reset_stream()
    finally:
        m.close()
