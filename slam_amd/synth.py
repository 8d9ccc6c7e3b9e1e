"""Deterministic synthetic inputs (images, descriptors) shared by tests and bench.

Everything is derived from a splitmix64 hash of (seed, index), so the bytes are
identical on every machine and numpy version.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def hash_u64(seed, n, stream=0):
    """n pseudo-random uint64 values for (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64) ^ (np.uint64(stream) << np.uint64(32)))
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
        return _splitmix64(idx + base[0])


def _box3(img):
    """3x3 box filter with edge replication, integer rounding."""
    p = np.pad(img.astype(np.int32), 1, mode="edge")
    s = sum(p[dy:dy + img.shape[0], dx:dx + img.shape[1]] for dy in range(3) for dx in range(3))
    return ((s + 4) // 9).astype(np.int32)


def synth_image(w, h, seed, rects=64):
    """A textured u8 image with many corners: blocky noise octaves, filled
    rectangles and fine noise, lightly smoothed.  Returns (h, w) uint8."""
    acc = np.zeros((h, w), dtype=np.int32)
    stream = 1
    for block, weight in ((32, 96), (16, 64), (8, 56)):
        gh, gw = (h + block - 1) // block + 1, (w + block - 1) // block + 1
        g = (hash_u64(seed, gh * gw, stream) >> np.uint64(56)).astype(np.int32).reshape(gh, gw)
        stream += 1
        up = np.repeat(np.repeat(g, block, axis=0), block, axis=1)
        oy = int(hash_u64(seed, 1, 100 + stream)[0] % np.uint64(block))
        ox = int(hash_u64(seed, 1, 200 + stream)[0] % np.uint64(block))
        acc += weight * up[oy:oy + h, ox:ox + w]
    img = acc // 256  # roughly 0..215
    r = hash_u64(seed, rects * 5, 50)
    for i in range(rects):
        rw = 8 + int(r[5 * i] % np.uint64(max(8, w // 6)))
        rh = 8 + int(r[5 * i + 1] % np.uint64(max(8, h // 6)))
        x0 = int(r[5 * i + 2] % np.uint64(max(1, w - rw)))
        y0 = int(r[5 * i + 3] % np.uint64(max(1, h - rh)))
        img[y0:y0 + rh, x0:x0 + rw] = int(r[5 * i + 4] >> np.uint64(56))
    img = _box3(img)
    fine = (hash_u64(seed, w * h, 60) >> np.uint64(61)).astype(np.int32).reshape(h, w)  # 0..7
    img = np.clip(img + fine - 3, 0, 255)
    return img.astype(np.uint8)


def synth_batch(n, w, h, seed0):
    """n frames (n, h, w) uint8 with seeds seed0 + frame index."""
    out = np.empty((n, h, w), dtype=np.uint8)
    for i in range(n):
        out[i] = synth_image(w, h, seed0 + i)
    return out


def synth_stereo_pair(w, h, seed, disparity_top, disparity_bottom, noise=0):
    """A rectified pair cut from one scene: rows above h/2 are displaced by disparity_top
    pixels between the views, the rest by disparity_bottom; the right view gets uniform
    noise in [-noise, noise].  Returns (left, right), (h, w) uint8 each."""
    margin = 64
    assert 0 <= disparity_top <= margin and 0 <= disparity_bottom <= margin
    scene = synth_image(w + 2 * margin, h, seed)
    left = np.ascontiguousarray(scene[:, margin:margin + w])
    right = np.empty_like(left)
    right[:h // 2] = scene[:h // 2, margin + disparity_top:margin + disparity_top + w]
    right[h // 2:] = scene[h // 2:, margin + disparity_bottom:margin + disparity_bottom + w]
    if noise:
        r = (hash_u64(seed, w * h, 70) % np.uint64(2 * noise + 1)).astype(np.int32).reshape(h, w) - noise
        right = np.clip(right.astype(np.int32) + r, 0, 255).astype(np.uint8)
    return left, np.ascontiguousarray(right)


def synth_stereo_sequence(w, h, seed, disparity_top, disparity_bottom, shifts, noise=0):
    """Consecutive rectified stereo frames of one static scene: frame k is the window of the scene
    displaced by shifts[k] = (dx, dy) pixels (|dx|, |dy| <= 16), with the disparities and the noise
    model of synth_stereo_pair (the disparity changes at a fixed row of the scene; every image of
    the sequence has its own noise).  Returns [(left, right), ...], (h, w) uint8 each."""
    margin, vm = 64 + 16, 16
    assert 0 <= disparity_top <= 64 and 0 <= disparity_bottom <= 64
    scene = synth_image(w + 2 * margin, h + 2 * vm, seed)
    split = vm + h // 2
    out = []
    for k, (dx, dy) in enumerate(shifts):
        assert abs(dx) <= 16 and abs(dy) <= 16
        x0, y0 = margin + dx, vm + dy
        left = scene[y0:y0 + h, x0:x0 + w]
        rows = np.arange(y0, y0 + h)
        right = np.where((rows < split)[:, None], scene[y0:y0 + h, x0 + disparity_top:x0 + disparity_top + w],
                         scene[y0:y0 + h, x0 + disparity_bottom:x0 + disparity_bottom + w])
        imgs = []
        for side, img in enumerate((left, right)):
            if noise:
                r = (hash_u64(seed, w * h, 80 + 2 * k + side) % np.uint64(2 * noise + 1)).astype(np.int32).reshape(h, w) - noise
                img = np.clip(img.astype(np.int32) + r, 0, 255)
            imgs.append(np.ascontiguousarray(img, dtype=np.uint8))
        out.append(tuple(imgs))
    return out


def synth_descriptors(n, seed):
    """n random 256-bit descriptors as (n, 32) uint8."""
    v = hash_u64(seed, n * 4, 7)
    return v.view(np.uint8).reshape(n, 32).copy()


def perturb_descriptors(desc, seed, max_flips=30):
    """Copy of desc with 0..max_flips random bit flips per row (planted near-duplicates)."""
    n = desc.shape[0]
    out = desc.copy()
    r = hash_u64(seed, n * (max_flips + 1), 9).reshape(n, max_flips + 1)
    nflip = (r[:, 0] % np.uint64(max_flips + 1)).astype(np.int64)
    for j in range(max_flips):
        bit = (r[:, j + 1] % np.uint64(256)).astype(np.int64)
        rows = np.nonzero(nflip > j)[0]
        out[rows, bit[rows] >> 3] ^= (1 << (bit[rows] & 7)).astype(np.uint8)
    return out


def popcount_rows(x):
    """Number of set bits of each row of an (n, 32) uint8 array."""
    return np.unpackbits(x, axis=1).sum(axis=1).astype(np.int32)


def synth_knn_workload(nq, nt, seed=0xD35C, max_flips=30, dup_every=16):
    """Brute-force matching workload: nt random train descriptors; every even query is a train
    row with at most max_flips flipped bits (a planted near-duplicate, so that the ratio test
    has something to accept), every odd query is random.  Every dup_every-th planted query has
    its train row repeated at a HIGHER index (an exact tie: the lower index must win and the
    second-best distance must equal the best).  Returns q, t and a dict: planted (index of the
    planted row or -1), planted_d (its distance), has_dup (bool)."""
    t = synth_descriptors(nt, seed)
    q = synth_descriptors(nq, seed + 1)
    lo = (3 * nt) // 4
    n_pl = (nq + 1) // 2
    r = np.random.RandomState(seed & 0x7FFFFFFF)
    src = r.permutation(lo)[:n_pl] if n_pl <= lo else r.randint(0, lo, n_pl)
    planted = np.full(nq, -1, np.int64)
    planted[0::2] = src[:len(planted[0::2])]
    has_dup = np.zeros(nq, bool)
    n_dup = min(len(src) // dup_every, nt - lo)
    if n_dup > 0:
        dup_q = np.arange(n_dup) * 2 * dup_every          # planted queries that get a tie
        dup_rows = lo + r.permutation(nt - lo)[:n_dup]       # distinct rows above every planted row
        t[dup_rows] = t[planted[dup_q]]
        has_dup[dup_q] = True
    pl = np.nonzero(planted >= 0)[0]
    q[pl] = perturb_descriptors(t[planted[pl]], seed + 2, max_flips)
    planted_d = np.full(nq, -1, np.int32)
    planted_d[pl] = popcount_rows(q[pl] ^ t[planted[pl]])
    return q, t, dict(planted=planted, planted_d=planted_d, has_dup=has_dup)


def synth_knn_adversarial(nq, nt, seed=0xD35C):
    """Worst case for a scan that skips its top-2 update when no distance can improve: train row
    j differs from a centre descriptor in 256 * (nt - 1 - j) // nt fixed bit positions (nested
    sets), and every query is the centre with a few flips, so along the scan the distance of
    every query keeps falling and nearly every row is a new best."""
    c = synth_descriptors(1, seed + 7)[0]
    order = np.random.RandomState((seed + 7) & 0x7FFFFFFF).permutation(256)
    masks = np.zeros((257, 32), np.uint8)
    for k in range(1, 257):
        masks[k] = masks[k - 1]
        masks[k, order[k - 1] >> 3] |= np.uint8(1 << (order[k - 1] & 7))
    kj = (256 * (nt - 1 - np.arange(nt, dtype=np.int64))) // nt
    t = c[None, :] ^ masks[kj]
    q = perturb_descriptors(np.repeat(c[None, :], nq, axis=0), seed + 8, 8)
    return q, np.ascontiguousarray(t)


# ---------------------------------------------------------------- torch twin
def _t_splitmix64(x):
    """splitmix64 on torch int64 (two's-complement wrap == uint64 arithmetic)."""
    import torch

    def lsr(v, s):  # logical shift right
        return (v >> s) & ((1 << (64 - s)) - 1)

    c1 = torch.tensor(0x9E3779B97F4A7C15 - (1 << 64), dtype=torch.int64, device=x.device)
    c2 = torch.tensor(0xBF58476D1CE4E5B9 - (1 << 64), dtype=torch.int64, device=x.device)
    c3 = torch.tensor(0x94D049BB133111EB - (1 << 64), dtype=torch.int64, device=x.device)
    x = x + c1
    x = (x ^ lsr(x, 30)) * c2
    x = (x ^ lsr(x, 27)) * c3
    return x ^ lsr(x, 31)


def _t_hash(seeds, n, stream):
    """(B, n) int64 holding the same bits as hash_u64(seed, n, stream) per row."""
    import torch
    base = _t_splitmix64(seeds ^ (stream << 32))
    c = torch.tensor(0xD1342543DE82EF95 - (1 << 64), dtype=torch.int64, device=seeds.device)
    idx = torch.arange(n, dtype=torch.int64, device=seeds.device) * c
    return _t_splitmix64(idx[None, :] + base[:, None])


def _t_top_bits(v, bits):
    return (v >> (64 - bits)) & ((1 << bits) - 1)


def _t_umod(v, m):
    """(uint64 view of v) % m for small positive m."""
    hi = ((v >> 1) & 0x7FFFFFFFFFFFFFFF)  # floor(u / 2)
    return ((hi % m) * 2 + (v & 1)) % m


def synth_batch_torch(n, w, h, seed0, device="cpu", rects=64):
    """Same bytes as synth_batch(n, w, h, seed0), computed with torch ops on `device`."""
    import torch
    import torch.nn.functional as F
    seeds = torch.arange(seed0, seed0 + n, dtype=torch.int64, device=device)
    ys = torch.arange(h, dtype=torch.int64, device=device)
    xs = torch.arange(w, dtype=torch.int64, device=device)
    bidx = torch.arange(n, device=device)[:, None, None]
    acc = torch.zeros((n, h, w), dtype=torch.int64, device=device)
    stream = 1
    for block, weight in ((32, 96), (16, 64), (8, 56)):
        gh, gw = (h + block - 1) // block + 1, (w + block - 1) // block + 1
        g = _t_top_bits(_t_hash(seeds, gh * gw, stream), 8).reshape(n, gh, gw)
        stream += 1
        oy = _t_umod(_t_hash(seeds, 1, 100 + stream)[:, 0], block)
        ox = _t_umod(_t_hash(seeds, 1, 200 + stream)[:, 0], block)
        yi = (ys[None, :] + oy[:, None]) // block
        xi = (xs[None, :] + ox[:, None]) // block
        acc += weight * g[bidx, yi[:, :, None], xi[:, None, :]]
    img = acc // 256
    r = _t_hash(seeds, rects * 5, 50)
    Y = ys[None, :, None]
    X = xs[None, None, :]
    for i in range(rects):
        rw = 8 + _t_umod(r[:, 5 * i], max(8, w // 6))
        rh = 8 + _t_umod(r[:, 5 * i + 1], max(8, h // 6))
        x0 = _t_umod_vec(r[:, 5 * i + 2], torch.clamp(w - rw, min=1))
        y0 = _t_umod_vec(r[:, 5 * i + 3], torch.clamp(h - rh, min=1))
        val = _t_top_bits(r[:, 5 * i + 4], 8)
        m = ((X >= x0[:, None, None]) & (X < (x0 + rw)[:, None, None]) &
             (Y >= y0[:, None, None]) & (Y < (y0 + rh)[:, None, None]))
        img = torch.where(m, val[:, None, None], img)
    p = F.pad(img.to(torch.float32)[:, None], (1, 1, 1, 1), mode="replicate")[:, 0]
    s = sum(p[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    img = (s.to(torch.int64) + 4) // 9
    fine = _t_top_bits(_t_hash(seeds, w * h, 60), 3).reshape(n, h, w)
    img = torch.clamp(img + fine - 3, 0, 255)
    return img.to(torch.uint8)


def _t_umod_vec(v, m):
    """(uint64 view of v) % m with a per-row positive modulus tensor m."""
    hi = ((v >> 1) & 0x7FFFFFFFFFFFFFFF)
    return ((hi % m) * 2 + (v & 1)) % m
