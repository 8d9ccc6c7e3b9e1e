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
