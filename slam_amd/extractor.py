"""Host-side mirror of ORB_SLAM2::ORBextractor over the orbx C ABI.

Same constructor arguments, call signature and getters as the reference class
(reference include/ORBextractor.h:32-105); the work runs as HIP kernels on an
MI355X.  Keypoints come back as a structured array laid out like cv::KeyPoint
and descriptors as an (n, 32) uint8 array, the two outputs of
ORBextractor::operator() (reference src/ORBextractor.cc:1052-1114).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, MEM_DEVICE, MEM_HOST, STAGES, Config, OrbxError, check


class ORBextractor:
    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width=640,
                 max_height=480, max_batch=1, device=0, blur_variant=0, fma_variant=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        cfg = Config(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height,
                     max_batch, device, blur_variant, fma_variant)
        self._cfg = cfg
        check(self._lib.orbx_create(C.byref(cfg), C.byref(self._h)))
        self.nlevels = nlevels
        self.max_batch = max_batch
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.orbx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference getters (include/ORBextractor.h:51-71)
    def GetLevels(self):
        return self._lib.orbx_levels(self._h)

    def GetScaleFactor(self):
        return self._lib.orbx_scale_factor(self._h)

    def _arr(self, fn, dtype=np.float32):
        p = fn(self._h)
        return np.ctypeslib.as_array(p, shape=(self.nlevels,)).astype(dtype).copy()

    def GetScaleFactors(self):
        return self._arr(self._lib.orbx_scale_factors)

    def GetInverseScaleFactors(self):
        return self._arr(self._lib.orbx_inv_scale_factors)

    def GetScaleSigmaSquares(self):
        return self._arr(self._lib.orbx_level_sigma2)

    def GetInverseScaleSigmaSquares(self):
        return self._arr(self._lib.orbx_inv_level_sigma2)

    def features_per_level(self):
        return self._arr(self._lib.orbx_features_per_level, np.int32)

    def max_keypoints(self):
        return self._lib.orbx_max_keypoints(self._h)

    # ---- operator()
    def __call__(self, image, mask=None):
        """(keypoints, descriptors) of one 8-bit image.  `mask` is ignored, as in
        the reference (include/ORBextractor.h:45)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise ValueError("image must be a 2-D uint8 array (CV_8UC1)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = max(self.max_keypoints(), 1)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        check(self._lib.orbx_extract(self._h, image.ctypes.data, w, h, image.strides[0],
                                     kps.ctypes.data, desc.ctypes.data, cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images, cap=None):
        """Host batch: images (B, H, W) uint8 -> (kps (B, cap), desc (B, cap, 32), n (B,))."""
        images = np.ascontiguousarray(images, np.uint8)
        b, h, w = images.shape
        cap = cap or max(self.max_keypoints(), 1)
        kps = np.zeros((b, cap), KP_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        check(self._lib.orbx_extract_batch(self._h, images.ctypes.data, b, w, h, images.strides[1],
                                           images.strides[0], kps.ctypes.data, desc.ctypes.data,
                                           cap, n.ctypes.data, MEM_HOST))
        return kps, desc, n

    def extract_batch_device(self, images, kps, desc, n_out, cap):
        """Device batch on torch tensors: images (B, H, W) uint8 (last dim contiguous), kps a
        uint8 tensor of B*cap*28 bytes, desc B*cap*32 bytes, n_out int32 (B,).  Only enqueues."""
        b, h, w = images.shape
        assert images.stride(2) == 1
        check(self._lib.orbx_extract_batch(self._h, images.data_ptr(), b, w, h, images.stride(1),
                                           images.stride(0), kps.data_ptr(), desc.data_ptr(), cap,
                                           n_out.data_ptr(), MEM_DEVICE))

    def set_stream(self, stream_ptr):
        """Run on the hipStream_t with this handle value (0 / None = the legacy default stream)."""
        check(self._lib.orbx_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def reset_stream(self):
        check(self._lib.orbx_reset_stream(self._h))

    def set_overlap(self, on):
        check(self._lib.orbx_set_overlap(self._h, int(bool(on))))

    def synchronize(self):
        check(self._lib.orbx_synchronize(self._h))

    # ---- mvImagePyramid (include/ORBextractor.h:73)
    def level_info(self, frame, level):
        ptr, w, h, st = C.c_void_p(), C.c_int(), C.c_int(), C.c_size_t()
        check(self._lib.orbx_get_level(self._h, frame, level, C.byref(ptr), C.byref(w), C.byref(h),
                                       C.byref(st)))
        return ptr.value, w.value, h.value, st.value

    def image_pyramid_level(self, frame, level):
        """Copy of mvImagePyramid[level] for `frame` of the last call, as a numpy array."""
        _, w, h, _ = self.level_info(frame, level)
        out = np.zeros((h, w), np.uint8)
        check(self._lib.orbx_debug_level(self._h, frame, level, out.ctypes.data))
        return out

    # ---- profiling and debug taps
    def profile_enable(self, on=True):
        check(self._lib.orbx_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(self._lib.orbx_profile_reset(self._h))

    def profile_read(self):
        ms = (C.c_float * len(STAGES))()
        ln = (C.c_int * len(STAGES))()
        check(self._lib.orbx_profile_read(self._h, ms, ln, len(STAGES)))
        return {s: (ms[i], ln[i]) for i, s in enumerate(STAGES)}

    def debug_sincos(self, angles):
        a = np.ascontiguousarray(angles, np.float32)
        sn, cs = np.zeros_like(a), np.zeros_like(a)
        check(self._lib.orbx_debug_sincos(self._h, a.ctypes.data, len(a), sn.ctypes.data, cs.ctypes.data))
        return sn, cs

    def debug_atan2(self, y, x):
        y = np.ascontiguousarray(y, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(y)
        check(self._lib.orbx_debug_atan2(self._h, y.ctypes.data, x.ctypes.data, len(y), out.ctypes.data))
        return out

    def debug_tables(self):
        out = np.zeros(752, np.uint32)
        check(self._lib.orbx_debug_tables(self._h, out.ctypes.data, len(out)))
        return out

    def debug_blurred(self, frame, level):
        _, w, h, _ = self.level_info(frame, level)
        out = np.zeros((h, w), np.uint8)
        check(self._lib.orbx_debug_blurred(self._h, frame, level, out.ctypes.data, None, None))
        return out

    def _debug_list(self, fn, frame, level):
        n = C.c_int(0)
        cap = 1 << 16
        while True:
            xs, ys, rs = (np.zeros(cap, np.int32) for _ in range(3))
            check(fn(self._h, frame, level, xs.ctypes.data, ys.ctypes.data, rs.ctypes.data, cap,
                     C.byref(n)))
            if n.value <= cap:
                return xs[:n.value].copy(), ys[:n.value].copy(), rs[:n.value].copy()
            cap = n.value

    def debug_candidates(self, frame, level):
        return self._debug_list(self._lib.orbx_debug_candidates, frame, level)

    def debug_selected(self, frame, level):
        return self._debug_list(self._lib.orbx_debug_selected, frame, level)
