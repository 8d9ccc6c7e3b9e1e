"""Loader for the HIP library (slam_amd/liborbx.so) and its ctypes signatures.

There is no CPU fallback: if the library is missing, or no MI355X-class HIP
device is present, loading or handle creation raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbx.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

MEM_HOST, MEM_DEVICE = 0, 1
STAGES = ("pyramid", "fast", "quadtree", "blur", "describe")


class OrbxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
                ("blur_variant", C.c_int32), ("fma_variant", C.c_int32)]


class ProjQueries(C.Structure):
    _fields_ = [("n", C.c_int32), ("valid", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("radius", C.c_void_p), ("min_level", C.c_void_p), ("max_level", C.c_void_p),
                ("desc", C.c_void_p), ("angle", C.c_void_p), ("has_obs", C.c_void_p),
                ("u_right", C.c_void_p), ("right_gate", C.c_void_p)]


class ProjPolicy(C.Structure):
    _fields_ = [("claim_rule", C.c_int32), ("max_dist", C.c_int32), ("ratio_test", C.c_int32),
                ("nn_ratio", C.c_float), ("check_orientation", C.c_int32)]


class MFrame(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32), ("min_x", C.c_float),
                ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


MAX_LEVELS = 16


class Pyramid(C.Structure):
    _fields_ = [("nlevels", C.c_int32), ("level", C.c_void_p * MAX_LEVELS), ("width", C.c_int32 * MAX_LEVELS),
                ("height", C.c_int32 * MAX_LEVELS), ("pitch", C.c_size_t * MAX_LEVELS)]


class ProjectionPair(C.Structure):
    _fields_ = [("cur", MFrame), ("last", MFrame), ("mp_desc", C.c_void_p), ("valid", C.c_void_p), ("u", C.c_void_p),
                ("v", C.c_void_p), ("invz", C.c_void_p), ("has_obs", C.c_void_p), ("cur_u_right", C.c_void_p),
                ("direction", C.c_int32), ("cur_claimed", C.c_void_p), ("cur_claimed_obs", C.c_void_p),
                ("query_best", C.c_void_p), ("nmatches", C.c_int32)]


class StereoJob(C.Structure):
    _fields_ = [("left", MFrame), ("right", MFrame), ("pyr_left", Pyramid), ("pyr_right", Pyramid),
                ("u_right", C.c_void_p), ("depth", C.c_void_p), ("n_stereo", C.c_int32)]


class ProjectedJob(C.Structure):
    _fields_ = [("train", MFrame), ("train_u_right", C.c_void_p), ("q", ProjQueries), ("claimed", C.c_void_p),
                ("claimed_obs", C.c_void_p), ("query_best", C.c_void_p), ("nmatches", C.c_int32)]


class Vocabulary(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_children", C.c_int32), ("levels", C.c_int32), ("node_desc", C.c_void_p),
                ("child_start", C.c_void_p), ("children", C.c_void_p), ("word_id", C.c_void_p), ("weight", C.c_void_p)]


class FeatureVector(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_items", C.c_int32), ("node_ids", C.c_void_p), ("start", C.c_void_p),
                ("items", C.c_void_p)]


class BowJob(C.Structure):
    _fields_ = [("f1", MFrame), ("f2", MFrame), ("fv1", FeatureVector), ("fv2", FeatureVector), ("valid1", C.c_void_p),
                ("valid2", C.c_void_p), ("match12", C.c_void_p), ("match21", C.c_void_p), ("nmatches", C.c_int32)]


class InitJob(C.Structure):
    _fields_ = [("f1", MFrame), ("f2", MFrame), ("prev_matched_xy", C.c_void_p), ("matches12", C.c_void_p),
                ("nmatches", C.c_int32)]


# Every exported symbol of include/orbx.h: (restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "orbx_last_error": (C.c_char_p, []),
    "orbx_version": (C.c_char_p, []),
    "orbx_create": (_i, [C.POINTER(Config), C.POINTER(_vp)]),
    "orbx_destroy": (None, [_vp]),
    "orbx_set_stream": (_i, [_vp, _vp]),
    "orbx_reset_stream": (_i, [_vp]),
    "orbx_synchronize": (_i, [_vp]),
    "orbx_set_overlap": (_i, [_vp, _i]),
    "orbx_max_keypoints": (_i, [_vp]),
    "orbx_extract": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _i, C.POINTER(_i)]),
    "orbx_extract_batch": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _vp, _i, _vp, _i]),
    "orbx_register_host": (_i, [_vp, _sz]),
    "orbx_unregister_host": (_i, [_vp]),
    "orbx_get_level": (_i, [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i),
                            C.POINTER(_sz)]),
    "orbx_levels": (_i, [_vp]),
    "orbx_scale_factor": (_f, [_vp]),
    "orbx_scale_factors": (C.POINTER(_f), [_vp]),
    "orbx_inv_scale_factors": (C.POINTER(_f), [_vp]),
    "orbx_level_sigma2": (C.POINTER(_f), [_vp]),
    "orbx_inv_level_sigma2": (C.POINTER(_f), [_vp]),
    "orbx_features_per_level": (C.POINTER(C.c_int32), [_vp]),
    "orbx_profile_enable": (_i, [_vp, _i]),
    "orbx_profile_read": (_i, [_vp, _vp, _vp, _i]),
    "orbx_profile_reset": (_i, [_vp]),
    "orbx_debug_sincos": (_i, [_vp, _vp, _i, _vp, _vp]),
    "orbx_debug_tables": (_i, [_vp, _vp, _i]),
    "orbx_debug_atan2": (_i, [_vp, _vp, _vp, _i, _vp]),
    "orbx_debug_level": (_i, [_vp, _i, _i, _vp]),
    "orbx_debug_blurred": (_i, [_vp, _i, _i, _vp, C.POINTER(_i), C.POINTER(_i)]),
    "orbx_debug_candidates": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, C.POINTER(_i)]),
    "orbx_debug_selected": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, C.POINTER(_i)]),
    "orbm_hamming256": (_i, [_vp, _vp]),
    "orbm_create": (_i, [_i, C.POINTER(_vp)]),
    "orbm_destroy": (None, [_vp]),
    "orbm_set_stream": (_i, [_vp, _vp]),
    "orbm_reset_stream": (_i, [_vp]),
    "orbm_synchronize": (_i, [_vp]),
    "orbm_grid_build": (_i, [_vp, C.POINTER(MFrame), _vp, _vp, _i]),
    "orbm_window_candidates": (_i, [_vp, C.POINTER(MFrame), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                    _vp, _vp, _vp, _i]),
    "orbm_search_for_initialization": (_i, [_vp, C.POINTER(MFrame), C.POINTER(MFrame), _vp, _vp, _i,
                                            _f, _i, C.POINTER(_i), _i]),
    "orbm_search_projected": (_i, [_vp, C.POINTER(MFrame), _vp, C.POINTER(ProjQueries),
                                   C.POINTER(ProjPolicy), _vp, _vp, _vp, C.POINTER(_i), _i]),
    "orbm_search_by_projection": (_i, [_vp, C.POINTER(MFrame), C.POINTER(MFrame), _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _i, _f, _f, _i, _i, _vp, _vp, _vp,
                                       C.POINTER(_i), _i]),
    "orbm_search_by_bow": (_i, [_vp, C.POINTER(MFrame), C.POINTER(FeatureVector), _vp, C.POINTER(MFrame),
                                C.POINTER(FeatureVector), _vp, _i, _f, _i, _vp, _vp, _vp, _i]),
    "orbm_search_for_triangulation": (_i, [_vp, C.POINTER(MFrame), C.POINTER(FeatureVector), _vp, _vp, C.POINTER(MFrame),
                                           C.POINTER(FeatureVector), _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _vp,
                                           _vp, _i]),
    "orbm_window_best": (_i, [_vp, C.POINTER(MFrame), _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp,
                              _i]),
    "orbm_search_by_sim3": (_i, [_vp, C.POINTER(MFrame), C.POINTER(MFrame), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _vp, _vp, _i]),
    "orbm_filter_by_mask": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _sz, _vp, _vp, _vp, _i]),
    "orbm_pack_keyframe_records": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i]),
    "orbm_unpack_keyframe_records": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i]),
    "orbm_distinctive_descriptors": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i]),
    "orbm_compute_stereo_matches": (_i, [_vp, C.POINTER(MFrame), C.POINTER(MFrame), C.POINTER(Pyramid),
                                         C.POINTER(Pyramid), _vp, _vp, _f, _f, _vp, _vp, _vp, _i]),
    "orbm_compute_stereo_matches_batch": (_i, [_vp, C.POINTER(StereoJob), _i, _vp, _vp, _i, _f, _f]),
    "orbm_search_by_projection_batch": (_i, [_vp, C.POINTER(ProjectionPair), _i, _vp, _i, _f, _f, _i]),
    "orbm_search_by_bow_batch": (_i, [_vp, C.POINTER(BowJob), _i, _i, _f, _i]),
    "orbm_search_for_initialization_batch": (_i, [_vp, C.POINTER(InitJob), _i, _i, _f, _i]),
    "orbm_search_projected_batch": (_i, [_vp, C.POINTER(ProjectedJob), _i, C.POINTER(ProjPolicy), _f]),
    "orbm_knn2": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i]),
    "orbm_bow_transform": (_i, [_vp, C.POINTER(Vocabulary), _vp, _i, _i, _vp, _vp, _vp, _i]),
    "orbm_ratio_test": (_i, [_vp, _i, _vp, _vp, _f, _i, _vp, _i]),
    "orbm_debug_knn_path": (_i, [_vp, _i]),
    "orbm_knn2_noalloc": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "orbm_knn2_merge": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
}

_lib = None


def load():
    """Load liborbx.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbxError("HIP library %s not built: run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (there is no CPU fallback)" % LIB_PATH)
    # A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own copy of
    # libamdhip64 (same soname as the system one); importing torch first makes the
    # loader resolve this library's HIP dependency to that already-loaded copy, so
    # torch tensors and these kernels share one runtime and one device context.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported: no partial library
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().orbx_last_error()
        raise OrbxError("orbx status %d: %s" % (rc, msg.decode() if msg else ""))
