"""ctypes binding of liborbx.so (the C ABI declared in include/orbx.h).

The library is the product: there is no fallback.  If it is missing, or a call needs a GPU
and none is usable, the call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbx.so")

NSTAGES = 6
OK, E_ARG, E_HIP, E_CAPACITY, E_GEOMETRY, E_NODEVICE, E_STATE, E_RCCL = 0, -1, -2, -3, -4, -5, -6, -7


class OrbxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().orbx_strerror(code).decode()
        if code == E_HIP:
            msg += " (hipError %d)" % lib().orbx_last_hip_error()
        if code == E_RCCL:
            msg += " (%s)" % lib().orbx_sharded_last_error().decode()
        super().__init__("%s: %s [%d]" % (where, msg, code))


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("blur_variant", C.c_int),
                ("max_width", C.c_int), ("max_height", C.c_int), ("max_batch", C.c_int), ("device", C.c_int)]


# numpy view of orbx_keypoint / cv::KeyPoint (28 bytes)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

_P = C.c_void_p
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int)
_f32p = C.POINTER(C.c_float)
_u16p = C.POINTER(C.c_uint16)

# name -> (restype, argtypes); every symbol declared in include/orbx.h plus the debug entries
SIGNATURES = {
    "orbx_strerror": (C.c_char_p, [C.c_int]),
    "orbx_last_hip_error": (C.c_int, []),
    "orbx_version": (C.c_int, []),
    "orbx_extractor_create": (C.c_int, [C.POINTER(Params), C.POINTER(_P)]),
    "orbx_extractor_destroy": (None, [_P]),
    "orbx_extractor_set_stream": (C.c_int, [_P, _P]),
    "orbx_pipeline_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "orbx_pipeline_destroy": (None, [_P]),
    "orbx_extractor_set_pipeline": (C.c_int, [_P, _P]),
    "orbx_extractor_levels": (C.c_int, [_P]),
    "orbx_extractor_scale_tables": (C.c_int, [_P, _f32p, _f32p, _f32p, _f32p]),
    "orbx_extractor_features_per_level": (C.c_int, [_P, _i32p]),
    "orbx_extractor_max_keypoints": (C.c_int, [_P]),
    "orbx_extract": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _i32p]),
    "orbx_extract_batch_device": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "orbx_extract_batch_host": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int]),
    "orbx_batch_counts": (C.c_int, [_P, _i32p, _i32p]),
    "orbx_batch_download": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _i32p]),
    "orbx_batch_download_all": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "orbx_batch_device_results": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _i32p]),
    "orbx_stage_name": (C.c_char_p, [C.c_int]),
    "orbx_extractor_profile": (C.c_int, [_P, C.c_int]),
    "orbx_extractor_stage_times": (C.c_int, [_P, _f32p, _i32p]),
    "orbx_pyramid_level": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _i32p, _i32p]),
    "orbx_pyramid_layout": (C.c_int, [_P, C.c_int, _P, _P, _P, C.POINTER(C.c_int64)]),
    "orbx_pyramid_download": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64]),
    "orbx_debug_level_map": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "orbx_debug_candidates": (C.c_int, [_P, C.c_int, C.c_int, _i32p, _i32p, _i32p, C.c_int, _i32p]),
    "orbx_debug_octree": (C.c_int, [C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p,
                                    C.c_int, _i32p]),
    "orbx_descriptor_distance": (C.c_int, [_P, _P]),
    "orbx_matcher_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "orbx_matcher_destroy": (None, [_P]),
    "orbx_matcher_set_stream": (C.c_int, [_P, _P]),
    "orbx_match_top2_device": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.c_int64, _P, _P, _P]),
    "orbx_match_top2_device_noalloc": (C.c_int, [C.c_int, _P, _P, C.c_int, _P, C.c_int64, C.c_int64, _P, _P, _P]),
    "orbx_match_top2": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.c_int64, _P, _P, _P]),
    "orbx_top2_merge": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "orbx_top2_merge_shards_device": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, _P, _P, _P]),
    "orbx_sharded_unique_id": (C.c_int, [_P]),
    "orbx_sharded_matcher_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "orbx_sharded_matcher_destroy": (None, [_P]),
    "orbx_sharded_matcher_set_stream": (C.c_int, [_P, _P]),
    "orbx_sharded_matcher_synchronize": (C.c_int, [_P]),
    "orbx_match_top2_sharded_device": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.c_int64, _P, _P, _P]),
    "orbx_sharded_last_error": (C.c_char_p, []),
    "orbx_ratio_test": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_float, _P]),
    "orbx_search_for_initialization": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, _P, C.c_int, C.c_float, C.c_int, _P,
                                                 C.POINTER(C.c_int64)]),
    "orbx_search_for_initialization_batch_device": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float,
                                                              C.c_float, C.c_float, C.c_float, _P, _P, C.c_int,
                                                              C.c_float, C.c_int, _P, _P, _P]),
    "orbx_compute_stereo_matches_device": (C.c_int, [_P, _P, C.c_float, C.c_float, _P, _P, _P]),
    "orbx_compute_stereo_matches": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float, _P, _P, C.c_int,
                                              C.POINTER(C.c_int)]),
    "orbx_frame_grid_build_device": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "orbx_features_in_area_device": (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P,
                                               _P, _P, C.c_int, C.c_int, _P, _P]),
    "orbx_features_in_area": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P,
                                        _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "orbx_search_by_projection": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P,
                                            _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                            _P, _P, _P]),
    "orbx_undistort_keypoints_device": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "orbx_undistort_points": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "orbx_search_by_bow": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P,
                                     C.c_int, C.c_int, C.c_float, C.c_int, _P, _P, _P]),
    "orbx_search_by_bow_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P,
                                            _P, C.c_int, C.c_float, C.c_int, _P, _P, _P]),
    "orbx_feature_vectors_build_device": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "orbx_vocabulary_create": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "orbx_vocabulary_destroy": (None, [_P]),
    "orbx_vocabulary_transform_device": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "orbx_vocabulary_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "orbx_distinctive_descriptors_device": (C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    "orbx_distinctive_descriptors": (C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    "orbx_synth_image": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "orbx_synth_descriptors": (C.c_int, [C.c_uint64, C.c_int64, _P]),
}

_lib = None


def _preload_shared_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not both
    see the GPU, so when torch is installed its copy is loaded first and liborbx.so (which needs the
    same soname) binds to it, whichever of the two packages is imported first."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Loads liborbx.so once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("liborbx.so not found at %s: build it with `make -C orbslam2_amd/csrc` "
                               "(or __graft_entry__.build())" % LIB_PATH)
        _preload_shared_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code, where):
    if code < 0:
        raise OrbxError(code, where)
    return code


def ptr(a):
    """Data pointer of a C-contiguous numpy array as a void pointer."""
    return C.c_void_p(a.ctypes.data)
