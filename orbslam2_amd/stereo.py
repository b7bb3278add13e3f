"""Frame::ComputeStereoMatches (src/Frame.cc:534-708 of the reference) on the batches two extractors
hold on the GPU."""
import ctypes as C

import numpy as np

from . import _lib as L


def ComputeStereoMatches(left, right, mb, mbf, frame=0):
    """mvuRight / mvDepth of one frame of the last batches of `left` and `right` (ORBextractor
    instances for the left and the right image).  Returns (nmatches, uright, depth), the arrays with
    one entry per left keypoint (-1 where unmatched)."""
    cap = left.max_keypoints
    ur = np.zeros(cap, np.float32)
    dp = np.zeros(cap, np.float32)
    n = C.c_int(0)
    nm = L.check(L.lib().orbx_compute_stereo_matches(left._h, right._h, frame, mb, mbf, L.ptr(ur), L.ptr(dp), cap,
                                                     C.byref(n)), "orbx_compute_stereo_matches")
    return nm, ur[:n.value].copy(), dp[:n.value].copy()


def compute_stereo_matches_device(left, right, mb, mbf, d_uright, d_depth, d_nmatches):
    """Whole batch, results left in device memory (addresses as ints): batch x max_keypoints floats
    each for uright and depth, batch ints for the match counts.  Asynchronous on the left stream."""
    L.check(L.lib().orbx_compute_stereo_matches_device(left._h, right._h, mb, mbf, d_uright, d_depth, d_nmatches),
            "orbx_compute_stereo_matches_device")
