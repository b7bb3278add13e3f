"""ctypes binding of oracle/liborb_oracle.so, the CPU restatement of the reference's ORB front end.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  The product package (orbslam2_amd) never imports this module.  Parity unpinned: see the
header of orb_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MAX_LEVELS = 16


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("blur_variant", C.c_int)]


class Trace(C.Structure):
    _fields_ = [("nlevels", C.c_int), ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS),
                ("img", C.POINTER(C.c_uint8) * MAX_LEVELS), ("blur", C.POINTER(C.c_uint8) * MAX_LEVELS),
                ("ncand", C.c_int * MAX_LEVELS), ("nkeep", C.c_int * MAX_LEVELS)]


_lib = None
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        h = C.CDLL(LIB_PATH)
        h.orc_fast_atan2.restype = C.c_float
        h.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        h.orc_ic_angle.restype = C.c_float
        h.orc_ic_angle.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip]
        h.orc_brief_descriptor.restype = None
        h.orc_brief_descriptor.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp]
        h.orc_extract.argtypes = [C.POINTER(Params), _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _ip,
                                  C.POINTER(Trace)]
        h.orc_trace_free.argtypes = [C.POINTER(Trace)]
        h.orc_trace_free.restype = None
        h.orc_match_top2.restype = None
        h.orc_match_top2.argtypes = [_vp, C.c_int, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp]
        h.orc_search_for_initialization.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_float, C.c_float,
                                                    C.c_float, C.c_float, _vp, C.c_int, C.c_float, C.c_int, _vp,
                                                    C.POINTER(C.c_int64)]
        h.orc_scale_tables.restype = None
        h.orc_scale_tables.argtypes = [C.c_float, C.c_int, _fp, _fp, _fp, _fp]
        h.orc_features_per_level.restype = None
        h.orc_features_per_level.argtypes = [C.c_int, C.c_float, C.c_int, _ip]
        h.orc_umax.restype = None
        h.orc_umax.argtypes = [_ip]
        h.orc_resize_linear_u8.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int]
        h.orc_gauss7_u8.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int]
        h.orc_fast_arc_score.argtypes = [_vp, C.c_int]
        h.orc_fast_detect.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, C.c_int]
        h.orc_level_candidates.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, C.c_int]
        h.orc_distribute_octree.argtypes = [_ip, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip,
                                            C.c_int]
        h.orc_descriptor_distance.argtypes = [_vp, _vp]
        h.orc_ratio_accept.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
        h.orc_three_maxima.restype = None
        h.orc_three_maxima.argtypes = [_ip, C.c_int, _ip, _ip, _ip]
        _lib = h
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _i(a):
    return a.ctypes.data_as(_ip)


def scale_tables(scale_factor, nlevels):
    t = [np.zeros(nlevels, np.float32) for _ in range(4)]
    lib().orc_scale_tables(scale_factor, nlevels, *[a.ctypes.data_as(_fp) for a in t])
    return t


def features_per_level(nfeatures, scale_factor, nlevels):
    q = np.zeros(nlevels, np.int32)
    lib().orc_features_per_level(nfeatures, scale_factor, nlevels, _i(q))
    return q


def umax():
    u = np.zeros(17, np.int32)
    lib().orc_umax(_i(u))
    return u[:16]


def resize(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    rc = lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    assert rc == 0
    return dst


def gauss7(src, variant=0):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    rc = lib().orc_gauss7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0], variant)
    assert rc == 0
    return dst


def fast_arc_score(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_fast_arc_score(C.c_void_p(img.ctypes.data + y * img.strides[0] + x), img.strides[0])


def fast_detect(win, threshold, nonmax=True):
    win = np.ascontiguousarray(win, np.uint8)
    h, w = win.shape
    cap = w * h
    xs, ys, rs = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_fast_detect(_p(win), w, h, win.strides[0], threshold, int(nonmax), _i(xs), _i(ys), _i(rs), cap)
    return xs[:n].copy(), ys[:n].copy(), rs[:n].copy()


def level_candidates(img, ini_th, min_th):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    xs, ys, rs = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_level_candidates(_p(img), w, h, img.strides[0], ini_th, min_th, _i(xs), _i(ys), _i(rs), cap)
    assert n <= cap
    return xs[:n].copy(), ys[:n].copy(), rs[:n].copy()


def distribute_octree(xs, ys, resp, min_x, max_x, min_y, max_y, quota):
    xs = np.ascontiguousarray(xs, np.int32)
    ys = np.ascontiguousarray(ys, np.int32)
    resp = np.ascontiguousarray(resp, np.int32)
    cap = len(xs) + 8
    out = np.zeros(cap, np.int32)
    n = lib().orc_distribute_octree(_i(xs), _i(ys), _i(resp), len(xs), min_x, max_x, min_y, max_y, quota, _i(out),
                                    cap)
    if n < 0:
        raise ValueError("orc_distribute_octree: %d" % n)
    return out[:n].copy()


def fast_atan2(y, x):
    return lib().orc_fast_atan2(y, x)


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    u = np.zeros(17, np.int32)
    lib().orc_umax(_i(u))
    m10, m01 = C.c_int(0), C.c_int(0)
    a = lib().orc_ic_angle(_p(img), img.strides[0], x, y, _i(u), C.byref(m10), C.byref(m01))
    return a, m10.value, m01.value


def brief_descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_brief_descriptor(_p(blurred), blurred.strides[0], x, y, angle_deg, _p(d))
    return d


def extract(image, nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant=0, trace=False):
    """Returns (keypoints, descriptors[, trace dict]).  Empty image -> (None, None)."""
    if image is None or image.size == 0:
        return (None, None, None) if trace else (None, None)
    image = np.ascontiguousarray(image, np.uint8)
    h, w = image.shape
    p = Params(nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant)
    cap = nfeatures + 4 * nlevels + 64
    while True:
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        tr = Trace()
        rc = lib().orc_extract(C.byref(p), _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap, C.byref(n),
                               C.byref(tr) if trace else None)
        if rc != 0:
            raise ValueError("orc_extract: %d" % rc)
        if n.value <= cap:
            break
        if trace:
            lib().orc_trace_free(C.byref(tr))
        cap = n.value
    if not trace:
        return kps[:n.value].copy(), desc[:n.value].copy()
    info = {"levels": [], "blur": [], "ncand": list(tr.ncand[:nlevels]), "nkeep": list(tr.nkeep[:nlevels])}
    for l in range(nlevels):
        lw, lh = tr.w[l], tr.h[l]
        info["levels"].append(np.ctypeslib.as_array(tr.img[l], shape=(lh, lw)).copy())
        info["blur"].append(np.ctypeslib.as_array(tr.blur[l], shape=(lh, lw)).copy() if tr.blur[l] else None)
    lib().orc_trace_free(C.byref(tr))
    return kps[:n.value].copy(), desc[:n.value].copy(), info


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def match_top2(q, db, idx_base=0):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    db = np.ascontiguousarray(db, np.uint8).reshape(-1, 32)
    nq = q.shape[0]
    best = np.zeros(nq, np.int32)
    d1 = np.zeros(nq, np.uint16)
    d2 = np.zeros(nq, np.uint16)
    lib().orc_match_top2(_p(q), nq, _p(db), db.shape[0], idx_base, _p(best), _p(d1), _p(d2))
    return best, d1, d2


def search_for_initialization(k1, d1, k2, d2, bounds, prev_matched, window, nnratio, check_ori):
    k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE)
    k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    m12 = np.full(len(k1), -1, np.int32)
    pairs = C.c_int64(0)
    n = lib().orc_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), bounds[0], bounds[1],
                                            bounds[2], bounds[3], _p(prev_matched), window, nnratio, int(check_ori),
                                            _p(m12), C.byref(pairs))
    return n, m12, pairs.value


def compute_stereo_matches(kL, dL, kR, dR, pyrL, pyrR, sf, inv, mb, mbf):
    """Frame::ComputeStereoMatches on the oracle's data: pyrL/pyrR are lists of level images.
    Returns (n, uright, depth, sad)."""
    kL = np.ascontiguousarray(kL, KEYPOINT_DTYPE)
    kR = np.ascontiguousarray(kR, KEYPOINT_DTYPE)
    dL = np.ascontiguousarray(dL, np.uint8)
    dR = np.ascontiguousarray(dR, np.uint8)
    nl = len(pyrL)
    pl = [np.ascontiguousarray(a, np.uint8) for a in pyrL]
    pr = [np.ascontiguousarray(a, np.uint8) for a in pyrR]
    PT = C.c_void_p * nl
    IT = C.c_int * nl
    lw = IT(*[a.shape[1] for a in pl])
    lh = IT(*[a.shape[0] for a in pl])
    ls_l = IT(*[a.strides[0] for a in pl])
    ls_r = IT(*[a.strides[0] for a in pr])
    sf = np.ascontiguousarray(sf, np.float32)
    inv = np.ascontiguousarray(inv, np.float32)
    ur = np.zeros(len(kL), np.float32)
    dp = np.zeros(len(kL), np.float32)
    sad = np.zeros(len(kL), np.int32)
    fn = lib().orc_compute_stereo_matches
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    n = fn(_p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR), PT(*[a.ctypes.data for a in pl]),
           PT(*[a.ctypes.data for a in pr]), lw, lh, ls_l, ls_r, nl, _p(sf), _p(inv), mb, mbf, _p(ur), _p(dp), _p(sad))
    return n, ur, dp, sad


def distinctive_descriptor(desc):
    """MapPoint::ComputeDistinctiveDescriptors on an (n, 32) uint8 array: index of the chosen row."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    fn = lib().orc_distinctive_descriptor
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int]
    return fn(_p(desc), desc.shape[0])


def search_by_bow(d1, ang1, valid1, nodes1, d2, ang2, valid2, nodes2, strict_lt, ratio, check_ori):
    """ORBmatcher::SearchByBoW.  nodes*: (ids, start, items) int32 arrays of a feature vector.
    Returns (nmatches, match12, match21)."""
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
    ang1 = np.ascontiguousarray(ang1, np.float32)
    ang2 = np.ascontiguousarray(ang2, np.float32)
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    i1, s1, t1 = (np.ascontiguousarray(a, np.int32) for a in nodes1)
    i2, s2, t2 = (np.ascontiguousarray(a, np.int32) for a in nodes2)
    m12 = np.zeros(len(d1), np.int32)
    m21 = np.zeros(len(d2), np.int32)
    fn = lib().orc_search_by_bow
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                   C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    n = fn(_p(d1), _p(ang1), None if v1 is None else _p(v1), len(d1), _p(i1), _p(s1), _p(t1), len(i1), _p(d2),
           _p(ang2), None if v2 is None else _p(v2), len(d2), _p(i2), _p(s2), _p(t2), len(i2), int(strict_lt), ratio,
           int(check_ori), _p(m12), _p(m21))
    return n, m12, m21


def grid_queries(kps, bounds, x, y, r, min_level, max_level, cap=512):
    """Frame::AssignFeaturesToGrid + GetFeaturesInArea: (list of index arrays, cell_start, cell_items)."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (x, y, r))
    lo, hi = (np.ascontiguousarray(a, np.int32) for a in (min_level, max_level))
    nq = len(qx)
    out = np.zeros((max(nq, 1), cap), np.int32)
    cnt = np.zeros(max(nq, 1), np.int32)
    cs = np.zeros(64 * 48 + 1, np.int32)
    ci = np.zeros(max(len(kps), 1), np.int32)
    fn = lib().orc_grid_queries
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(_p(kps), len(kps), bounds[0], bounds[1], bounds[2], bounds[3], _p(qx), _p(qy), _p(qr), _p(lo), _p(hi), nq, cap,
       _p(out), _p(cnt), _p(cs), _p(ci))
    assert nq == 0 or cnt[:nq].max() <= cap
    return [out[q, :cnt[q]].copy() for q in range(nq)], cs, ci[:cs[-1]].copy()


def search_by_projection(kps, desc, bounds, x, y, r, min_level, max_level, src_desc, nnratio, *, uright=None,
                         taken=None, valid=None, xr=None, blocks=None, src_angle=None, use_second=True, th_dist=100,
                         check_orientation=False):
    """ORBmatcher::SearchByProjection, both tracking forms: (nmatches, match per source, feature_match)."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n, ns = len(kps), len(x)
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
    ur, tk, va, xr_, bl, an = f32(uright), u8(taken), u8(valid), f32(xr), u8(blocks), f32(src_angle)
    qx, qy, qr = f32(x), f32(y), f32(r)
    lo, hi = np.ascontiguousarray(min_level, np.int32), np.ascontiguousarray(max_level, np.int32)
    sd = np.ascontiguousarray(src_desc, np.uint8).reshape(-1, 32)
    match = np.full(max(ns, 1), -1, np.int32)
    fmatch = np.full(max(n, 1), -1, np.int32)
    opt = lambda a: None if a is None else _p(a)
    fn = lib().orc_search_by_projection
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    nm = fn(_p(kps), _p(desc), opt(ur), opt(tk), n, bounds[0], bounds[1], bounds[2], bounds[3], opt(va), _p(qx), _p(qy),
            _p(qr), _p(lo), _p(hi), opt(xr_), _p(sd), opt(bl), opt(an), ns, int(use_second), nnratio, th_dist,
            int(check_orientation), _p(match), _p(fmatch))
    return int(nm), match[:ns].copy(), fmatch[:n].copy()


def undistort_points(xy, cam, dist):
    """cv::undistortPoints(xy, xy, K, dist, Mat(), K) as Frame::UndistortKeyPoints calls it: (n, 2) float32."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    cam = np.ascontiguousarray(cam, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(xy)
    fn = lib().orc_undistort_points
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    fn(_p(xy), len(xy), _p(cam), _p(dist), len(dist), _p(out))
    return out


def vocabulary_transform(tree, feat, levelsup):
    """tree = dict(desc (N,32) uint8, child_start (N+1,) int32, children int32, word_id (N,) int32,
    weight (N,) float64, L).  Returns (word, node, weight) per feature."""
    feat = np.ascontiguousarray(feat, np.uint8).reshape(-1, 32)
    n = len(feat)
    word = np.zeros(max(n, 1), np.int32)
    node = np.zeros(max(n, 1), np.int32)
    wt = np.zeros(max(n, 1), np.float64)
    fn = lib().orc_vocabulary_transform
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p]
    arrs = [np.ascontiguousarray(tree["desc"], np.uint8), np.ascontiguousarray(tree["child_start"], np.int32),
            np.ascontiguousarray(tree["children"], np.int32), np.ascontiguousarray(tree["word_id"], np.int32),
            np.ascontiguousarray(tree["weight"], np.float64)]
    fn(*[_p(a) for a in arrs], tree["L"], _p(feat), n, levelsup, _p(word), _p(node), _p(wt))
    return word[:n], node[:n], wt[:n]
