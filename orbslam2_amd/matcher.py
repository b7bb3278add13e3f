"""Python mirror of the batched part of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h of the
reference) on top of the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L


def synth_descriptors(seed, n):
    out = np.zeros((n, 32), np.uint8)
    L.check(L.lib().orbx_synth_descriptors(seed, n, L.ptr(out)), "orbx_synth_descriptors")
    return out


class ORBmatcher:
    """ORBmatcher(nnratio=0.6, checkOri=True) (src/ORBmatcher.cc:41-43)."""

    TH_HIGH = 100
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, *, device=0, max_queries=4096):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._device = device
        self._max_queries = max_queries
        self._h = None
        self._lib = L.lib()

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            L.check(self._lib.orbx_matcher_create(self._device, self._max_queries, C.byref(h)), "orbx_matcher_create")
            self._h = h
        return self._h

    def close(self):
        if self._h is not None:
            self._lib.orbx_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        L.check(self._lib.orbx_matcher_set_stream(self._handle(), C.c_void_p(hip_stream)), "orbx_matcher_set_stream")

    @staticmethod
    def DescriptorDistance(a, b):
        """Hamming distance of two 32-byte descriptors (src/ORBmatcher.cc:1744-1760); host only."""
        a = np.ascontiguousarray(a, np.uint8).reshape(-1)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1)
        assert a.size == 32 and b.size == 32
        return L.lib().orbx_descriptor_distance(L.ptr(a), L.ptr(b))

    def match_top2(self, queries, db, idx_base=0):
        """Brute-force best / second-best of each query row over db (host arrays).
        Returns (best_idx int32, d1 uint16, d2 uint16)."""
        q = np.ascontiguousarray(queries, np.uint8).reshape(-1, 32)
        d = np.ascontiguousarray(db, np.uint8).reshape(-1, 32)
        nq = q.shape[0]
        best = np.zeros(nq, np.int32)
        d1 = np.zeros(nq, np.uint16)
        d2 = np.zeros(nq, np.uint16)
        L.check(self._lib.orbx_match_top2(self._handle(), L.ptr(q), nq, L.ptr(d) if d.shape[0] else None,
                                          d.shape[0], idx_base, L.ptr(best), L.ptr(d1), L.ptr(d2)),
                "orbx_match_top2")
        return best, d1, d2

    def match_top2_device(self, d_q, nq, d_db, ndb, idx_base, d_best, d_d1, d_d2):
        L.check(self._lib.orbx_match_top2_device(self._handle(), C.c_void_p(d_q), nq, C.c_void_p(d_db), ndb,
                                                 idx_base, C.c_void_p(d_best), C.c_void_p(d_d1), C.c_void_p(d_d2)),
                "orbx_match_top2_device")

    def merge_top2_shards_device(self, nshards, nq, d_blocks, shard_stride, d_best, d_d1, d_d2):
        """Merge of per-shard result blocks on the device (orbx_top2_merge_shards_device)."""
        L.check(self._lib.orbx_top2_merge_shards_device(self._handle(), nshards, nq, C.c_void_p(d_blocks), shard_stride,
                                                        C.c_void_p(d_best), C.c_void_p(d_d1), C.c_void_p(d_d2)),
                "orbx_top2_merge_shards_device")

    @staticmethod
    def match_top2_device_noalloc(device, hip_stream, d_q, nq, d_db, ndb, idx_base, d_best, d_d1, d_d2):
        """match_top2_device without a handle and without any allocation (orbx_match_top2_device_noalloc):
        the partial results live in an array inside the library's device code.  Calls on one GPU must
        not overlap."""
        L.check(L.lib().orbx_match_top2_device_noalloc(device, C.c_void_p(hip_stream), C.c_void_p(d_q), nq,
                                                       C.c_void_p(d_db), ndb, idx_base, C.c_void_p(d_best),
                                                       C.c_void_p(d_d1), C.c_void_p(d_d2)),
                "orbx_match_top2_device_noalloc")

    @staticmethod
    def merge_top2(best, d1, d2):
        """Merges per-shard results of shape (nshards, nq) (shards in ascending index order)."""
        best = np.ascontiguousarray(best, np.int32)
        d1 = np.ascontiguousarray(d1, np.uint16)
        d2 = np.ascontiguousarray(d2, np.uint16)
        ns, nq = best.shape
        ob, o1, o2 = np.zeros(nq, np.int32), np.zeros(nq, np.uint16), np.zeros(nq, np.uint16)
        L.check(L.lib().orbx_top2_merge(ns, nq, L.ptr(best), L.ptr(d1), L.ptr(d2), L.ptr(ob), L.ptr(o1), L.ptr(o2)),
                "orbx_top2_merge")
        return ob, o1, o2

    def ratio_test(self, d1, d2, th=None):
        """Acceptance of src/ORBmatcher.cc:245-249 with this matcher's ratio."""
        d1 = np.ascontiguousarray(d1, np.uint16)
        d2 = np.ascontiguousarray(d2, np.uint16)
        acc = np.zeros(d1.shape[0], np.uint8)
        n = L.check(L.lib().orbx_ratio_test(d1.shape[0], L.ptr(d1), L.ptr(d2), self.TH_LOW if th is None else th,
                                            self.mfNNratio, L.ptr(acc)), "orbx_ratio_test")
        return acc.astype(bool), n

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, bounds, vbPrevMatched, windowSize=10):
        """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:433-549).
        kps*: KEYPOINT_DTYPE arrays (the frames' mvKeysUn), desc*: (N,32) uint8, bounds = (mnMinX, mnMaxX,
        mnMinY, mnMaxY), vbPrevMatched: (N1,2) float32, updated in place.
        Returns (nmatches, vnMatches12, pair_count)."""
        k1 = np.ascontiguousarray(kps1, L.KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(kps2, L.KEYPOINT_DTYPE)
        a = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags.c_contiguous
        m12 = np.full(len(k1), -1, np.int32)
        pairs = C.c_int64(0)
        n = L.check(self._lib.orbx_search_for_initialization(
            self._handle(), L.ptr(k1), L.ptr(a), len(k1), L.ptr(k2), L.ptr(b), len(k2), bounds[0], bounds[1],
            bounds[2], bounds[3], L.ptr(vbPrevMatched), windowSize, self.mfNNratio, int(self.mbCheckOrientation),
            L.ptr(m12), C.byref(pairs)), "orbx_search_for_initialization")
        return n, m12, pairs.value

    def search_for_initialization_batch_device(self, d_keypoints, d_descriptors, d_counts, max_keypoints, nframes,
                                               max_level0, bounds, d_matches12, d_nmatches, d_pair_counts,
                                               windowSize=100, d_prev_matched=None, d_prev_matched_out=None):
        """SearchForInitialization of every consecutive frame pair of a device-resident batch (the arrays
        of ORBextractor.device_results()); all arguments are device addresses (ints).  Asynchronous on
        the matcher's stream."""
        L.check(self._lib.orbx_search_for_initialization_batch_device(
            self._handle(), d_keypoints, d_descriptors, d_counts, max_keypoints, nframes, max_level0, bounds[0],
            bounds[1], bounds[2], bounds[3], d_prev_matched, d_prev_matched_out, windowSize, self.mfNNratio,
            int(self.mbCheckOrientation), d_matches12, d_nmatches, d_pair_counts),
            "orbx_search_for_initialization_batch_device")

    def ComputeDistinctiveDescriptors(self, segments):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:248-319) for a list of map points:
        segments[p] is the (n_p, 32) uint8 array of the descriptors observed for point p.
        Returns (best index per point, -1 for an empty one; (npoints, 32) chosen descriptors)."""
        segs = [np.ascontiguousarray(s, np.uint8).reshape(-1, 32) for s in segments]
        start = np.zeros(len(segs) + 1, np.int32)
        start[1:] = np.cumsum([len(s) for s in segs])
        rows = np.concatenate(segs) if segs and start[-1] > 0 else np.zeros((1, 32), np.uint8)
        best = np.zeros(len(segs), np.int32)
        out = np.zeros((len(segs), 32), np.uint8)
        L.check(self._lib.orbx_distinctive_descriptors(self._handle(), L.ptr(rows), L.ptr(start), len(segs),
                                                       L.ptr(best), L.ptr(out)), "orbx_distinctive_descriptors")
        return best, out

    def SearchByBoW(self, keyframes, frame, strict_lt=False):
        """ORBmatcher::SearchByBoW (src/ORBmatcher.cc:176-311; strict_lt=True with frame["valid"] set is
        the key-frame-against-key-frame form, :552-686) of a list of key frames against one frame.
        Every item is a dict: desc (n,32) uint8, angle (n,) float32, valid (n,) bool or None, and the
        feature vector as node_ids (ascending), node_start, node_items (int32 CSR).
        Returns (nmatches per key frame, [match12 per key frame], match21 (nb, n2))."""
        nb = len(keyframes)
        if nb == 0:
            return np.zeros(0, np.int32), [], np.zeros((0, len(frame["desc"])), np.int32)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)  # noqa: E731
        feat_start = np.zeros(nb + 1, np.int32)
        feat_start[1:] = np.cumsum([len(k["desc"]) for k in keyframes])
        node_kf_start = np.zeros(nb + 1, np.int32)
        node_kf_start[1:] = np.cumsum([len(k["node_ids"]) for k in keyframes])
        cat = lambda key, dt, shape: (np.ascontiguousarray(np.concatenate(  # noqa: E731
            [np.asarray(k[key], dt).reshape(shape) for k in keyframes]), dt))
        d1 = cat("desc", np.uint8, (-1, 32))
        a1 = cat("angle", np.float32, (-1,))
        has_valid = nb > 0 and keyframes[0].get("valid") is not None
        v1 = cat("valid", np.uint8, (-1,)) if has_valid else None
        ids1 = cat("node_ids", np.int32, (-1,))
        items1 = cat("node_items", np.int32, (-1,))
        # node_start of every key frame is relative to its own items: shift to the concatenation
        starts, base = [], 0
        for k in keyframes:
            st = i32(k["node_start"])
            starts.append(st[:-1] + base)
            base += int(st[-1])
        start1 = i32(np.concatenate(starts + [np.array([base], np.int32)])) if nb else np.zeros(1, np.int32)
        d2 = np.ascontiguousarray(frame["desc"], np.uint8).reshape(-1, 32)
        a2 = np.ascontiguousarray(frame["angle"], np.float32)
        v2 = None if frame.get("valid") is None else np.ascontiguousarray(frame["valid"], np.uint8)
        ids2, start2, items2 = i32(frame["node_ids"]), i32(frame["node_start"]), i32(frame["node_items"])
        n2 = len(d2)
        m12 = np.zeros(max(len(d1), 1), np.int32)
        m21 = np.zeros((max(nb, 1), max(n2, 1)), np.int32)
        nm = np.zeros(max(nb, 1), np.int32)
        P = lambda a: None if a is None else L.ptr(a)  # noqa: E731
        L.check(self._lib.orbx_search_by_bow(self._handle(), P(d1), P(a1), P(v1), P(feat_start), nb, P(node_kf_start),
                                             P(ids1), P(start1), P(items1), P(d2), P(a2), P(v2), n2, P(ids2),
                                             P(start2), P(items2), len(ids2), int(strict_lt), self.mfNNratio,
                                             int(self.mbCheckOrientation), P(m12), P(m21), P(nm)),
                "orbx_search_by_bow")
        per_kf = [m12[feat_start[b]:feat_start[b + 1]].copy() for b in range(nb)]
        return nm[:nb].copy(), per_kf, m21[:nb, :n2].copy()

    def feature_vectors_build_device(self, d_node, d_counts, nframes, max_keypoints, d_node_ids, d_node_start,
                                     d_node_items, d_num_nodes):
        """DBoW2::FeatureVector of device-resident frames (orbx_feature_vectors_build_device in include/orbx.h);
        device addresses (ints); asynchronous on the matcher's stream."""
        P = C.c_void_p
        L.check(self._lib.orbx_feature_vectors_build_device(self._handle(), P(d_node), P(d_counts), nframes,
                                                            max_keypoints, P(d_node_ids), P(d_node_start),
                                                            P(d_node_items), P(d_num_nodes)),
                "orbx_feature_vectors_build_device")

    def search_by_bow_device(self, side1, nb, max_keypoints1, side2, max_keypoints2, d_match12, d_match21, d_nmatches,
                             strict_lt=False):
        """ORBmatcher::SearchByBoW on device-resident frames (orbx_search_by_bow_device in include/orbx.h).
        side1 = (d_keypoints, d_desc, d_counts, d_valid or 0, d_node_ids, d_node_start, d_node_items),
        side2 = (d_keypoints, d_desc, d_valid or 0, d_node_ids, d_node_start, d_node_items): device addresses."""
        P = lambda v: C.c_void_p(v) if v else None  # noqa: E731
        k1, de1, c1, v1, i1, s1, t1 = side1
        k2, de2, v2, i2, s2, t2 = side2
        L.check(self._lib.orbx_search_by_bow_device(self._handle(), P(k1), P(de1), P(c1), P(v1), nb, max_keypoints1,
                                                    P(i1), P(s1), P(t1), P(k2), P(de2), P(v2), max_keypoints2, P(i2),
                                                    P(s2), P(t2), int(strict_lt), self.mfNNratio,
                                                    int(self.mbCheckOrientation), P(d_match12), P(d_match21),
                                                    P(d_nmatches)), "orbx_search_by_bow_device")

    def search_by_projection(self, keypoints, descriptors, bounds, x, y, r, min_level, max_level, src_desc, *,
                             uright=None, taken=None, valid=None, xr=None, blocks=None, src_angle=None,
                             use_second=True, th_dist=100, check_orientation=False):
        """The window search both tracking forms of ORBmatcher::SearchByProjection share (src/ORBmatcher.cc:46-137,
        :1412-1566); arguments as orbx_search_by_projection in include/orbx.h.  Returns (nmatches, match per
        source, feature_match per frame feature: source index, -1 none, -2 cleared by the rotation check)."""
        kps = np.ascontiguousarray(keypoints, L.KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n, ns = len(kps), len(x)
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        ur, tk, va, xr_, bl, an = f32(uright), u8(taken), u8(valid), f32(xr), u8(blocks), f32(src_angle)
        qx, qy, qr, lo, hi = f32(x), f32(y), f32(r), i32(min_level), i32(max_level)
        sd = np.ascontiguousarray(src_desc, np.uint8).reshape(-1, 32)
        match = np.full(max(ns, 1), -1, np.int32)
        fmatch = np.full(max(n, 1), -1, np.int32)
        nm = C.c_int32(0)
        opt = lambda a: None if a is None else L.ptr(a)
        L.check(self._lib.orbx_search_by_projection(
            self._handle(), L.ptr(kps), L.ptr(desc), opt(ur), opt(tk), n, bounds[0], bounds[1], bounds[2], bounds[3],
            opt(va), L.ptr(qx), L.ptr(qy), L.ptr(qr), L.ptr(lo), L.ptr(hi), opt(xr_), L.ptr(sd), opt(bl), opt(an), ns,
            int(bool(use_second)), float(self.mfNNratio), int(th_dist), int(check_orientation), L.ptr(match),
            L.ptr(fmatch), C.byref(nm)), "orbx_search_by_projection")
        return int(nm.value), match[:ns].copy(), fmatch[:n].copy()

    def SearchByProjectionMapPoints(self, keypoints, descriptors, bounds, scale_factors, track_in_view, proj_x, proj_y,
                                    scale_level, view_cos, mp_desc, th=1.0, *, uright=None, proj_xr=None, taken=None,
                                    blocks=None):
        """ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th)
        (src/ORBmatcher.cc:46-137): the per-map-point fields are mbTrackInView && !isBad, mTrackProjX, mTrackProjY,
        mnTrackScaleLevel, mTrackViewCos, GetDescriptor(), mTrackProjXR."""
        lvl = np.ascontiguousarray(scale_level, np.int32)
        r = np.where(np.asarray(view_cos, np.float32) > np.float32(0.998), np.float32(2.5), np.float32(4.0))
        if th != 1.0:
            r = r * np.float32(th)
        r = (r.astype(np.float32) * np.asarray(scale_factors, np.float32)[lvl]).astype(np.float32)
        return self.search_by_projection(keypoints, descriptors, bounds, proj_x, proj_y, r, lvl - 1, lvl, mp_desc,
                                         uright=uright, taken=taken, valid=track_in_view, xr=proj_xr, blocks=blocks,
                                         use_second=True, th_dist=self.TH_HIGH)

    def SearchByProjectionLastFrame(self, keypoints, descriptors, bounds, scale_factors, valid, u, v, last_octave,
                                    last_angle, mp_desc, th, *, forward=False, backward=False, uright=None, ur=None,
                                    taken=None, blocks=None, check_orientation=True):
        """ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
        (src/ORBmatcher.cc:1412-1566) from the projections (u, v[, ur]) of the last frame's map points: valid =
        has a map point, no outlier, in front of the camera and inside the image bounds (:1438-1462)."""
        octv = np.ascontiguousarray(last_octave, np.int32)
        r = (np.float32(th) * np.asarray(scale_factors, np.float32)[octv]).astype(np.float32)
        if forward:
            lo, hi = octv, np.full_like(octv, -1)
        elif backward:
            lo, hi = np.zeros_like(octv), octv
        else:
            lo, hi = octv - 1, octv + 1
        return self.search_by_projection(keypoints, descriptors, bounds, u, v, r, lo, hi, mp_desc, uright=uright,
                                         taken=taken, valid=valid, xr=ur, blocks=blocks, src_angle=last_angle,
                                         use_second=False, th_dist=self.TH_HIGH, check_orientation=check_orientation)

    def SearchByProjectionKeyFrame(self, keypoints, descriptors, bounds, scale_factors, valid, u, v, predicted_level,
                                   kf_angle, mp_desc, th, ORBdist, *, taken=None):
        """ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)
        (src/ORBmatcher.cc:1568-1690, Tracking::Relocalization): valid = the key frame's map point exists, is not
        bad, not in sAlreadyFound, projects into the image and lies in its distance range (:1585-1617); (u, v) its
        projection, predicted_level = PredictScale, kf_angle = pKF->mvKeysUn[i].angle, taken = the frame's
        features that hold a map point.  Best only, accepted if best <= ORBdist; histogram factor 1/HISTO_LENGTH."""
        lvl = np.ascontiguousarray(predicted_level, np.int32)
        r = (np.float32(th) * np.asarray(scale_factors, np.float32)[lvl]).astype(np.float32)
        return self.search_by_projection(keypoints, descriptors, bounds, u, v, r, lvl - 1, lvl + 1, mp_desc, taken=taken,
                                         valid=valid, src_angle=kf_angle, use_second=False, th_dist=ORBdist,
                                         check_orientation=2 if self.mbCheckOrientation else 0)

    def UndistortKeyPoints(self, keypoints, cam, dist):
        """Frame::UndistortKeyPoints (src/Frame.cc:467-499): keypoints = KEYPOINT_DTYPE array (mvKeys), cam = (fx,
        fy, cx, cy) of mK, dist = mDistCoef (k1, k2, p1, p2[, k3]).  Returns mvKeysUn."""
        kps = np.ascontiguousarray(keypoints, L.KEYPOINT_DTYPE).copy()
        xy = np.ascontiguousarray(np.stack([kps["x"], kps["y"]], axis=1), np.float32)
        out = self.undistort_points(xy, cam, dist)
        kps["x"], kps["y"] = out[:, 0], out[:, 1]
        return kps

    def undistort_points(self, xy, cam, dist):
        """cv::undistortPoints(xy, xy, mK, mDistCoef, Mat(), mK) on an (n, 2) float32 array."""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        cam = np.ascontiguousarray(cam, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        out = np.zeros_like(xy)
        L.check(self._lib.orbx_undistort_points(self._handle(), L.ptr(xy), len(xy), L.ptr(cam), L.ptr(dist), len(dist),
                                                L.ptr(out)), "orbx_undistort_points")
        return out

    def ComputeImageBounds(self, cols, rows, cam, dist):
        """Frame::ComputeImageBounds (src/Frame.cc:501-531): (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
        dist = np.ascontiguousarray(dist, np.float32)
        if dist[0] == 0.0:
            return 0.0, float(cols), 0.0, float(rows)
        c = self.undistort_points(np.array([[0, 0], [cols, 0], [0, rows], [cols, rows]], np.float32), cam, dist)
        return (float(min(c[0, 0], c[2, 0])), float(max(c[1, 0], c[3, 0])), float(min(c[0, 1], c[1, 1])),
                float(max(c[2, 1], c[3, 1])))

    def undistort_keypoints_device(self, d_keypoints, n, cam, dist, d_out):
        """Device form: d_keypoints / d_out are device pointers to n orbx_keypoint records (may be equal)."""
        cam = np.ascontiguousarray(cam, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        L.check(self._lib.orbx_undistort_keypoints_device(self._handle(), d_keypoints, n, L.ptr(cam), L.ptr(dist),
                                                          len(dist), d_out), "orbx_undistort_keypoints_device")

    def GetFeaturesInArea(self, keypoints, bounds, x, y, r, min_level, max_level, cap=512):
        """Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea (src/Frame.cc:239-253, 349-420) for a batch of
        queries: keypoints = KEYPOINT_DTYPE array (mvKeysUn), bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY),
        x/y/r/min_level/max_level = arrays of equal length.  Returns (list of index arrays, cell_start,
        cell_items)."""
        kps = np.ascontiguousarray(keypoints, L.KEYPOINT_DTYPE)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (x, y, r))
        lo, hi = (np.ascontiguousarray(a, np.int32) for a in (min_level, max_level))
        nq = len(qx)
        out = np.zeros((max(nq, 1), cap), np.int32)
        cnt = np.zeros(max(nq, 1), np.int32)
        cs = np.zeros(64 * 48 + 1, np.int32)
        ci = np.zeros(max(len(kps), 1), np.int32)
        L.check(self._lib.orbx_features_in_area(self._handle(), L.ptr(kps), len(kps), bounds[0], bounds[1], bounds[2],
                                                bounds[3], L.ptr(qx), L.ptr(qy), L.ptr(qr), L.ptr(lo), L.ptr(hi), nq,
                                                cap, L.ptr(out), L.ptr(cnt), L.ptr(cs), L.ptr(ci)),
                "orbx_features_in_area")
        if nq and cnt[:nq].max() > cap:
            raise L.OrbxError(-3, "GetFeaturesInArea: a list exceeds cap=%d" % cap)
        return [out[q, :cnt[q]].copy() for q in range(nq)], cs, ci[:cs[-1]].copy()
