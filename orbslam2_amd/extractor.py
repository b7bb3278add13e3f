"""Python mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-111 of the reference)
on top of the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L


def synth_image(seed, width, height, variant=0, stride=None):
    """Deterministic synthetic frame (orbx_synth_image). Returns an (height, stride) uint8 array
    whose first `width` columns are the image."""
    stride = width if stride is None else stride
    img = np.zeros((height, stride), np.uint8)
    L.check(L.lib().orbx_synth_image(seed, width, height, variant, L.ptr(img), stride), "orbx_synth_image")
    return img


class Pipeline:
    """Extractor handles of one device that take turns on batches (orbx_pipeline_create): see
    ORBextractor.set_pipeline.  GPU_MAX_HW_QUEUES should be 8 or more before the first HIP call."""

    def __init__(self, device=0):
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.orbx_pipeline_create(int(device), C.byref(h)), "orbx_pipeline_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.orbx_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
    (constructor src/ORBextractor.cc:432-506).  Extra keyword arguments size the device workspace."""

    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, *, max_width=1920, max_height=1088,
                 max_batch=1, device=0, blur_variant=0):
        p = L.Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, blur_variant, max_width, max_height,
                     max_batch, device)
        self._h = C.c_void_p()
        self._lib = L.lib()
        L.check(self._lib.orbx_extractor_create(C.byref(p), C.byref(self._h)), "orbx_extractor_create")
        self.nlevels = nlevels
        self.max_batch = max_batch
        self.max_keypoints = self._lib.orbx_extractor_max_keypoints(self._h)
        self._last = None  # (batch, width, height)
        self.mvImagePyramid = []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.orbx_extractor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:63-83) ----
    def GetLevels(self):
        return self._lib.orbx_extractor_levels(self._h)

    def _tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        L.check(self._lib.orbx_extractor_scale_tables(self._h, *[a.ctypes.data_as(C.POINTER(C.c_float)) for a in t]),
                "orbx_extractor_scale_tables")
        return t

    def GetScaleFactor(self):
        return float(self._tables()[0][1]) if self.nlevels > 1 else 1.0

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        L.check(self._lib.orbx_extractor_features_per_level(self._h, q.ctypes.data_as(C.POINTER(C.c_int))),
                "orbx_extractor_features_per_level")
        return q

    def set_stream(self, hip_stream):
        L.check(self._lib.orbx_extractor_set_stream(self._h, C.c_void_p(hip_stream)), "orbx_extractor_set_stream")

    def set_pipeline(self, pipeline):
        """Attaches the handle to `pipeline` (a Pipeline shared by the handles of one device that take turns on
        batches: orbx_extractor_set_pipeline); None detaches."""
        L.check(self._lib.orbx_extractor_set_pipeline(self._h, pipeline._h if pipeline is not None else None),
                "orbx_extractor_set_pipeline")
        self._pipeline = pipeline  # keeps it alive as long as the handle uses it

    # ---- operator() (src/ORBextractor.cc:1114-1182) ----
    def __call__(self, image, mask=None, pyramid_border=None):
        """Returns (keypoints, descriptors): a KEYPOINT_DTYPE array and an (N,32) uint8 array.
        `mask` is ignored, as in the reference.  An empty image returns (None, None) (the reference
        leaves its outputs untouched); zero keypoints return empty arrays.  With pyramid_border set
        (the reference uses 19) mvImagePyramid is filled with the padded levels."""
        if image is None or image.size == 0:
            return None, None
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 image expected"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints
        kps = np.zeros(cap, L.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        L.check(self._lib.orbx_extract(self._h, L.ptr(image), w, h, image.strides[0], L.ptr(kps), L.ptr(desc), cap,
                                       C.byref(n)), "orbx_extract")
        self._last = (1, w, h)
        if pyramid_border is not None:
            self.mvImagePyramid = self.pyramid(0, pyramid_border)
        return kps[:n.value].copy(), desc[:n.value].copy()

    # ---- batch form ----
    def extract_batch_device(self, d_ptr, batch, width, height, stride, frame_stride):
        L.check(self._lib.orbx_extract_batch_device(self._h, C.c_void_p(d_ptr), batch, width, height, stride,
                                                    frame_stride), "orbx_extract_batch_device")
        self._last = (batch, width, height)

    def extract_batch_host(self, images):
        """images: list of equally sized (H,W) uint8 arrays."""
        imgs = [np.ascontiguousarray(i) for i in images]
        h, w = imgs[0].shape
        arr = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        L.check(self._lib.orbx_extract_batch_host(self._h, arr, len(imgs), w, h, imgs[0].strides[0]),
                "orbx_extract_batch_host")
        self._last = (len(imgs), w, h)

    def batch_counts(self):
        b = self._last[0]
        n = np.zeros(b, np.int32)
        nl = np.zeros((b, self.nlevels), np.int32)
        L.check(self._lib.orbx_batch_counts(self._h, n.ctypes.data_as(C.POINTER(C.c_int)),
                                            nl.ctypes.data_as(C.POINTER(C.c_int))), "orbx_batch_counts")
        return n, nl

    def batch_download(self, frame):
        cap = self.max_keypoints
        kps = np.zeros(cap, L.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        L.check(self._lib.orbx_batch_download(self._h, frame, L.ptr(kps), L.ptr(desc), cap, C.byref(n)),
                "orbx_batch_download")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def batch_download_all(self):
        """Keypoints and descriptors of every frame of the last batch with one synchronisation:
        returns a list of (keypoints, descriptors) per frame."""
        b, cap = self._last[0], self.max_keypoints
        kps = np.empty((b, cap), L.KEYPOINT_DTYPE)   # rows behind a frame's count are not written
        desc = np.empty((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        L.check(self._lib.orbx_batch_download_all(self._h, L.ptr(kps), L.ptr(desc), cap, L.ptr(n)),
                "orbx_batch_download_all")
        return [(kps[f, :n[f]], desc[f, :n[f]]) for f in range(b)]

    def device_results(self):
        kp, de, cn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int(0)
        L.check(self._lib.orbx_batch_device_results(self._h, C.byref(kp), C.byref(de), C.byref(cn), C.byref(cap)),
                "orbx_batch_device_results")
        return kp.value, de.value, cn.value, cap.value

    def profile(self, enable=True):
        L.check(self._lib.orbx_extractor_profile(self._h, int(enable)), "orbx_extractor_profile")

    def stage_times(self):
        """Returns ({stage name: summed ms}, number of profiled batches); synchronises the stream."""
        ms = np.zeros(L.NSTAGES, np.float32)
        n = C.c_int(0)
        L.check(self._lib.orbx_extractor_stage_times(self._h, ms.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)),
                "orbx_extractor_stage_times")
        names = [self._lib.orbx_stage_name(i).decode() for i in range(L.NSTAGES)]
        return dict(zip(names, ms.tolist())), n.value

    # ---- mvImagePyramid (include/ORBextractor.h:85) and stage outputs ----
    def level_size(self, frame, level):
        w, h = C.c_int(0), C.c_int(0)
        L.check(self._lib.orbx_pyramid_level(self._h, frame, level, 0, None, 0, C.byref(w), C.byref(h)),
                "orbx_pyramid_level")
        return w.value, h.value

    def pyramid_level(self, frame, level, border=0):
        w, h = self.level_size(frame, level)
        out = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
        L.check(self._lib.orbx_pyramid_level(self._h, frame, level, border, L.ptr(out), out.strides[0], None, None),
                "orbx_pyramid_level")
        return out

    def pyramid(self, frame=0, border=19):
        """All levels of one frame with their reflect-101 frames in one round trip (orbx_pyramid_download):
        a list of (h+2*border, w+2*border) arrays, views into one buffer."""
        nl = self.nlevels
        offs = np.zeros(nl, np.int64)
        ws, hs = np.zeros(nl, np.int32), np.zeros(nl, np.int32)
        total = C.c_int64(0)
        L.check(self._lib.orbx_pyramid_layout(self._h, border, L.ptr(offs), L.ptr(ws), L.ptr(hs), C.byref(total)),
                "orbx_pyramid_layout")
        buf = np.zeros(total.value, np.uint8)
        L.check(self._lib.orbx_pyramid_download(self._h, frame, border, L.ptr(buf), total.value),
                "orbx_pyramid_download")
        return [buf[offs[l]:offs[l] + (ws[l] + 2 * border) * (hs[l] + 2 * border)].reshape(hs[l] + 2 * border,
                                                                                           ws[l] + 2 * border)
                for l in range(nl)]

    def debug_level_map(self, what, frame, level):
        w, h = self.level_size(frame, level)
        out = np.zeros((h, w), np.uint8)
        L.check(self._lib.orbx_debug_level_map(self._h, what, frame, level, L.ptr(out), out.strides[0]),
                "orbx_debug_level_map")
        return out

    def debug_candidates(self, frame, level):
        w, h = self.level_size(frame, level)
        cap = w * h // 4 + 8
        xs, ys, rs = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int(0)
        ip = C.POINTER(C.c_int)
        L.check(self._lib.orbx_debug_candidates(self._h, frame, level, xs.ctypes.data_as(ip), ys.ctypes.data_as(ip),
                                                rs.ctypes.data_as(ip), cap, C.byref(n)), "orbx_debug_candidates")
        return xs[:n.value].copy(), ys[:n.value].copy(), rs[:n.value].copy()


def debug_octree(xs, ys, resp, region_w, region_h, quota, device=0):
    """Standalone quadtree distribution on the GPU; returns kept candidate indices in list order."""
    xs = np.ascontiguousarray(xs, np.int32)
    ys = np.ascontiguousarray(ys, np.int32)
    resp = np.ascontiguousarray(resp, np.int32)
    cap = max(quota, 4 * max(1, round(region_w / region_h))) + 8
    out = np.zeros(cap, np.int32)
    n = C.c_int(0)
    ip = C.POINTER(C.c_int)
    L.check(L.lib().orbx_debug_octree(device, xs.ctypes.data_as(ip), ys.ctypes.data_as(ip), resp.ctypes.data_as(ip),
                                      len(xs), region_w, region_h, quota, out.ctypes.data_as(ip), cap, C.byref(n)),
            "orbx_debug_octree")
    return out[:n.value].copy()
