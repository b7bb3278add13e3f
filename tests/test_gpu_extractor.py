"""GPU parity of the extractor: every stage of the HIP path, called through the C ABI, against the
CPU oracle on the same seeded synthetic frames.  Bars: pyramid, blur, candidates, quadtree selection,
keypoint coordinates/octaves/responses bit-exact; angles within 1e-3 degrees; descriptor bit
mismatch <= 0.1 %."""
import numpy as np
import pytest

import orbslam2_amd as orbx
from orbslam2_amd.extractor import debug_octree

from conftest import EUROC_STEREO, KITTI, KITTI_LOW, TUM

pytestmark = pytest.mark.gpu

SEED = 0x0B5EED00


def make_extractor(p, w, h, batch=1, blur_variant=0):
    return orbx.ORBextractor(p["nfeatures"], p["scale"], p["nlevels"], p["ini"], p["minth"], max_width=w,
                             max_height=h, max_batch=batch, blur_variant=blur_variant)


def oracle_extract(oracle, img, p, blur_variant=0, trace=False):
    return oracle.extract(img, p["nfeatures"], p["scale"], p["nlevels"], p["ini"], p["minth"], blur_variant, trace)


DESC_BIT_TOL = 1e-3  # share of descriptor bits that may differ (cos/sin rounding); observed: none


def compare_frame(kg, dg, ko, do):
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kg[f], ko[f]), f
    if len(kg) == 0:
        return 0.0
    dang = np.abs(kg["angle"].astype(np.float64) - ko["angle"].astype(np.float64))
    dang = np.minimum(dang, 360.0 - dang)
    assert dang.max() <= 1e-3, dang.max()
    bits = np.unpackbits(dg ^ do).sum()
    frac = bits / (dg.size * 8.0)
    assert frac <= DESC_BIT_TOL, frac
    return frac


@pytest.mark.parametrize("w,h,p,variant", [
    (640, 480, TUM, 0),
    (1241, 376, KITTI, 0),
    (752, 480, TUM, 0),
    (640, 480, TUM, 2),
    (1241, 376, KITTI_LOW, 2),
    (752, 480, EUROC_STEREO, 0),
    (333, 217, TUM, 0),
    (813, 120, TUM, 0),   # level 0: the last cell column starts 6 px before the border and is skipped (:864)
    (560, 813, TUM, 0),   # the same remainder in y does not skip the row (:853): the 6-against-3 asymmetry
])
def test_stages_and_frame(oracle, w, h, p, variant):
    img = orbx.synth_image(SEED + 7 * variant, w, h, variant)
    ex = make_extractor(p, w, h)
    kg, dg = ex(img)
    ko, do, tr = oracle_extract(oracle, img, p, trace=True)
    for l in range(p["nlevels"]):
        lev = ex.pyramid_level(0, l)
        assert lev.shape == tr["levels"][l].shape
        assert np.array_equal(lev, tr["levels"][l]), "pyramid level %d" % l
        xs, ys, rs = ex.debug_candidates(0, l)
        oxs, oys, ors = oracle.level_candidates(tr["levels"][l], p["ini"], p["minth"])
        assert len(xs) == len(oxs), "candidate count level %d: %d vs %d" % (l, len(xs), len(oxs))
        assert np.array_equal(xs, oxs) and np.array_equal(ys, oys) and np.array_equal(rs, ors), l
        if tr["blur"][l] is not None:
            assert np.array_equal(ex.debug_level_map(1, 0, l), tr["blur"][l]), "blur level %d" % l
    _, per_level = ex.batch_counts()
    assert list(per_level[0]) == tr["nkeep"]
    compare_frame(kg, dg, ko, do)
    ex.close()


def test_equal_responses_across_a_cell_border(oracle):
    """Two equally bright pixels side by side: astride a cell border both are candidates (suppression never
    looks into the neighbouring cell, src/ORBextractor.cc:868-877), inside a cell neither is."""
    w, h = 176, 112
    wcell = int(np.ceil(np.float32(w - 32) / int(np.float32(w - 32) / np.float32(30))))
    b = 19 + wcell
    img = np.full((h, w), 50, np.uint8)
    img[40, b - 1:b + 1] = 200
    img[70, b + 10:b + 12] = 200
    ex = orbx.ORBextractor(500, 1.2, 1, 20, 7, max_width=w, max_height=h)
    ex(img)
    xs, ys, rs = ex.debug_candidates(0, 0)
    oxs, oys, ors = oracle.level_candidates(img, 20, 7)
    assert list(zip(xs, ys, rs)) == list(zip(oxs, oys, ors)) == [(b - 17, 24, 149), (b - 16, 24, 149)]
    ex.close()


def test_blur_variant_cv2(oracle):
    img = orbx.synth_image(SEED + 3, 640, 480, 0)
    ex = make_extractor(TUM, 640, 480, blur_variant=1)
    kg, dg = ex(img)
    ko, do, tr = oracle_extract(oracle, img, TUM, blur_variant=1, trace=True)
    assert np.array_equal(ex.debug_level_map(1, 0, 0), tr["blur"][0])
    compare_frame(kg, dg, ko, do)


def test_mono_init_extractor_4000(oracle):
    # the initialisation extractor asks for 2*nFeatures (src/Tracking.cc:128)
    p = dict(KITTI, nfeatures=4000)
    img = orbx.synth_image(SEED + 11, 1241, 376, 0)
    ex = make_extractor(p, 1241, 376)
    kg, dg = ex(img)
    ko, do = oracle_extract(oracle, img, p)
    compare_frame(kg, dg, ko, do)


def test_flat_and_empty(oracle):
    ex = make_extractor(TUM, 640, 480)
    flat = orbx.synth_image(SEED, 640, 480, 1)
    k, d = ex(flat)
    ko, do = oracle_extract(oracle, flat, TUM)
    assert len(k) == 0 and d.shape == (0, 32) and len(ko) == 0
    assert ex(np.zeros((0, 0), np.uint8)) == (None, None)
    # constant image: no corners anywhere
    k, d = ex(np.full((480, 640), 77, np.uint8))
    assert len(k) == 0


def test_tiny_images(oracle):
    # small levels lose their cell grid (fewer than 30 px of detection region): no keypoints there
    for (w, h) in [(96, 80), (70, 64), (40, 40)]:
        img = orbx.synth_image(SEED + w, w, h, 0)
        ex = make_extractor(TUM, w, h)
        kg, dg = ex(img)
        ko, do = oracle_extract(oracle, img, TUM)
        compare_frame(kg, dg, ko, do)
        ex.close()


def test_strided_input_and_reuse(oracle):
    # a row stride larger than the width, and two different sizes through one handle
    ex = make_extractor(TUM, 800, 600)
    for (w, h, stride) in [(640, 480, 704), (800, 600, 800), (641, 479, 1000)]:
        buf = orbx.synth_image(SEED + w, w, h, 0, stride=stride)
        kg, dg = ex(buf[:, :w])
        ko, do = oracle_extract(oracle, np.ascontiguousarray(buf[:, :w]), TUM)
        compare_frame(kg, dg, ko, do)


def test_batch_matches_single(oracle):
    w, h, B = 752, 480, 6
    imgs = [orbx.synth_image(SEED + i, w, h, 0 if i % 3 else 2) for i in range(B)]
    ex = make_extractor(TUM, w, h, batch=B)
    ex.extract_batch_host(imgs)
    n, _ = ex.batch_counts()
    for i in range(B):
        kg, dg = ex.batch_download(i)
        ko, do = oracle_extract(oracle, imgs[i], TUM)
        assert n[i] == len(ko)
        compare_frame(kg, dg, ko, do)


@pytest.mark.parametrize("w,h,B", [(402, 301, 40), (1241, 376, 33)])
def test_batch_of_thirty_two_or_more(oracle, w, h, B):
    """Batches of 32 frames or more take kernels of their own (16 destination rows per wave in the resize, a
    frame's workgroups on one XCD, the quadtree in two launches): every frame, also those of the incomplete last
    group of eight, must equal the oracle, and the pyramid of one of them bit for bit."""
    rng = np.random.default_rng(B)
    imgs = []
    for i in range(B):
        if i % 11 == 5:
            imgs.append((128 + rng.integers(-40, 41, size=(h, w))).clip(0, 255).astype(np.uint8))
        else:
            imgs.append(orbx.synth_image(SEED + 31 * i, w, h, 0 if i % 4 else 2))
    ex = make_extractor(KITTI, w, h, batch=B)
    ex.extract_batch_host(imgs)
    n, _ = ex.batch_counts()
    step = 1 if w < 1000 else 4  # (the oracle takes 35 ms for a frame of the larger size)
    for i in list(range(0, B, step)) + [B - 1]:
        kg, dg = ex.batch_download(i)
        ko, do, tr = oracle_extract(oracle, imgs[i], KITTI, trace=True)
        assert n[i] == len(ko)
        compare_frame(kg, dg, ko, do)
        if i == B - 1:
            for lvl in range(KITTI["nlevels"]):
                assert np.array_equal(ex.pyramid_level(i, lvl), tr["levels"][lvl]), lvl


def test_batch_of_ten_with_crowded_levels(oracle):
    """Ten frames in one batch, some with so many candidates (white noise) that a level's candidates no longer
    fit the registers of the distribution kernel's threads: each must equal the oracle, and the same frame
    extracted alone."""
    w, h, B = 1241, 376, 10
    rng = np.random.default_rng(5)
    imgs = []
    for i in range(B):
        if i % 3 == 1:
            amp = 30 + 20 * i
            imgs.append((128 + rng.integers(-amp, amp + 1, size=(h, w))).clip(0, 255).astype(np.uint8))
        else:
            imgs.append(orbx.synth_image(SEED + 17 * i, w, h, 0))
    ex = make_extractor(KITTI, w, h, batch=B)
    ex.extract_batch_host(imgs)
    n, _ = ex.batch_counts()
    one = make_extractor(KITTI, w, h)
    for i in range(B):
        kg, dg = ex.batch_download(i)
        ko, do = oracle_extract(oracle, imgs[i], KITTI)
        assert n[i] == len(ko)
        compare_frame(kg, dg, ko, do)
        k1, d1 = one(imgs[i])
        assert np.array_equal(k1, kg) and np.array_equal(d1, dg)


def test_padded_pyramid(oracle):
    img = orbx.synth_image(SEED, 320, 240, 0)
    ex = make_extractor(TUM, 320, 240)
    ex(img, pyramid_border=19)
    for l, padded in enumerate(ex.mvImagePyramid):
        inner = ex.pyramid_level(0, l)
        ref = np.pad(inner, 19, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
        assert np.array_equal(padded, ref)


@pytest.mark.parametrize("seed", range(6))
def test_octree_random(oracle, seed):
    rng = np.random.default_rng(seed)
    rw, rh = [(1209, 344), (608, 448), (300, 200), (1209, 344), (720, 448), (95, 90)][seed]
    n = [5000, 1500, 40, 20000, 3000, 300][seed]
    quota = [434, 217, 60, 869, 1, 122][seed]
    # distinct pixels, clustered to force deep and uneven subdivision
    pts = set()
    while len(pts) < n:
        if rng.random() < 0.5:
            cx, cy = rng.integers(0, rw), rng.integers(0, rh)
            x = int(np.clip(cx + rng.normal(0, 6), 0, rw - 1))
            y = int(np.clip(cy + rng.normal(0, 6), 0, rh - 1))
        else:
            x, y = int(rng.integers(0, rw)), int(rng.integers(0, rh))
        pts.add((x, y))
    pts = np.array(sorted(pts, key=lambda t: (t[1], t[0])), np.int32)
    xs, ys = pts[:, 0].copy(), pts[:, 1].copy()
    resp = rng.integers(7, 60, size=n).astype(np.int32)  # many ties
    got = debug_octree(xs, ys, resp, rw, rh, quota)
    want = oracle.distribute_octree(xs, ys, resp, 16, 16 + rw, 16, 16 + rh, quota)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("stride", [1241, 1280, 1244])
def test_device_batch_in_place_and_copied(oracle, stride):
    """Frames resident in device memory: rows that are multiples of 4 bytes are read in place,
    others are copied into the pyramid first; both must give the oracle's result."""
    import torch
    w, h, B = 1241, 376, 3
    host = np.zeros((B, h, stride), np.uint8)
    for f in range(B):
        host[f] = orbx.synth_image(SEED + 31 * f, w, h, 0, stride=stride)
    dev = torch.from_numpy(host).cuda()
    ex = make_extractor(KITTI, w, h, batch=B)
    ex.extract_batch_device(dev.data_ptr(), B, w, h, stride, h * stride)
    n, _ = ex.batch_counts()
    for f in range(B):
        kg, dg = ex.batch_download(f)
        ko, do = oracle_extract(oracle, np.ascontiguousarray(host[f][:, :w]), KITTI)
        assert n[f] == len(ko)
        compare_frame(kg, dg, ko, do)
    lev0 = ex.pyramid_level(1, 0)
    assert np.array_equal(lev0, host[1][:, :w])
    # all levels of a frame in one round trip equal the level-by-level reads (level 0 read in place or copied)
    for border in (0, 19):
        for l, got in enumerate(ex.pyramid(2, border)):
            assert np.array_equal(got, ex.pyramid_level(2, l, border)), (border, l)


def test_handles_taking_turns_in_a_pipeline(oracle):
    """Three handles attached to one pipeline, each with a stream of its own, called round robin (a handle's blur is
    enqueued on the stream of the handle called before it): every batch equals the oracle, also when a handle is
    destroyed or detached in between and when the batch is too small to take part."""
    import torch
    w, h, B = 752, 480, 6
    host = np.zeros((3, B, h, w), np.uint8)
    for i in range(3):
        for f in range(B):
            host[i, f] = orbx.synth_image(SEED + 1000 * i + 7 * f, w, h, 0)
    dev = torch.from_numpy(host).cuda()
    want = [[oracle_extract(oracle, host[i, f], EUROC_STEREO) for f in range(B)] for i in range(3)]
    pipe = orbx.Pipeline(0)
    streams = [torch.cuda.Stream() for _ in range(3)]
    exs = [make_extractor(EUROC_STEREO, w, h, batch=B) for _ in range(3)]
    for e, st in zip(exs, streams):
        e.set_stream(st.cuda_stream)
        e.set_pipeline(pipe)

    def check(i, nframes=B):
        n, _ = exs[i].batch_counts()
        for f in range(nframes):
            kg, dg = exs[i].batch_download(f)
            assert n[f] == len(want[i][f][0])
            compare_frame(kg, dg, *want[i][f])

    for rnd in range(3):  # the rounds run ahead of each other on the device: nothing waits in between
        for i in range(3):
            exs[i].extract_batch_device(dev[i].data_ptr(), B, w, h, w, h * w)
    for i in range(3):
        check(i)
    # a batch of four frames or fewer does not take part (and does not become the predecessor)
    exs[1].extract_batch_device(dev[1].data_ptr(), 3, w, h, w, h * w)
    exs[2].extract_batch_device(dev[2].data_ptr(), B, w, h, w, h * w)
    check(1, 3)
    check(2)
    # the predecessor goes away: the next handle falls back to its own second stream once
    exs[2].close()
    exs[0].extract_batch_device(dev[0].data_ptr(), B, w, h, w, h * w)
    exs[1].set_pipeline(None)
    exs[1].extract_batch_device(dev[1].data_ptr(), B, w, h, w, h * w)
    exs[0].extract_batch_device(dev[0].data_ptr(), B, w, h, w, h * w)
    check(0)
    check(1)
    for e in exs[:2]:
        e.close()
    pipe.close()


def test_gpu_reproduces_golden_fixtures():
    """The committed fixtures (tests/golden, written by tests/golden/make_golden.py from the oracle)."""
    import glob
    import hashlib
    import json
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for path in sorted(glob.glob(os.path.join(gdir, "small_*.npz"))):
        g = np.load(path)
        nf, nl, ini, mn = [int(v) for v in g["params"]]
        img = g["image"]
        ex = orbx.ORBextractor(nf, 1.2, nl, ini, mn, max_width=img.shape[1], max_height=img.shape[0])
        kg, dg = ex(img)
        compare_frame(kg, dg, g["keypoints"], g["descriptors"])
        _, per_level = ex.batch_counts()
        assert list(per_level[0]) == list(g["nkeep"])
        ex.close()
    large = json.load(open(os.path.join(gdir, "large_digests.json")))
    large_desc = np.load(os.path.join(gdir, "large_descriptors.npz"))
    for name, e in large.items():
        img = orbx.synth_image(e["seed"], e["width"], e["height"], e["variant"])
        ex = orbx.ORBextractor(e["nfeatures"], 1.2, 8, 20, 7, max_width=e["width"], max_height=e["height"])
        kg, dg = ex(img)
        assert len(kg) == e["n"], name
        # keypoint fields: bit-exact, asserted on their own digest
        h = hashlib.sha256()
        for f in ("x", "y", "size", "response", "octave", "class_id"):
            h.update(np.ascontiguousarray(kg[f]).tobytes())
        assert h.hexdigest() == e["keypoints_sha256"], name
        _, per_level = ex.batch_counts()
        assert list(per_level[0]) == e["nkeep"], name
        # descriptors: at most 0.1 % of the bits may differ from the fixture (device cos/sin against the
        # host's; observed: none)
        want = large_desc[name]
        assert want.shape == dg.shape, name
        assert int(np.unpackbits(dg ^ want).sum()) <= DESC_BIT_TOL * dg.size * 8, name
        ex.close()


def test_error_codes_and_capacity(oracle):
    import ctypes as C
    from orbslam2_amd import _lib as L
    lib = L.lib()
    ex = make_extractor(TUM, 640, 480)
    img = orbx.synth_image(SEED, 640, 480, 0)
    # caller buffers too small: the required count comes back, nothing else is written
    kps = np.zeros(10, L.KEYPOINT_DTYPE)
    desc = np.full((10, 32), 7, np.uint8)
    n = C.c_int(0)
    rc = lib.orbx_extract(ex._h, L.ptr(img), 640, 480, 640, L.ptr(kps), L.ptr(desc), 10, C.byref(n))
    assert rc == L.E_CAPACITY and n.value > 900 and (desc == 7).all()
    # larger than the workspace the handle was created for
    big = orbx.synth_image(SEED, 800, 600, 0)
    with pytest.raises(orbx.OrbxError) as e:
        ex(big)
    assert e.value.code == L.E_CAPACITY
    # a portrait image whose detection region has round(width/height) == 0: the reference divides by zero
    tall = make_extractor(TUM, 100, 400)
    with pytest.raises(orbx.OrbxError) as e:
        tall(orbx.synth_image(SEED, 100, 400, 0))
    assert e.value.code == L.E_GEOMETRY
    with pytest.raises(ValueError):
        oracle.extract(orbx.synth_image(SEED, 100, 400, 0), 1000, 1.2, 8, 20, 7)
    # a stride smaller than the width is an argument error
    rc = lib.orbx_extract(ex._h, L.ptr(img), 640, 480, 600, L.ptr(kps), L.ptr(desc), 10, C.byref(n))
    assert rc == L.E_ARG
    # results of the last good call survive a failed call's checks
    k, d = ex(img)
    ko, do = oracle_extract(oracle, img, TUM)
    compare_frame(k, d, ko, do)


def test_two_handles_from_two_threads(oracle):
    """The stereo constructor of Frame runs the left and the right extractor in two threads
    (src/Frame.cc:79-82): two handles must work concurrently."""
    import threading
    w, h = 752, 480
    imgs = [orbx.synth_image(SEED + 100 + i, w, h, 0) for i in range(2)]
    exs = [make_extractor(EUROC_STEREO, w, h) for _ in range(2)]
    out = [None, None]

    def run(i):
        for _ in range(5):
            out[i] = exs[i](imgs[i])

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(2):
        ko, do = oracle_extract(oracle, imgs[i], EUROC_STEREO)
        compare_frame(out[i][0], out[i][1], ko, do)


def test_scale_tables_match_oracle(oracle):
    ex = make_extractor(KITTI, 1241, 376)
    sf, inv, s2, is2 = oracle.scale_tables(1.2, 8)
    assert np.array_equal(ex.GetScaleFactors(), sf) and np.array_equal(ex.GetInverseScaleFactors(), inv)
    assert np.array_equal(ex.GetScaleSigmaSquares(), s2) and np.array_equal(ex.GetInverseScaleSigmaSquares(), is2)
    assert list(ex.features_per_level()) == list(oracle.features_per_level(2000, 1.2, 8))
    assert ex.GetLevels() == 8 and abs(ex.GetScaleFactor() - 1.2) < 1e-6
    # other pyramid shapes: 4 levels of scale 1.5, 12 levels of scale 1.1
    for nl, sc in [(4, 1.5), (12, 1.1)]:
        p = dict(nfeatures=800, scale=sc, nlevels=nl, ini=20, minth=7)
        img = orbx.synth_image(SEED + nl, 640, 480, 0)
        e2 = make_extractor(p, 640, 480)
        kg, dg = e2(img)
        ko, do = oracle_extract(oracle, img, p)
        compare_frame(kg, dg, ko, do)


def test_large_image_and_wide_scale_steps(oracle):
    """Full-HD frame with many features (level 0 exceeds the register-cached quadtree path), and scale
    factors whose resize takes the widest source windows (2.0), the byte-wise path (2.6), and in between (2.3, 2.45)
    the fused two-level kernel of single calls with lanes whose source span exceeds the word path's."""
    p = dict(nfeatures=5000, scale=1.2, nlevels=8, ini=20, minth=7)
    img = orbx.synth_image(SEED + 21, 1920, 1080, 0)
    ex = make_extractor(p, 1920, 1080)
    kg, dg = ex(img)
    ko, do = oracle_extract(oracle, img, p)
    compare_frame(kg, dg, ko, do)
    assert len(kg) >= 5000
    ex.close()
    for sc, nl in [(2.0, 4), (2.6, 3), (2.3, 3), (2.45, 3)]:
        p = dict(nfeatures=1500, scale=sc, nlevels=nl, ini=20, minth=7)
        img = orbx.synth_image(SEED + nl, 1280, 720, 0)
        e2 = make_extractor(p, 1280, 720)
        kg, dg = e2(img)
        ko, do, tr = oracle_extract(oracle, img, p, trace=True)
        for l in range(nl):
            assert np.array_equal(e2.pyramid_level(0, l), tr["levels"][l]), (sc, l)
        compare_frame(kg, dg, ko, do)
        e2.close()


def test_maximum_image_size(oracle):
    """The largest frame a handle accepts (4096 x 4096): tile tables, 12-bit tile coordinates and candidate
    packing at their limits, every level against the oracle's pyramid and blurred levels."""
    p = dict(nfeatures=3000, scale=1.2, nlevels=8, ini=20, minth=7)
    img = orbx.synth_image(SEED + 77, 4096, 4096, 0)
    ex = make_extractor(p, 4096, 4096)
    kg, dg = ex(img)
    ko, do, tr = oracle_extract(oracle, img, p, trace=True)
    for l in range(p["nlevels"]):
        assert np.array_equal(ex.pyramid_level(0, l), tr["levels"][l]), l
        if tr["blur"][l] is not None:
            assert np.array_equal(ex.debug_level_map(1, 0, l), tr["blur"][l]), "blur level %d" % l
    compare_frame(kg, dg, ko, do)
    ex.close()


def test_batch_download_all_equals_per_frame(oracle):
    w, h, B = 640, 480, 9   # nine frames: the staging copy runs on several threads
    imgs = [orbx.synth_image(SEED + 31 * i, w, h, 0 if i != 4 else 1) for i in range(B)]
    ex = make_extractor(TUM, w, h, batch=B)
    ex.extract_batch_host(imgs)
    both = ex.batch_download_all()
    for f in range(B):
        k, d = ex.batch_download(f)
        assert np.array_equal(both[f][0], k) and np.array_equal(both[f][1], d)
    ko, do = oracle_extract(oracle, imgs[7], TUM)
    compare_frame(both[7][0], both[7][1], ko, do)
    assert len(both[4][0]) == 0


def test_separately_pinned_frames_at_equal_distances():
    """Frames pinned one by one often lie at equal distances (a caching pinned allocator hands out equal
    blocks) without sharing an allocation: they must travel as one copy per frame, not as one copy across the
    gaps between allocations.  Same results as pageable frames."""
    import ctypes as C
    import torch
    from orbslam2_amd import _lib as L
    w, h, B = 1241, 376, 6
    imgs = [orbx.synth_image(SEED + 31 * i, w, h, 0) for i in range(B)]
    lib = L.lib()
    ex = make_extractor(KITTI, w, h, batch=B)
    ex.extract_batch_host(imgs)
    ref = ex.batch_download_all()
    # (a) torch's pinned allocator, one tensor per frame
    tens = [torch.empty((h, w), dtype=torch.uint8).pin_memory() for _ in range(B)]
    for t, im in zip(tens, imgs):
        t.numpy()[:] = im
    ex.extract_batch_host([t.numpy() for t in tens])
    got = ex.batch_download_all()
    for f in range(B):
        assert np.array_equal(ref[f][0], got[f][0]) and np.array_equal(ref[f][1], got[f][1])
    # (b) one hipHostMalloc per frame: consecutive allocations of one size, usually at equal distances
    import os
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))  # the runtime already loaded
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    ptrs, views = [], []
    for f in range(B):
        p = C.c_void_p()
        assert hip.hipHostMalloc(C.byref(p), h * w, 0) == 0
        ptrs.append(p)
        v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h, w))
        v[:] = imgs[f]
        views.append(v)
    try:
        ex.extract_batch_host(views)
        got = ex.batch_download_all()
    finally:
        for p in ptrs:
            hip.hipHostFree(p)
    for f in range(B):
        assert np.array_equal(ref[f][0], got[f][0]) and np.array_equal(ref[f][1], got[f][1])
    ex.close()


@pytest.mark.parametrize("w,h", [(640, 480), (333, 217)])
def test_page_locked_frames_and_results(oracle, w, h):
    """Frames in page-locked host memory are uploaded where they lie (one contiguous copy per frame, then
    the device path: in place for word-aligned rows, repacked otherwise), and page-locked result arrays
    are filled by the copy engine: both must give what the staged paths give."""
    import ctypes as C
    import torch
    from orbslam2_amd import _lib as L
    B = 9
    imgs = [orbx.synth_image(SEED + 17 * i, w, h, 0) for i in range(B)]
    pin = torch.empty((B, h, w), dtype=torch.uint8).pin_memory()
    pinned = [pin[i].numpy() for i in range(B)]
    for i in range(B):
        pinned[i][:] = imgs[i]
    ex = make_extractor(TUM, w, h, batch=B)
    ex.extract_batch_host(imgs)
    ref = ex.batch_download_all()
    ex.extract_batch_host(pinned)
    got = ex.batch_download_all()
    for f in range(B):
        assert np.array_equal(ref[f][0], got[f][0]) and np.array_equal(ref[f][1], got[f][1])
    ko, do = oracle_extract(oracle, imgs[3], TUM)
    compare_frame(got[3][0], got[3][1], ko, do)
    # page-locked destination
    cap = ex.max_keypoints
    kp_t = torch.empty((B, cap * L.KEYPOINT_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
    de_t = torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory()
    n = np.zeros(B, np.int32)
    L.check(L.lib().orbx_batch_download_all(ex._h, C.c_void_p(kp_t.data_ptr()), C.c_void_p(de_t.data_ptr()), cap,
                                            L.ptr(n)), "orbx_batch_download_all")
    kps = kp_t.numpy().view(L.KEYPOINT_DTYPE).reshape(B, cap)
    for f in range(B):
        assert n[f] == len(ref[f][0])
        assert np.array_equal(kps[f, :n[f]], ref[f][0]) and np.array_equal(de_t.numpy()[f, :n[f]], ref[f][1])
    ex.close()


def _texture_images(w, h):
    """Image statistics the rectangle generator does not produce: a checkerboard (hundreds of corners with
    equal responses, so the quadtree's keep-best and its last-round ordering work on ties), a low-pass
    random field (soft, blob-like corners, many near the thresholds), text-like bars with saturated
    pixels, and a steep ramp with salt noise (isolated maxima)."""
    yy, xx = np.mgrid[0:h, 0:w]
    board = (((xx // 23) + (yy // 19)) % 2 * 170 + 40).astype(np.uint8)
    rng = np.random.default_rng(99)
    field = rng.normal(0, 1, (h // 4 + 2, w // 4 + 2))
    field = np.kron(field, np.ones((4, 4)))[:h, :w]
    for _ in range(3):
        field = (field + np.roll(field, 1, 0) + np.roll(field, -1, 0) + np.roll(field, 1, 1) + np.roll(field, -1, 1)) / 5
    field = np.clip(128 + field * 220, 0, 255).astype(np.uint8)
    bars = np.full((h, w), 255, np.uint8)
    bars[(yy % 37 < 9) & (xx % 11 < 4)] = 0
    bars[(yy % 53 < 3)] = 0
    ramp = np.clip((xx * 255 // max(w - 1, 1)) + (rng.random((h, w)) < 0.002) * 90, 0, 255).astype(np.uint8)
    # white noise: most pixels pass the detector's first (four-point) test, far more than its work queue
    # holds; isolated bright pixels on the densest lattice that keeps every ring clear, (2a, a + 4b): one
    # pixel in eight is a corner, all with the same response, more than the detector's corner list holds
    white = rng.integers(0, 256, (h, w)).astype(np.uint8)
    lattice = np.where((xx % 2 == 0) & ((yy - xx // 2) % 4 == 0), 230, 20).astype(np.uint8)
    return {"board": board, "field": field, "bars": bars, "ramp": ramp, "white": white, "lattice": lattice}


@pytest.mark.parametrize("kind", ["board", "field", "bars", "ramp", "white", "lattice"])
def test_other_image_statistics(oracle, kind):
    w, h = 752, 480
    img = _texture_images(w, h)[kind]
    ex = make_extractor(EUROC_STEREO, w, h)
    kg, dg = ex(img)
    ko, do, tr = oracle_extract(oracle, img, EUROC_STEREO, trace=True)
    _, per_level = ex.batch_counts()
    assert list(per_level[0]) == tr["nkeep"]
    compare_frame(kg, dg, ko, do)
    if kind in ("board", "field", "bars"):
        assert len(kg) > 300
    ex.close()


@pytest.mark.parametrize("w,h", [(1241, 376), (333, 217), (130, 71), (44, 44)])
def test_matrix_core_blur_equals_oracle(oracle, monkeypatch, w, h):
    """The opt-in matrix-core form of the 7x7 blur (ORBX_BLUR=mfma, read when a handle is created): every blurred
    level, borders and partial tiles included, bit-equal to the oracle, and the frame's results with it."""
    monkeypatch.setenv("ORBX_BLUR", "mfma")
    p = dict(KITTI)
    img = orbx.synth_image(SEED + w, w, h, 0)
    ex = make_extractor(p, w, h)
    kg, dg = ex(img)
    ko, do, tr = oracle_extract(oracle, img, p, trace=True)
    for l in range(p["nlevels"]):
        if tr["blur"][l] is not None:
            assert np.array_equal(ex.debug_level_map(1, 0, l), tr["blur"][l]), "blur level %d" % l
    compare_frame(kg, dg, ko, do)
    ex.close()
    # a batch with rows that are not read in place (stride not a multiple of 4 is copied into the pyramid)
    ex = make_extractor(p, w, h, batch=3)
    imgs = [orbx.synth_image(SEED + 5 * i, w, h, 0) for i in range(3)]
    ex.extract_batch_host(imgs)
    got = ex.batch_download_all()
    k2, d2 = oracle_extract(oracle, imgs[2], p)
    compare_frame(got[2][0], got[2][1], k2, d2)
    ex.close()
