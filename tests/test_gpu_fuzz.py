"""Randomised differential tests on the GPU: the HIP path through the C ABI against the CPU oracle on
seeded random image sizes, ORB settings, image statistics and row strides (extractor), and on random query
/ database sizes with planted near-duplicates and ties (matcher).  Same bars as the fixed-size tests:
keypoints, octaves, responses bit-exact; angles within 1e-3 degrees; descriptor bits within 0.1 %; match
indices and distances bit-exact.  ORBX_FUZZ_CASES sets the number of cases (default 24; a soak run uses
several hundred), ORBX_FUZZ_SEED the seed."""
import os

import numpy as np
import pytest

import orbslam2_amd as orbx

from test_gpu_extractor import compare_frame

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("ORBX_FUZZ_CASES", "24"))
SEED = int(os.environ.get("ORBX_FUZZ_SEED", "20240611"))


def random_image(rng, w, h):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return orbx.synth_image(int(rng.integers(1, 1 << 40)), w, h, 0), "rectangles"
    if kind == 1:
        return orbx.synth_image(int(rng.integers(1, 1 << 40)), w, h, 2), "low contrast"
    if kind == 2:   # white noise of random amplitude: very many weak corners, equal scores
        amp = int(rng.integers(4, 128))
        return (128 + rng.integers(-amp, amp + 1, size=(h, w))).clip(0, 255).astype(np.uint8), "noise %d" % amp
    if kind == 3:   # blobs on a gradient
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = 40 + 150 * xx / max(w - 1, 1) + 30 * np.sin(yy / 9.0)
        for _ in range(int(rng.integers(5, 60))):
            cx, cy, s = rng.random() * w, rng.random() * h, 2 + rng.random() * 12
            img += (rng.random() * 200 - 100) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        return img.clip(0, 255).astype(np.uint8), "blobs"
    if kind == 4:   # checkerboard of random pitch with saturated values: plateaus of equal scores
        pitch = int(rng.integers(3, 40))
        yy, xx = np.mgrid[0:h, 0:w]
        img = (((xx // pitch) + (yy // pitch)) & 1) * 255
        return img.astype(np.uint8), "board %d" % pitch
    # sparse salt on black / pepper on white
    base = 0 if rng.random() < 0.5 else 255
    img = np.full((h, w), base, np.uint8)
    n = int(w * h * rng.random() * 0.02) + 1
    img[rng.integers(0, h, n), rng.integers(0, w, n)] = 255 - base
    return img, "salt"


def test_extractor_random_cases(oracle):
    rng = np.random.default_rng(SEED)
    ran = 0
    log = []
    for case in range(CASES):
        w = int(rng.integers(44, 900))
        h = int(rng.integers(44, min(700, int(w * 1.3) + 1)))
        p = dict(nfeatures=int(rng.choice([30, 200, 500, 1000, 2000, 3000])),
                 scale=float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0])), nlevels=int(rng.integers(1, 11)),
                 ini=int(rng.integers(2, 60)), minth=int(rng.integers(1, 30)))
        variant = int(rng.integers(0, 2))
        img, kind = random_image(rng, w, h)
        stride = w if rng.random() < 0.5 else w + int(rng.integers(1, 70))
        buf = np.zeros((h, stride), np.uint8)
        buf[:, :w] = img
        tag = "case %d: %dx%d stride %d %s %s blur %d" % (case, w, h, stride, kind, p, variant)
        try:
            ex = orbx.ORBextractor(p["nfeatures"], p["scale"], p["nlevels"], p["ini"], p["minth"], max_width=w,
                                   max_height=h, blur_variant=variant)
            kg, dg = ex(buf[:, :w])
        except orbx.OrbxError as e:
            # levels the reference cannot distribute (round(width/height) of the detection region is 0) or
            # whose size rounds to zero are refused; nothing to compare
            assert e.code in (-1, -4), tag + ": " + str(e)
            log.append(tag + " -> refused (%d)" % e.code)
            continue
        ko, do = oracle.extract(np.ascontiguousarray(img), p["nfeatures"], p["scale"], p["nlevels"], p["ini"],
                                p["minth"], variant, False)
        try:
            compare_frame(kg, dg, ko, do)
        except AssertionError as e:
            raise AssertionError(tag + ": " + str(e)) from e
        ex.close()
        ran += 1
        log.append(tag + " -> %d keypoints" % len(kg))
    out = os.environ.get("ORBX_FUZZ_LOG")
    if out:
        with open(out, "w") as fh:
            fh.write("\n".join(log) + "\n%d of %d cases compared\n" % (ran, CASES))
    assert ran >= CASES * 3 // 4


def test_matcher_random_cases(oracle):
    rng = np.random.default_rng(SEED + 1)
    m = orbx.ORBmatcher(0.7, True, max_queries=4096)
    for case in range(max(CASES // 2, 6)):
        nq = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1000, 2048, int(rng.integers(1, 4097))]))
        ndb = int(rng.choice([0, 1, 2, 31, 32, 33, 1023, 1024, 1025, int(rng.integers(1, 60000))]))
        db = rng.integers(0, 256, size=(ndb, 32), dtype=np.uint8)
        q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
        if ndb:
            # queries near database rows, exact duplicates (distance 0) and duplicated rows (index ties)
            src = rng.integers(0, ndb, size=nq)
            near = rng.random(nq) < 0.6
            q[near] = db[src[near]]
            flips = rng.integers(0, 60, size=nq)
            for i in np.nonzero(near)[0]:
                bits = np.unpackbits(q[i])
                bits[rng.choice(256, size=int(flips[i]), replace=False)] ^= 1
                q[i] = np.packbits(bits)
            ndup = min(ndb // 3, 50)
            if ndup:
                a, b = rng.integers(0, ndb, size=ndup), rng.integers(0, ndb, size=ndup)
                db[a] = db[b]
        base = int(rng.choice([0, 7, 1 << 20]))
        bi, d1, d2 = m.match_top2(q, db, idx_base=base)
        oi, o1, o2 = oracle.match_top2(q, db, base)
        tag = "case %d: nq %d ndb %d base %d" % (case, nq, ndb, base)
        assert np.array_equal(bi, oi), tag
        assert np.array_equal(d1, o1) and np.array_equal(d2, o2), tag


def _noisy_shift(rng, img, dx, dy):
    out = np.roll(img, (dy, dx), axis=(0, 1)).astype(np.int16) + rng.integers(-2, 3, size=img.shape)
    return np.clip(out, 0, 255).astype(np.uint8)


def test_matchers_on_random_frame_pairs(oracle):
    """Two frames of a random size and ORB setting (the second = the first moved a few pixels, fresh noise),
    then every matcher that works on frames against the oracle: SearchForInitialization (host form and
    device batch form), ComputeStereoMatches, the frame grid with window queries, SearchByProjection in both
    tracking forms, ComputeDistinctiveDescriptors on random observation lists."""
    import torch
    rng = np.random.default_rng(SEED + 2)
    for case in range(max(CASES // 4, 4)):
        w = int(rng.integers(200, 1000))
        h = int(rng.integers(160, min(520, w) + 1))   # no taller than wide: every level gets a quadtree root
        p = dict(nfeatures=int(rng.choice([300, 1000, 2000])), scale=float(rng.choice([1.2, 1.2, 1.3])),
                 nlevels=int(rng.integers(3, 9)), ini=int(rng.integers(10, 30)), minth=int(rng.integers(3, 10)))
        tag = "case %d: %dx%d %s" % (case, w, h, p)
        a = orbx.synth_image(int(rng.integers(1, 1 << 40)), w, h, 0)
        b = _noisy_shift(rng, a, int(rng.integers(-12, 13)), int(rng.integers(-3, 4)))
        exl = orbx.ORBextractor(p["nfeatures"], p["scale"], p["nlevels"], p["ini"], p["minth"], max_width=w,
                                max_height=h, max_batch=2)
        exr = orbx.ORBextractor(p["nfeatures"], p["scale"], p["nlevels"], p["ini"], p["minth"], max_width=w,
                                max_height=h, max_batch=2)
        exl.extract_batch_host([a, b])
        exr.extract_batch_host([b, a])
        (k1, d1), (k2, d2) = exl.batch_download(0), exl.batch_download(1)
        bounds = (0.0, float(w), 0.0, float(h))
        ratio, ori, win = float(rng.choice([0.6, 0.75, 0.9])), bool(rng.integers(0, 2)), int(rng.choice([10, 30, 100]))
        m = orbx.ORBmatcher(ratio, ori)
        # --- SearchForInitialization, host form
        prev_g = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
        prev_o = prev_g.copy()
        n, m12, pairs = m.SearchForInitialization(k1, d1, k2, d2, bounds, prev_g, win)
        on, om12, opairs = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev_o, win, ratio, ori)
        assert n == on and pairs == opairs and np.array_equal(m12, om12) and np.array_equal(prev_g, prev_o), tag
        # --- the same pair through the device batch form
        d_kp, d_desc, d_cnt, cap = exl.device_results()
        t12 = torch.full((1, cap), -7, dtype=torch.int32, device="cuda")
        tn = torch.zeros(1, dtype=torch.int32, device="cuda")
        tp = torch.zeros(1, dtype=torch.int32, device="cuda")
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        n0 = int((k1["octave"] == 0).sum()), int((k2["octave"] == 0).sum())
        if max(n0) <= 1024:
            m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, 2, max(max(n0), 1), bounds,
                                                     t12.data_ptr(), tn.data_ptr(), tp.data_ptr(), windowSize=win)
            torch.cuda.synchronize()
            assert int(tn[0]) == on and int(tp[0]) == opairs, tag
            assert np.array_equal(t12.cpu().numpy()[0, :len(k1)], om12), tag
        m.set_stream(None)
        # --- ComputeStereoMatches: (a, b) and (b, a) as stereo pairs
        mb, mbf = 0.537, float(rng.choice([386.1448, 40.0]))
        sf, inv = exl.GetScaleFactors(), exl.GetInverseScaleFactors()
        for f in range(2):
            nm, ur, dp = orbx.ComputeStereoMatches(exl, exr, mb, mbf, frame=f)
            kl, dl = exl.batch_download(f)
            kr, dr = exr.batch_download(f)
            pl = [exl.pyramid_level(f, l, 0) for l in range(p["nlevels"])]
            pr = [exr.pyramid_level(f, l, 0) for l in range(p["nlevels"])]
            onm, our, odp, _ = oracle.compute_stereo_matches(kl, dl, kr, dr, pl, pr, sf, inv, mb, mbf)
            assert nm == onm and np.array_equal(ur, our) and np.array_equal(dp, odp), tag + " stereo %d" % f
        if len(k2) == 0 or len(k1) == 0:
            continue
        # --- frame grid and window queries on frame 2
        nq = 200
        qx = (rng.random(nq) * (w + 60) - 30).astype(np.float32)
        qy = (rng.random(nq) * (h + 60) - 30).astype(np.float32)
        qr = rng.choice([4.0, 9.0, 15.0, 33.0], size=nq).astype(np.float32)
        lo = rng.choice([-1, 0, 1, 3], size=nq).astype(np.int32)
        hi = np.where(rng.random(nq) < 0.3, -1, lo + rng.integers(0, 3, size=nq)).astype(np.int32)
        got, cs, ci = m.GetFeaturesInArea(k2, bounds, qx, qy, qr, lo, hi, cap=2048)
        want, ocs, oci = oracle.grid_queries(k2, bounds, qx, qy, qr, lo, hi, cap=2048)
        assert np.array_equal(cs, ocs) and np.array_equal(ci, oci), tag
        assert all(np.array_equal(g, o) for g, o in zip(got, want)), tag
        # --- SearchByProjection: frame 1's features as sources, searched in frame 2
        scale = np.asarray(sf, np.float32)
        th = float(rng.choice([3.0, 7.0, 15.0]))
        r = (np.float32(th) * scale[k1["octave"]]).astype(np.float32)
        octv = k1["octave"].astype(np.int32)
        valid = (rng.random(len(k1)) < 0.9).astype(np.uint8)
        taken = (rng.random(len(k2)) < 0.1).astype(np.uint8)
        blocks = (rng.random(len(k1)) < 0.8).astype(np.uint8)
        for use_second in (True, False):
            lo2, hi2 = (octv - 1, octv) if use_second else (octv - 1, octv + 1)
            try:
                nm, match, fm = m.search_by_projection(k2, d2, bounds, k1["x"], k1["y"], r, lo2, hi2, d1, valid=valid,
                                                       taken=taken, blocks=blocks, src_angle=k1["angle"],
                                                       use_second=use_second, check_orientation=ori and not use_second)
            except orbx.OrbxError as e:
                assert e.code == -3, tag       # a window with more than 256 candidates
                continue
            onm, omatch, ofm = oracle.search_by_projection(k2, d2, bounds, k1["x"], k1["y"], r, lo2, hi2, d1, ratio,
                                                           valid=valid, taken=taken, blocks=blocks,
                                                           src_angle=k1["angle"], use_second=use_second,
                                                           check_orientation=ori and not use_second)
            assert nm == onm and np.array_equal(match, omatch) and np.array_equal(fm, ofm), tag + " proj"
        # --- ComputeDistinctiveDescriptors: observation lists cut from the two frames' descriptors
        pool = np.concatenate([d1, d2])
        segs = [pool[rng.integers(0, len(pool), size=int(rng.integers(0, 90)))] for _ in range(40)]
        best, out = m.ComputeDistinctiveDescriptors(segs)
        for s, rows in enumerate(segs):
            want = oracle.distinctive_descriptor(rows)
            assert best[s] == want, tag
        exl.close(), exr.close()
