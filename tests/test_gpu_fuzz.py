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
