"""CPU suite: the oracle against the committed golden fixtures, host-side entry points of the
C ABI, and the exported symbol list.  No GPU compute is called here."""
import ctypes as C
import glob
import hashlib
import json
import os
import re

import numpy as np
import pytest

import orbslam2_amd as orbx
from orbslam2_amd import _lib as L

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(kps, desc):
    h = hashlib.sha256()
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        h.update(np.ascontiguousarray(kps[f]).tobytes())
    h.update(np.ascontiguousarray(desc).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "small_*.npz"))))
def test_oracle_reproduces_small_golden(oracle, path):
    g = np.load(path)
    nf, nl, ini, mn = [int(v) for v in g["params"]]
    kps, desc, tr = oracle.extract(g["image"], nf, 1.2, nl, ini, mn, trace=True)
    assert tr["ncand"] == list(g["ncand"]) and tr["nkeep"] == list(g["nkeep"])
    assert np.array_equal(desc, g["descriptors"])
    for f in kps.dtype.names:
        assert np.array_equal(kps[f], g["keypoints"][f]), f


def test_oracle_reproduces_large_digests(oracle):
    large = json.load(open(os.path.join(GOLDEN, "large_digests.json")))
    for name, e in large.items():
        img = orbx.synth_image(e["seed"], e["width"], e["height"], e["variant"])
        assert hashlib.sha256(img.tobytes()).hexdigest() == e["image_sha256"], name
        kps, desc, tr = oracle.extract(img, e["nfeatures"], 1.2, 8, 20, 7, trace=True)
        assert len(kps) == e["n"] and tr["ncand"] == e["ncand"] and tr["nkeep"] == e["nkeep"], name
        assert digest(kps, desc) == e["sha256"], name


def test_oracle_top2_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "top2_64x5000.npz"))
    db = orbx.synth_descriptors(11, 5000)
    q = db[::78][:64].copy()
    q[:, 3] ^= 0x3C
    best, d1, d2 = oracle.match_top2(q, db)
    assert np.array_equal(best, g["best"]) and np.array_equal(d1, g["d1"]) and np.array_equal(d2, g["d2"])


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "orbx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 30
    lib = L.lib()
    for n in sorted(names):
        assert hasattr(lib, n), "liborbx.so does not export " + n
    assert names <= set(L.SIGNATURES), names - set(L.SIGNATURES)


def test_error_reporting_without_compute():
    lib = L.lib()
    assert lib.orbx_version() >= 100
    assert lib.orbx_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert lib.orbx_strerror(code) not in (b"", b"unknown error")
    h = C.c_void_p()
    assert lib.orbx_extractor_create(None, C.byref(h)) == L.E_ARG
    bad = L.Params(0, 1.2, 8, 20, 7, 0, 640, 480, 1, 0)
    assert lib.orbx_extractor_create(C.byref(bad), C.byref(h)) == L.E_ARG
    bad = L.Params(1000, 1.0, 8, 20, 7, 0, 640, 480, 1, 0)
    assert lib.orbx_extractor_create(C.byref(bad), C.byref(h)) == L.E_ARG
    bad = L.Params(1000, 1.2, 17, 20, 7, 0, 640, 480, 1, 0)
    assert lib.orbx_extractor_create(C.byref(bad), C.byref(h)) == L.E_ARG


def test_descriptor_distance_host(oracle):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(40, 32), dtype=np.uint8)
    for i in range(39):
        d = orbx.ORBmatcher.DescriptorDistance(a[i], a[i + 1])
        assert d == int(np.unpackbits(a[i] ^ a[i + 1]).sum()) == oracle.descriptor_distance(a[i], a[i + 1])
    assert orbx.ORBmatcher.DescriptorDistance(a[0], a[0]) == 0
    assert orbx.ORBmatcher.DescriptorDistance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256
    assert (orbx.ORBmatcher.TH_LOW, orbx.ORBmatcher.TH_HIGH, orbx.ORBmatcher.HISTO_LENGTH) == (50, 100, 30)


def test_merge_and_ratio_host(oracle):
    """Shard merge = single pass, checked with the oracle computing every shard on the CPU."""
    rng = np.random.default_rng(1)
    db = rng.integers(0, 256, size=(3000, 32), dtype=np.uint8)
    db[1500] = db[10]  # duplicate rows across shards: the lower index must win
    q = db[[10, 77, 1500, 2999]].copy()
    q[1, 0] ^= 1
    bounds = [0, 700, 1400, 1401, 3000]
    parts = [oracle.match_top2(q, db[a:b], idx_base=a) for a, b in zip(bounds[:-1], bounds[1:])]
    mb, m1, m2 = orbx.ORBmatcher.merge_top2(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                                            np.stack([p[2] for p in parts]))
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(mb, ob) and np.array_equal(m1, o1) and np.array_equal(m2, o2)
    assert mb[0] == 10 and m1[0] == 0 and m2[0] == 0 and mb[2] == 10
    m = orbx.ORBmatcher(0.75)
    acc, n = m.ratio_test(np.array([50, 51, 30, 0], np.uint16), np.array([67, 200, 40, 0], np.uint16))
    assert list(acc) == [True, False, False, False] and n == 1


def test_synthetic_inputs_are_deterministic():
    a = orbx.synth_image(5, 123, 77, 0, stride=128)
    b = orbx.synth_image(5, 123, 77, 0, stride=128)
    assert np.array_equal(a, b) and (a[:, 123:] == 0).all()
    assert not np.array_equal(a, orbx.synth_image(6, 123, 77, 0, stride=128))
    flat = orbx.synth_image(5, 64, 64, 1)
    assert 125 <= flat.min() and flat.max() <= 131
    d = orbx.synth_descriptors(3, 100)
    assert np.array_equal(d, orbx.synth_descriptors(3, 100)) and d.shape == (100, 32)
    assert abs(np.unpackbits(d).mean() - 0.5) < 0.02
