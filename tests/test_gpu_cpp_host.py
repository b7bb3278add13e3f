"""The C++ host layer (orbslam2_amd/cpp/orbx.hpp) on the GPU: the prebuilt test program must produce
the same keypoints and descriptors as the Python binding of the same C ABI."""
import os
import subprocess

import numpy as np
import pytest

import orbslam2_amd as orbx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_orbx_cpp")


def fnv(kps, desc):
    s = 1469598103934665603
    buf = b"".join(b"".join(k[f].tobytes() for f in ("x", "y", "size", "response", "octave", "class_id")) for k in kps)
    for b in buf + desc.tobytes():
        s = ((s ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return s


def test_cpp_host_layer_matches_python_binding():
    assert os.path.exists(BIN), "build it with __graft_entry__.build()"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "orbslam2_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([BIN, "640", "480"], capture_output=True, text=True, env=env, timeout=100)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
    fields = dict(t.split("=") for t in out.stdout.split()[1:])
    fnv_bytes = lambda *arrays: fnv([], np.concatenate([np.frombuffer(a.tobytes(), np.uint8) for a in arrays]))
    img = orbx.synth_image(0x0B5EED00, 640, 480, 0)
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    kps, desc = ex(img)
    assert int(fields["n"]) == len(kps)
    assert int(fields["fnv"], 16) == fnv(kps, desc)
    assert fields["levels"] == "8" and fields["padded"] == "%dx%d" % (533 + 38, 400 + 38)
    assert int(fields["d01"]) == orbx.ORBmatcher.DescriptorDistance(desc[0], desc[1])
    # the stereo pair of the program (right = left shifted by 16 px) through the Python binding
    right = np.concatenate([img[:, 16:], np.repeat(img[:, -1:], 16, axis=1)], axis=1)
    exr = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    kr, dr = exr(right)
    nm, ur, _ = orbx.ComputeStereoMatches(ex, exr, 0.537, 386.1448)
    assert int(fields["stereo"]) == nm and nm > 200
    d = kps["x"] - ur
    assert int(fields["near16"]) == int(((ur >= 0) & (d > 15.5) & (d < 16.5)).sum())
    assert abs(np.median(d[ur >= 0]) - 16) < 1.0
    # orbx::Matcher of the program against the Python ORBmatcher on the same two frames
    mt = orbx.ORBmatcher(0.9, True)
    best, d1, d2 = mt.match_top2(desc, dr)
    assert int(fields["top2"], 16) == fnv_bytes(best.astype(np.int32), d1.astype(np.uint16), d2.astype(np.uint16))
    assert int(fields["acc"]) == mt.ratio_test(d1, d2)[1]
    cam = (517.3, 516.5, 318.6, 255.3)
    dist = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    bd = mt.ComputeImageBounds(640, 480, cam, dist)
    assert fields["bounds"] == ",".join("%.9g" % v for v in bd)
    ku, kru = mt.UndistortKeyPoints(kps, cam, dist), mt.UndistortKeyPoints(kr, cam, dist)
    assert int(fields["und"], 16) == fnv_bytes(np.stack([ku["x"], ku["y"]], axis=1).astype(np.float32))
    prev = np.ascontiguousarray(np.stack([ku["x"], ku["y"]], axis=1), np.float32)
    nsfi, m12, _ = mt.SearchForInitialization(ku, desc, kru, dr, bd, prev, 30)
    assert int(fields["sfi"]) == nsfi and nsfi > 100
    assert int(fields["m12"]) == int(((m12.astype(np.int64) + 1) * (np.arange(len(m12)) + 1)).sum())
    sel = np.arange(0, len(ku), 7)
    area, _, _ = mt.GetFeaturesInArea(ku, bd, ku["x"][sel], ku["y"][sel], 20.0 + (sel % 5) * 8.0, np.zeros(len(sel)),
                                      np.where(sel % 3 != 0, -1, 2))
    asum = sum(int(((a.astype(np.int64) + 1) * (q + np.arange(len(a)) + 1)).sum()) for q, a in enumerate(area))
    assert int(fields["area"]) == asum and asum > 0
    seg = list(range(0, len(kps) + 1, 9))
    bidx, _ = mt.ComputeDistinctiveDescriptors([desc[a:b] for a, b in zip(seg[:-1], seg[1:])])
    assert int(fields["dist"]) == int(((bidx.astype(np.int64) + 1) * (np.arange(len(bidx)) + 1)).sum())
    sf = np.asarray(ex.GetScaleFactors(), np.float32)
    nproj, pm, pfm = mt.SearchByProjectionLastFrame(kru, dr, bd, sf, None, ku["x"], ku["y"], ku["octave"], ku["angle"],
                                                    desc, 7.0)
    psum = int(((pm.astype(np.int64) + 1) * (np.arange(len(pm)) + 1)).sum())
    psum += int(((pfm.astype(np.int64) + 3) * (np.arange(len(pfm)) + 1)).sum())
    assert int(fields["proj"]) == nproj and nproj > 100 and int(fields["pm"]) == psum


def test_cv_adaptor_driven_like_frame():
    """ORB_SLAM2::ORBextractor (orbslam2_amd/cpp/ORBextractor.{h,cc}) called the way src/Frame.cc:62-118, 256-262
    calls the class it replaces: two extractors in two threads, cv::KeyPoint / cv::Mat outputs, the lazily
    fetched mvImagePyramid, DescriptorDistance and ComputeStereoMatches as the integration guide writes them.  The
    program checks every result against the same call through the C ABI."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_adaptor")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "orbslam2_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for size in (("752", "480"), ("641", "479")):
        out = subprocess.run([exe, *size], capture_output=True, text=True, env=env, timeout=100)
        assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
