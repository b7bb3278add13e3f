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
    exr(right)
    nm, ur, _ = orbx.ComputeStereoMatches(ex, exr, 0.537, 386.1448)
    assert int(fields["stereo"]) == nm and nm > 200
    d = kps["x"] - ur
    assert int(fields["near16"]) == int(((ur >= 0) & (d > 15.5) & (d < 16.5)).sum())
    assert abs(np.median(d[ur >= 0]) - 16) < 1.0
