#!/usr/bin/env python3
"""Writes the golden fixtures of tests/golden/.

The reference holds no golden vectors for the ORB front end and cannot be built here (OpenCV is
absent), so these vectors come from the CPU oracle (oracle/orb_oracle.c) on seeded synthetic frames:
they pin the oracle against regressions and give the GPU path a committed target.  Small cases are
stored whole (keypoints, descriptors, per-level counts); the benchmark-sized cases as SHA-256 digests of
the keypoint fields plus their descriptor rows.
Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import orbslam2_amd as orbx  # noqa: E402  (host-side synthetic image generator only)
from oracle import pyoracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 0x0B5EED00

SMALL = [  # name, width, height, variant, nfeatures, iniTh, minTh
    ("small_160x120", 160, 120, 0, 300, 20, 7),
    ("small_lowcontrast_200x140", 200, 140, 2, 500, 20, 7),
    ("small_ini12_176x144", 176, 144, 0, 250, 12, 7),
]
LARGE = [  # name, width, height, variant, nfeatures
    ("tum_640x480_1000", 640, 480, 0, 1000),
    ("euroc_752x480_1000", 752, 480, 0, 1000),
    ("kitti_1241x376_2000", 1241, 376, 0, 2000),
    ("kitti_1241x376_4000", 1241, 376, 0, 4000),
]


def digest(kps, desc=None):
    h = hashlib.sha256()
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        h.update(np.ascontiguousarray(kps[f]).tobytes())
    if desc is not None:
        h.update(np.ascontiguousarray(desc).tobytes())
    return h.hexdigest()


def main():
    for name, w, h, var, nf, ini, mn in SMALL:
        img = orbx.synth_image(SEED + len(name), w, h, var)
        kps, desc, tr = orc.extract(img, nf, 1.2, 8, ini, mn, trace=True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, keypoints=kps, descriptors=desc,
                            ncand=np.array(tr["ncand"], np.int32), nkeep=np.array(tr["nkeep"], np.int32),
                            params=np.array([nf, 8, ini, mn], np.int32))
        print(name, len(kps))
    out = {}
    large_desc = {}
    for name, w, h, var, nf in LARGE:
        img = orbx.synth_image(SEED, w, h, var)
        kps, desc, tr = orc.extract(img, nf, 1.2, 8, 20, 7, trace=True)
        out[name] = {"width": w, "height": h, "variant": var, "nfeatures": nf, "seed": SEED, "n": int(len(kps)),
                     "ncand": [int(v) for v in tr["ncand"]], "nkeep": [int(v) for v in tr["nkeep"]],
                     "image_sha256": hashlib.sha256(img.tobytes()).hexdigest(),
                     "angles_sha256": hashlib.sha256(np.ascontiguousarray(kps["angle"]).tobytes()).hexdigest(),
                     "keypoints_sha256": digest(kps), "sha256": digest(kps, desc)}
        large_desc[name] = desc
        print(name, len(kps))
    with open(os.path.join(HERE, "large_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # the descriptors themselves: a GPU run may differ in a bounded number of bits (device cos/sin), which a
    # digest cannot express
    np.savez_compressed(os.path.join(HERE, "large_descriptors.npz"), **large_desc)
    # matcher: top-2 of 64 queries over 5000 rows
    db = orbx.synth_descriptors(11, 5000)
    q = db[::78][:64].copy()
    q[:, 3] ^= 0x3C
    best, d1, d2 = orc.match_top2(q, db)
    np.savez_compressed(os.path.join(HERE, "top2_64x5000.npz"), best=best, d1=d1, d2=d2)


if __name__ == "__main__":
    main()
