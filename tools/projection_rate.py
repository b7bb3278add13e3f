#!/usr/bin/env python3
"""TrackLocalMap-style SearchByProjection: 1500 map points against a frame of 2000 features (KITTI
shape), through the host-array entry point (uploads and downloads included), and the oracle on one
host core beside it.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402


def main():
    rng = np.random.default_rng(3)
    n, ns = 2000, 1500
    kps = np.zeros(n, orbx.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 1241, n).astype(np.float32)
    kps["y"] = rng.uniform(0, 376, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    desc = orbx.synth_descriptors(7, n)
    src = rng.permutation(n)[:ns]
    x = (kps["x"][src] + rng.normal(0, 2, ns)).astype(np.float32)
    y = (kps["y"][src] + rng.normal(0, 2, ns)).astype(np.float32)
    lvl = kps["octave"][src].astype(np.int32)
    sd = desc[src].copy()
    sd[:, :2] ^= rng.integers(0, 256, size=(ns, 2), dtype=np.uint8)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    cosv = np.full(ns, 0.999, np.float32)
    m = orbx.ORBmatcher(0.8)
    args = (kps, desc, bounds, scale, np.ones(ns, np.uint8), x, y, lvl, cosv, sd, 3.0)
    nm, _, _ = m.SearchByProjectionMapPoints(*args)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        nm, _, _ = m.SearchByProjectionMapPoints(*args)
    dt = (time.perf_counter() - t0) / reps
    r = (np.float32(2.5) * np.float32(3.0) * scale[lvl]).astype(np.float32)
    t0 = time.perf_counter()
    for _ in range(reps):
        onm, _, _ = orc.search_by_projection(kps, desc, bounds, x, y, r, lvl - 1, lvl, sd, 0.8)
    dc = (time.perf_counter() - t0) / reps
    print(json.dumps({"workload": "SearchByProjection (local map form), 1500 map points against 2000 features, th 3",
                      "ms_per_call_host_arrays": round(dt * 1e3, 3), "oracle_ms_per_call_one_core": round(dc * 1e3, 3),
                      "matches": int(nm), "oracle_matches": int(onm)}))


if __name__ == "__main__":
    main()
