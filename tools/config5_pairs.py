#!/usr/bin/env python3
"""Extraction plus SearchForInitialization-style matching of consecutive frames (ratio 0.9, window 100,
orientation check on: src/Tracking.cc:729-730 of the reference) on KITTI-shape synthetic frames.

Frame t+1 is frame t shifted by a few pixels with fresh noise, so that real matches exist.  The
extractor runs batched on the GPU; the matcher takes each pair's keypoints and descriptors from the
host (its candidate windows and greedy replay run on the host, the pair distances on the GPU), so
the pair rate includes the downloads.  Prints one JSON line.
Usage: python tools/config5_pairs.py [--frames 256]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    args = ap.parse_args()
    w, h, n = 1241, 376, args.frames
    rng = np.random.default_rng(5)
    base = orbx.synth_image(0x0B5EED00, w, h, 0)
    frames = []
    for t in range(n):
        img = np.roll(base, (t % 7, 2 * t % 11), axis=(0, 1)).astype(np.int16) + rng.integers(-2, 3, size=base.shape)
        frames.append(np.clip(img, 0, 255).astype(np.uint8))
    ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=64)
    m = orbx.ORBmatcher(0.9, True)
    bounds = (0.0, float(w), 0.0, float(h))
    # warm-up
    ex.extract_batch_host(frames[:2])
    ex.batch_download(0)
    t0 = time.perf_counter()
    kd = []
    for s in range(0, n, 64):
        chunk = frames[s:s + 64]
        ex.extract_batch_host(chunk)
        for f in range(len(chunk)):
            kd.append(ex.batch_download(f))
    t_extract = time.perf_counter() - t0
    t0 = time.perf_counter()
    pairs = matches = 0
    for t in range(n - 1):
        (k1, d1), (k2, d2) = kd[t], kd[t + 1]
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
        nm, _, np_ = m.SearchForInitialization(k1, d1, k2, d2, bounds, prev, 100)
        pairs += np_
        matches += nm
    t_match = time.perf_counter() - t0
    print(json.dumps({"workload": "consecutive-frame initialisation matching, 1241x376, 2000 kp",
                      "frames": n, "extract_frames_per_s_host_to_host": round(n / t_extract, 1),
                      "frame_pairs_per_s": round((n - 1) / t_match, 1),
                      "candidate_pair_distances_per_s": round(pairs / t_match, 1),
                      "frames_per_s_extract_plus_match": round(n / (t_extract + t_match), 1),
                      "mean_matches_per_pair": round(matches / (n - 1), 1),
                      "mean_candidate_pairs_per_frame_pair": round(pairs / (n - 1), 1)}))


if __name__ == "__main__":
    main()
