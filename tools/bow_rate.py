#!/usr/bin/env python3
"""Relocalisation-style SearchByBoW: 32 candidate key frames of 2000 features against one frame of
2000 features, 100 vocabulary nodes, through the host-array entry point (uploads and downloads
included).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402


def feature_vector(words):
    ids = np.unique(words).astype(np.int32)
    order = np.argsort(words, kind="stable").astype(np.int32)
    start = np.zeros(len(ids) + 1, np.int32)
    start[1:] = np.cumsum([(words == i).sum() for i in ids])
    return ids, start, order


def main():
    rng = np.random.default_rng(1)
    n, nkf, nodes = 2000, 32, 100
    d2 = orbx.synth_descriptors(5, n)
    w2 = rng.integers(0, nodes, size=n)
    ids2, s2, t2 = feature_vector(w2)
    frame = dict(desc=d2, angle=np.full(n, 90, np.float32), valid=None, node_ids=ids2, node_start=s2, node_items=t2)
    kfs = []
    for _ in range(nkf):
        pick = rng.permutation(n)
        desc = d2[pick].copy()
        desc[:, :2] ^= rng.integers(0, 256, size=(n, 2), dtype=np.uint8)  # about 8 differing bits
        ids, st, it = feature_vector(w2[pick])
        kfs.append(dict(desc=desc, angle=np.full(n, 100, np.float32), valid=rng.random(n) < 0.7, node_ids=ids,
                        node_start=st, node_items=it))
    m = orbx.ORBmatcher(0.75, True)
    nm, _, _ = m.SearchByBoW(kfs, frame)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        nm, _, _ = m.SearchByBoW(kfs, frame)
    dt = (time.perf_counter() - t0) / reps
    pairs = sum(int((w2[:, None] == np.arange(nodes)[None]).sum(0) @ (w2[:, None] == np.arange(nodes)[None]).sum(0))
                for _ in range(1)) * nkf
    print(json.dumps({"workload": "SearchByBoW, 32 key frames x 2000 features against a frame of 2000, 100 nodes",
                      "ms_per_call_host_arrays": round(dt * 1e3, 3), "key_frames_per_s": round(nkf / dt, 1),
                      "candidate_distances_per_call_upper_bound": pairs, "mean_matches": float(nm.mean())}))


if __name__ == "__main__":
    main()
