#!/usr/bin/env python3
"""Relocalisation-style SearchByBoW: 32 candidate key frames of 2000 features against one frame of
2000 features, 100 vocabulary nodes, through the host-array entry point (uploads and downloads
included) and through the device-pointer chain (feature vectors built on the device from per-feature node
ids, then the match; nothing crosses PCIe).  Prints one JSON line."""
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
    dev = device_chain(rng, m, d2, w2, kfs, nkf, n)
    print(json.dumps({"workload": "SearchByBoW, 32 key frames x 2000 features against a frame of 2000, 100 nodes",
                      "ms_per_call_host_arrays": round(dt * 1e3, 3), "key_frames_per_s": round(nkf / dt, 1),
                      "device_chain": dev,
                      "candidate_distances_per_call_upper_bound": pairs, "mean_matches": float(nm.mean())}))


def device_chain(rng, m, d2, w2, kfs, nkf, n):
    """The same key frames and frame resident in device memory in the extractor's result layout."""
    import torch

    from orbslam2_amd import _lib as L
    cap = n + 48
    kp1 = np.zeros((nkf, cap), L.KEYPOINT_DTYPE)
    de1 = np.zeros((nkf, cap, 32), np.uint8)
    nd1 = np.full((nkf, cap), -1, np.int32)
    v1 = np.zeros((nkf, cap), np.uint8)
    for b, k in enumerate(kfs):
        kp1["angle"][b, :n] = k["angle"]
        de1[b, :n] = k["desc"]
        v1[b, :n] = k["valid"]
        words = np.zeros(n, np.int32)
        for e, nid in enumerate(k["node_ids"]):
            words[k["node_items"][k["node_start"][e]:k["node_start"][e + 1]]] = nid
        nd1[b, :n] = words
    kp2 = np.zeros(cap, L.KEYPOINT_DTYPE)
    kp2["angle"][:n] = 90
    de2 = np.zeros((cap, 32), np.uint8)
    de2[:n] = d2
    nd2 = np.full(cap, -1, np.int32)
    nd2[:n] = w2
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    i32 = lambda k: torch.zeros(k, dtype=torch.int32, device="cuda")  # noqa: E731
    t = dict(kp1=dev(kp1), de1=dev(de1), nd1=dev(nd1), v1=dev(v1), c1=dev(np.full(nkf, n, np.int32)), kp2=dev(kp2),
             de2=dev(de2), nd2=dev(nd2), c2=dev(np.array([n], np.int32)))
    ids1, st1, it1, nn1 = i32(nkf * cap), i32(nkf * (cap + 1)), i32(nkf * cap), i32(nkf)
    ids2, st2, it2, nn2 = i32(cap), i32(cap + 1), i32(cap), i32(1)
    m12, m21, nm = i32(nkf * cap), i32(nkf * cap), i32(nkf)
    stream = torch.cuda.Stream()
    m.set_stream(stream.cuda_stream)
    p = lambda x: x.data_ptr()  # noqa: E731

    def build():
        m.feature_vectors_build_device(p(t["nd1"]), p(t["c1"]), nkf, cap, p(ids1), p(st1), p(it1), p(nn1))
        m.feature_vectors_build_device(p(t["nd2"]), p(t["c2"]), 1, cap, p(ids2), p(st2), p(it2), p(nn2))

    def match():
        m.search_by_bow_device((p(t["kp1"]), p(t["de1"]), p(t["c1"]), p(t["v1"]), p(ids1), p(st1), p(it1)), nkf, cap,
                               (p(t["kp2"]), p(t["de2"]), 0, p(ids2), p(st2), p(it2)), cap, p(m12), p(m21), p(nm))

    torch.cuda.synchronize()
    out = {}
    for name, fn in (("feature_vectors_ms", build), ("match_ms", match), ("frame_vector_and_match_ms", None)):
        def both():
            m.feature_vectors_build_device(p(t["nd2"]), p(t["c2"]), 1, cap, p(ids2), p(st2), p(it2), p(nn2))
            match()
        f = fn or both
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20):
            f()
        e1.record(stream)
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 20, 4)
    out["key_frames_per_s"] = round(nkf / (out["frame_vector_and_match_ms"] * 1e-3), 1)
    out["mean_matches"] = float(nm.cpu().numpy().mean())
    out["what"] = ("frame_vector_and_match = the current frame's feature vector (the key frames' are built once, when "
                   "they enter the database) plus the match of all key frames; device pointers in and out")
    return out


if __name__ == "__main__":
    main()
