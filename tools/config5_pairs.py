#!/usr/bin/env python3
"""Extraction plus SearchForInitialization-style matching of consecutive frames (ratio 0.9, window 100,
orientation check on: src/Tracking.cc:729-730 of the reference) on KITTI-shape synthetic frames.

Frame t+1 is frame t shifted by a few pixels with fresh noise, so that real matches exist.  The
extractor runs batched on the GPU; the matcher takes each pair's keypoints and descriptors from the
host (its candidate windows and greedy replay run on the host, the pair distances on the GPU), so
the pair rate includes the downloads.  A second measurement keeps everything on the GPU: frames
resident in device memory, batches of 256 frames that overlap by one frame (every consecutive pair
is matched once), extraction and orbx_search_for_initialization_batch_device on one stream.
Prints one JSON line.
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


def device_resident(frames, w, h):
    import torch
    n = len(frames)
    stride = (w + 63) // 64 * 64
    host = np.zeros((n, h, stride), np.uint8)
    for t, f in enumerate(frames):
        host[t, :, :w] = f
    dev = torch.from_numpy(host).cuda()
    B = min(256, n)
    st = torch.cuda.Stream()
    ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    ex.set_stream(st.cuda_stream)
    m = orbx.ORBmatcher(0.9, True)
    m.set_stream(st.cuda_stream)
    ex.extract_batch_device(dev.data_ptr(), 2, w, h, stride, h * stride)
    d_kp, d_desc, d_cnt, cap = ex.device_results()
    nbatch = (n - 1 + B - 2) // (B - 1)
    m12 = torch.zeros((nbatch, B - 1, cap), dtype=torch.int32, device="cuda")
    nm = torch.zeros((nbatch, B - 1), dtype=torch.int32, device="cuda")
    npairs = torch.zeros((nbatch, B - 1), dtype=torch.int32, device="cuda")
    bounds = (0.0, float(w), 0.0, float(h))

    def run():
        for b in range(nbatch):
            s0 = b * (B - 1)
            nb = min(B, n - s0)
            ex.extract_batch_device(dev[s0].data_ptr(), nb, w, h, stride, h * stride)
            m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, nb, 600, bounds, m12[b].data_ptr(),
                                                     nm[b].data_ptr(), npairs[b].data_ptr(), windowSize=100)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # the matcher alone on the last batch
    nb = min(B, n - (nbatch - 1) * (B - 1))
    e0.record(st)
    for _ in range(20):
        m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, nb, 600, bounds, m12[-1].data_ptr(),
                                                 nm[-1].data_ptr(), npairs[-1].data_ptr(), windowSize=100)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot_pairs = int(npairs.sum().item())
    return {"frames_per_s_extract_plus_match": round(n / dt, 1), "frame_pairs": n - 1,
            "candidate_pair_distances_per_s_pipeline": round(tot_pairs / dt, 1),
            "matcher_ms_per_launch": round(ms, 4), "matcher_pairs_per_launch": nb - 1,
            "matcher_frame_pairs_per_s": round((nb - 1) / (ms * 1e-3), 1),
            "mean_matches_per_pair": round(float(nm.sum().item()) / (n - 1), 1)}


def device_resident_overlapped(frames, w, h):
    """The same work as a pipeline: two extractor handles take the batches in turn, each on its own stream, and
    the matcher runs on a third stream behind the extraction it reads; a handle starts its next batch only after
    the matcher has read its results.  The matcher's 0.35 ms of latency-bound walking then runs beside the next
    batch's kernels."""
    import torch
    n = len(frames)
    stride = (w + 63) // 64 * 64
    host = np.zeros((n, h, stride), np.uint8)
    for t, f in enumerate(frames):
        host[t, :, :w] = f
    dev = torch.from_numpy(host).cuda()
    B = min(256, n)
    nbatch = (n - 1 + B - 2) // (B - 1)
    exs, sts, res = [], [], []
    for _ in range(2):
        st = torch.cuda.Stream()
        ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
        ex.set_stream(st.cuda_stream)
        ex.extract_batch_device(dev.data_ptr(), 2, w, h, stride, h * stride)
        exs.append(ex), sts.append(st), res.append(ex.device_results())
    sm = torch.cuda.Stream()
    m = orbx.ORBmatcher(0.9, True)
    m.set_stream(sm.cuda_stream)
    cap = res[0][3]
    m12 = torch.zeros((nbatch, B - 1, cap), dtype=torch.int32, device="cuda")
    nm = torch.zeros((nbatch, B - 1), dtype=torch.int32, device="cuda")
    npairs = torch.zeros((nbatch, B - 1), dtype=torch.int32, device="cuda")
    bounds = (0.0, float(w), 0.0, float(h))
    read = [None, None]  # per handle: the event after the matcher that read its last batch

    def run():
        for b in range(nbatch):
            k = b & 1
            s0 = b * (B - 1)
            nb = min(B, n - s0)
            if read[k] is not None:
                sts[k].wait_event(read[k])
            exs[k].extract_batch_device(dev[s0].data_ptr(), nb, w, h, stride, h * stride)
            done = torch.cuda.Event()
            done.record(sts[k])
            sm.wait_event(done)
            d_kp, d_desc, d_cnt, _ = res[k]
            m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, nb, 600, bounds, m12[b].data_ptr(),
                                                     nm[b].data_ptr(), npairs[b].data_ptr(), windowSize=100)
            read[k] = torch.cuda.Event()
            read[k].record(sm)

    run()
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"frames_per_s_extract_plus_match": round(n / dt, 1), "frame_pairs": n - 1,
            "mean_matches_per_pair": round(float(nm.sum().item()) / (n - 1), 1),
            "pipeline": "two extractor handles in turn, matcher on its own stream"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1021)
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
    n_host = min(n, 256)  # the host-side path is slow: a bounded sample
    t0 = time.perf_counter()
    kd = []
    for s in range(0, n_host, 64):
        chunk = frames[s:min(s + 64, n_host)]
        ex.extract_batch_host(chunk)
        for f in range(len(chunk)):
            kd.append(ex.batch_download(f))
    t_extract = time.perf_counter() - t0
    t0 = time.perf_counter()
    pairs = matches = 0
    for t in range(n_host - 1):
        (k1, d1), (k2, d2) = kd[t], kd[t + 1]
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
        nm, _, np_ = m.SearchForInitialization(k1, d1, k2, d2, bounds, prev, 100)
        pairs += np_
        matches += nm
    t_match = time.perf_counter() - t0
    gpu = device_resident(frames, w, h)
    gpu2 = device_resident_overlapped(frames, w, h)
    print(json.dumps({"device_resident": gpu, "device_resident_overlapped": gpu2, "workload": "consecutive-frame initialisation matching, 1241x376, 2000 kp",
                      "frames": n, "host_path": {
                          "frames": n_host, "extract_frames_per_s_host_to_host": round(n_host / t_extract, 1),
                          "frame_pairs_per_s": round((n_host - 1) / t_match, 1),
                          "candidate_pair_distances_per_s": round(pairs / t_match, 1),
                          "frames_per_s_extract_plus_match": round(n_host / (t_extract + t_match), 1),
                          "mean_matches_per_pair": round(matches / (n_host - 1), 1),
                          "mean_candidate_pairs_per_frame_pair": round(pairs / (n_host - 1), 1)}}))


if __name__ == "__main__":
    main()
