#!/usr/bin/env python3
"""Stereo frames per second: extraction of the left and the right image on two extractor handles
(two streams) followed by Frame::ComputeStereoMatches on the GPU, KITTI shape, frames resident in
HBM.  The right images are the left ones shifted by a band-wise disparity.  Prints one JSON line.
Usage: python tools/stereo_rate.py [--pairs 128]
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
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    args = ap.parse_args()
    w, h, n = 1241, 376, args.pairs
    stride = (w + 63) // 64 * 64
    hl = np.zeros((n, h, stride), np.uint8)
    hr = np.zeros((n, h, stride), np.uint8)
    rng = np.random.default_rng(4)
    for f in range(n):
        wide = orbx.synth_image(0x0B5EED00 + f, w + 160, h, 0)
        hl[f, :, :w] = wide[:, 48:48 + w]
        d0, d1 = 5 + f % 40, 60 - f % 50
        hr[f, :h // 2, :w] = wide[:h // 2, 48 + d0:48 + d0 + w]
        hr[f, h // 2:, :w] = wide[h // 2:, 48 + d1:48 + d1 + w]
        hr[f, :, :w] = np.clip(hr[f, :, :w].astype(np.int16) + rng.integers(-2, 3, size=(h, w)), 0, 255)
    dl, dr = torch.from_numpy(hl).cuda(), torch.from_numpy(hr).cuda()
    sl, sr = torch.cuda.Stream(), torch.cuda.Stream()
    exl = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n)
    exr = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n)
    exl.set_stream(sl.cuda_stream)
    exr.set_stream(sr.cuda_stream)
    cap = exl.max_keypoints
    ur = torch.zeros((n, cap), dtype=torch.float32, device="cuda")
    dp = torch.zeros((n, cap), dtype=torch.float32, device="cuda")
    nm = torch.zeros(n, dtype=torch.int32, device="cuda")
    mb, mbf = 0.537, 386.1448

    def step(match=True):
        exl.extract_batch_device(dl.data_ptr(), n, w, h, stride, h * stride)
        exr.extract_batch_device(dr.data_ptr(), n, w, h, stride, h * stride)
        if match:
            orbx.compute_stereo_matches_device(exl, exr, mb, mbf, ur.data_ptr(), dp.data_ptr(), nm.data_ptr())

    out = {"workload": "stereo frames 2 x 1241x376, 2000 kp per image", "pairs_per_step": n}
    for name, match in (("extract_only", False), ("extract_and_match", True)):
        for _ in range(3):
            step(match)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            step(match)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + "_stereo_frames_per_s"] = round(n / dt, 1)
        out[name + "_ms_per_step"] = round(dt * 1e3, 4)
    # the matcher alone (three kernels) on the resident batch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(sl)
    for _ in range(20):
        orbx.compute_stereo_matches_device(exl, exr, mb, mbf, ur.data_ptr(), dp.data_ptr(), nm.data_ptr())
    e1.record(sl)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out["matcher_ms_per_step"] = round(ms, 4)
    out["matcher_stereo_frames_per_s"] = round(n / (ms * 1e-3), 1)
    out["mean_matches_per_frame"] = round(float(nm.float().mean().item()), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
