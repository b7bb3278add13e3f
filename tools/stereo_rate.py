#!/usr/bin/env python3
"""Stereo frames per second: extraction of the left and the right image on two extractor handles
(two streams) followed by Frame::ComputeStereoMatches on the GPU, KITTI shape, frames resident in
HBM.  The right images are the left ones shifted by a band-wise disparity.  Prints one JSON line.
Usage: python tools/stereo_rate.py [--pairs 128] [--rigs 2]
--rigs: left/right extractor pairs that share the step's frames and take turns in one orbx_pipeline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"  # before the HIP runtime starts: a stream per extractor, each on its own queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--rigs", type=int, default=2)
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
    rigs = max(1, min(args.rigs, n))
    per = (n + rigs - 1) // rigs
    pipe = orbx.Pipeline(0) if rigs > 1 else None
    mb, mbf = 0.537, 386.1448
    R = []
    for k in range(rigs):
        lo, cnt = k * per, min(per, n - k * per)
        sl, sr = torch.cuda.Stream(), torch.cuda.Stream()
        exl = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=cnt)
        exr = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=cnt)
        exl.set_stream(sl.cuda_stream)
        exr.set_stream(sr.cuda_stream)
        if pipe is not None:
            exl.set_pipeline(pipe)
            exr.set_pipeline(pipe)
        cap = exl.max_keypoints
        R.append({"exl": exl, "exr": exr, "sl": sl, "sr": sr, "lo": lo, "n": cnt,
                  "ur": torch.zeros((cnt, cap), dtype=torch.float32, device="cuda"),
                  "dp": torch.zeros((cnt, cap), dtype=torch.float32, device="cuda"),
                  "nm": torch.zeros(cnt, dtype=torch.int32, device="cuda")})
    fb = h * stride

    def match(r):
        orbx.compute_stereo_matches_device(r["exl"], r["exr"], mb, mbf, r["ur"].data_ptr(), r["dp"].data_ptr(),
                                           r["nm"].data_ptr())

    def step(do_match=True):
        for r in R:
            r["exl"].extract_batch_device(dl.data_ptr() + r["lo"] * fb, r["n"], w, h, stride, fb)
            r["exr"].extract_batch_device(dr.data_ptr() + r["lo"] * fb, r["n"], w, h, stride, fb)
            if do_match:
                match(r)

    out = {"workload": "stereo frames 2 x 1241x376, 2000 kp per image", "pairs_per_step": n, "rigs": rigs}
    for name, with_match in (("extract_only", False), ("extract_and_match", True)):
        for _ in range(3):
            step(with_match)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            step(with_match)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + "_stereo_frames_per_s"] = round(n / dt, 1)
        out[name + "_ms_per_step"] = round(dt * 1e3, 4)
    # the matcher alone (three kernels) on the resident batch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for r in R:
            match(r)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    out["matcher_ms_per_step"] = round(ms, 4)
    out["matcher_stereo_frames_per_s"] = round(n / (ms * 1e-3), 1)
    out["mean_matches_per_frame"] = round(float(torch.cat([r["nm"] for r in R]).float().mean().item()), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
