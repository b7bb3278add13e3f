#!/usr/bin/env python3
"""Time of the device-batch SearchForInitialization kernel by search window (window 0: the set-up and the walk
without candidates), on 64 consecutive KITTI-shape frames.  usage: python3 tools/sfi_time.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402

w, h, B = 1241, 376, 64
base = orbx.synth_image(0x0B5EED00, w, h, 0)
rng = np.random.default_rng(7)
frames = []
for t in range(B):
    img = np.roll(base, (t % 5, 3 * t), axis=(0, 1)).astype(np.int16) + rng.integers(-2, 3, size=base.shape)
    frames.append(np.clip(img, 0, 255).astype(np.uint8))
ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
st = torch.cuda.Stream()
ex.set_stream(st.cuda_stream)
ex.extract_batch_host(frames)
d_kp, d_desc, d_cnt, cap = ex.device_results()
m = orbx.ORBmatcher(0.9, True)
m.set_stream(st.cuda_stream)
m12 = torch.zeros((B - 1, cap), dtype=torch.int32, device="cuda")
nm = torch.zeros(B - 1, dtype=torch.int32, device="cuda")
npairs = torch.zeros(B - 1, dtype=torch.int32, device="cuda")
bounds = (0.0, float(w), 0.0, float(h))
for win in (100, 30, 10, 0):
    def run():
        m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, B, 600, bounds, m12.data_ptr(), nm.data_ptr(),
                                                 npairs.data_ptr(), windowSize=win)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    print("window %3d: %.3f ms per launch of %d pairs, %.1f matches and %.0f candidate pairs per frame pair" % (
        win, e0.elapsed_time(e1) / 10, B - 1, float(nm.float().mean()), float(npairs.float().mean())))
