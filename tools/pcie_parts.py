#!/usr/bin/env python3
"""Parts of the host-to-host path of a 64-frame batch: upload + extraction from pageable and from
page-locked frames, and the whole-batch download."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx
import torch
w, h, nf, B = 1241, 376, 2000, 64
imgs = [orbx.synth_image(0x0B5EED00 + i, w, h, 0) for i in range(B)]
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
pin = torch.empty((B, h, w), dtype=torch.uint8).pin_memory()
pinned = [pin[i].numpy() for i in range(B)]
for i in range(B):
    pinned[i][:] = imgs[i]
for name, src in (("pageable", imgs), ("pinned", pinned)):
    ex.extract_batch_host(src); ex.batch_counts()
    t0 = time.perf_counter()
    for _ in range(10):
        ex.extract_batch_host(src); ex.batch_counts()
    print(name, "extract+sync %.2f ms per batch" % ((time.perf_counter() - t0) / 10 * 1e3))
t0 = time.perf_counter()
for _ in range(10):
    out = ex.batch_download_all()
print("download_all %.2f ms per batch" % ((time.perf_counter() - t0) / 10 * 1e3))
