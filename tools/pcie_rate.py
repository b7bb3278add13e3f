#!/usr/bin/env python3
"""Host-to-host rate of the extractor: frames start in pageable host memory, keypoints and
descriptors end in host memory (upload, extraction, download).  Not the benchmark's `value`,
which starts with the frames resident in HBM; recorded in profiles/README.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402

w, h, nf, B = 1241, 376, 2000, 64
imgs = [orbx.synth_image(0x0B5EED00 + i, w, h, 0) for i in range(B)]
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
for _ in range(2):
    ex.extract_batch_host(imgs)
    ex.batch_counts()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    ex.extract_batch_host(imgs)
    n, _ = ex.batch_counts()
    for f in range(B):
        ex.batch_download(f)
dt = time.perf_counter() - t0
print("host-to-host: %.0f frames/s (batch %d, per-frame download calls included)" % (reps * B / dt, B))
t0 = time.perf_counter()
for _ in range(reps):
    ex.extract_batch_host(imgs)
    out = ex.batch_download_all()
dt = time.perf_counter() - t0
print("host-to-host: %.0f frames/s (batch %d, one download call per batch)" % (reps * B / dt, B))
import torch  # noqa: E402  (page-locked frames: the copy engine reads them in place)

pin = torch.empty((B, h, w), dtype=torch.uint8).pin_memory()
pinned = [pin[i].numpy() for i in range(B)]
for i in range(B):
    pinned[i][:] = imgs[i]
ex.extract_batch_host(pinned)
ref = ex.batch_download_all()
t0 = time.perf_counter()
for _ in range(reps):
    ex.extract_batch_host(pinned)
    out2 = ex.batch_download_all()
dt = time.perf_counter() - t0
same = all(np.array_equal(a[1], b[1]) for a, b in zip(out, out2))
print("host-to-host: %.0f frames/s (batch %d, frames in page-locked memory, one download call per batch; "
      "descriptors equal to the pageable run: %s)" % (reps * B / dt, B, same))
t0 = time.perf_counter()
for _ in range(reps * 4):
    ex(imgs[0])
dt = time.perf_counter() - t0
print("single-frame operator(): %.2f ms per call" % (dt / (reps * 4) * 1e3))
