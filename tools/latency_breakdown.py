#!/usr/bin/env python3
"""Where a single-frame orbx_extract call spends its time (upload, kernels, count read-back, download)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402

w, h = 1241, 376
img = orbx.synth_image(0x0B5EED00, w, h, 0)
ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
for _ in range(3):
    ex(img)


def t(fn, n=20):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


print("whole call          %.3f ms" % t(lambda: ex(img)))
print("enqueue batch(1)    %.3f ms" % t(lambda: ex.extract_batch_host([img])))
ex.batch_counts()
print("enqueue + counts    %.3f ms" % t(lambda: (ex.extract_batch_host([img]), ex.batch_counts())))
print("download            %.3f ms" % t(lambda: ex.batch_download(0)))
ex.profile(True)
for _ in range(10):
    ex.extract_batch_host([img])
ms, n = ex.stage_times()
print("device stages (ms per call):", {k: round(v / n, 4) for k, v in ms.items()})
