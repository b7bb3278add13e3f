#!/usr/bin/env python3
"""Map points per second of the batched MapPoint::ComputeDistinctiveDescriptors kernel: 200,000 map
points with 2..40 observations each, descriptors resident in HBM.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402
from orbslam2_amd import _lib as L  # noqa: E402


def main():
    import torch
    npts = 200_000
    rng = np.random.default_rng(1)
    n = rng.integers(2, 41, size=npts)
    start = np.zeros(npts + 1, np.int32)
    start[1:] = np.cumsum(n)
    rows = orbx.synth_descriptors(3, int(start[-1]))
    d_rows, d_start = torch.from_numpy(rows).cuda(), torch.from_numpy(start).cuda()
    best = torch.zeros(npts, dtype=torch.int32, device="cuda")
    out = torch.zeros((npts, 32), dtype=torch.uint8, device="cuda")
    m = orbx.ORBmatcher()
    st = torch.cuda.Stream()
    m.set_stream(st.cuda_stream)

    def run():
        L.check(L.lib().orbx_distinctive_descriptors_device(m._handle(), d_rows.data_ptr(), d_start.data_ptr(), npts,
                                                            best.data_ptr(), out.data_ptr()), "distinctive")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    pairs = float((n.astype(np.int64) ** 2).sum())
    print(json.dumps({"workload": "distinctive descriptor of 200000 map points, 2..40 observations",
                      "ms_per_call": round(ms, 4), "map_points_per_s": round(npts / (ms * 1e-3), 1),
                      "descriptor_distances_per_s": round(pairs / (ms * 1e-3), 1)}))


if __name__ == "__main__":
    main()
