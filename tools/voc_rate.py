#!/usr/bin/env python3
"""Descriptor -> word quantisation rate: a full 10-ary vocabulary tree of depth 6 (1,111,111 nodes, the
shape of ORBvoc) with random node descriptors, 512,000 random descriptors resident in HBM.  Prints one
JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402
from orbslam2_amd import _lib as L  # noqa: E402


def main():
    import torch
    k, depth = 10, 6
    n = (k ** (depth + 1) - 1) // (k - 1)
    inner = (k ** depth - 1) // (k - 1)                 # nodes above the last level, in breadth-first order
    child_start = np.minimum(np.arange(n + 1, dtype=np.int64) * k, inner * k).astype(np.int32)
    children = np.arange(1, inner * k + 1, dtype=np.int32)
    word_id = np.full(n, -1, np.int32)
    word_id[inner:] = np.arange(n - inner)
    weight = np.where(word_id >= 0, 1.0, 0.0)
    desc = orbx.synth_descriptors(9, n)
    voc = orbx.ORBVocabulary(desc, child_start, children, word_id, weight, depth)
    nf = 512_000
    feats = torch.from_numpy(orbx.synth_descriptors(10, nf)).cuda()
    word = torch.zeros(nf, dtype=torch.int32, device="cuda")
    node = torch.zeros(nf, dtype=torch.int32, device="cuda")
    wt = torch.zeros(nf, dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()

    def run():
        L.check(L.lib().orbx_vocabulary_transform_device(voc._h, C.c_void_p(st.cuda_stream), feats.data_ptr(), nf, 4,
                                                         word.data_ptr(), node.data_ptr(), wt.data_ptr()), "transform")

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
    print(json.dumps({"workload": "vocabulary transform, 10-ary tree of depth 6, 512000 descriptors",
                      "ms_per_call": round(ms, 4), "descriptors_per_s": round(nf / (ms * 1e-3), 1),
                      "node_distances_per_s": round(nf * k * depth / (ms * 1e-3), 1)}))


if __name__ == "__main__":
    main()
