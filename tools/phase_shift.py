"""Does a phase shift between the extractor instances raise the overlapped throughput?  All instances run the
same kernel sequence, so started together they run in phase (four detection kernels at once, then four
distribution kernels, ...).  Instance i first extracts i*shift frames, which delays its chain against
the others; the rate is taken over the steps that follow, the shift included in the time.
usage: python3 tools/phase_shift.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbslam2_amd as orbx  # noqa: E402


def run(ninst, per, steps, shift):
    w, h, nfeat = 1241, 376, 2000
    stride = (w + 63) // 64 * 64
    batch = ninst * per
    host = np.zeros((batch, h, stride), np.uint8)
    for f in range(batch):
        host[f] = orbx.synth_image(0x0B5EED00 + f, w, h, 0, stride=stride)
    frames = torch.from_numpy(host).cuda()
    streams = [torch.cuda.Stream() for _ in range(ninst)]
    exs = [orbx.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=per) for _ in range(ninst)]
    for e, st in zip(exs, streams):
        e.set_stream(st.cuda_stream)

    def step():
        for i, e in enumerate(exs):
            e.extract_batch_device(frames[i * per:(i + 1) * per].data_ptr(), per, w, h, stride, h * stride)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    extra = 0
    for i, e in enumerate(exs):
        n = min(i * shift, per)
        if n:
            e.extract_batch_device(frames[i * per:i * per + n].data_ptr(), n, w, h, stride, h * stride)
            extra += n
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in exs:
        e.close()
    return (batch * steps + extra) / dt, batch * steps / dt


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    for ninst, per in ((4, 64), (2, 128), (8, 32)):
        for shift in (0, per // (2 * ninst), per // ninst, 2 * per // ninst):
            a, b = run(ninst, per, steps, shift)
            print("instances %d x %d frames, shift %3d frames per instance: %.0f frames/s (%.0f without counting the "
                  "shift frames)" % (ninst, per, shift, a, b), flush=True)
