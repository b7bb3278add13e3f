"""Phase times inside k_describe from a measurement build of the library (-DDESC_STAMPS):
python tools/desc_stamps.py <path to that liborbx.so>.  Prints the mean time between the stamps of the
sampled workgroups in microseconds (the constant-rate counter runs at 100 MHz)."""
import ctypes as C
import shutil
import sys
import os

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
if len(sys.argv) > 1:
    shutil.copy(sys.argv[1], os.path.join(root, "orbslam2_amd", "liborbx.so"))
import torch
import orbslam2_amd as orbx
from orbslam2_amd import _lib

w, h, n = 1241, 376, 64
host = np.stack([orbx.synth_image(0x0B5EED00 + f, w, h, 0) for f in range(n)])
frames = torch.from_numpy(host).cuda()
ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, device=0)
lib = _lib.lib()
lib.orbx_debug_desc_stamps.argtypes = [C.c_void_p]
lib.orbx_debug_desc_stamps.restype = C.c_int
buf = np.zeros((1024, 8), dtype=np.uint64)
for rep in range(4):
    ex.extract_batch_device(frames.data_ptr(), n, w, h, w, w * h)
    torch.cuda.synchronize()
    k = lib.orbx_debug_desc_stamps(buf.ctypes.data)
d = buf[:k].astype(np.int64)
names = ["nkept+barrier", "slots+barrier", "patches+moments", "reduce+barrier", "angle+barrier", "windows+tests", "records"]
per = (d[:, 1:] - d[:, :-1]) / 100.0
print("workgroups sampled:", k)
for i, nm in enumerate(names):
    print("%-18s %8.2f us (median %.2f)" % (nm, per[:, i].mean(), np.median(per[:, i])))
print("%-18s %8.2f us" % ("workgroup total", ((d[:, 7] - d[:, 0]) / 100.0).mean()))
print("kernel span     %8.2f us" % ((d[:, 7].max() - d[:, 0].min()) / 100.0))
