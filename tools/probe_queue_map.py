#!/usr/bin/env python3
"""Which hardware queue does the HIP runtime give the n-th stream of a process?  Creates streams in order,
runs one small kernel on each and prints nothing itself: run it under `rocprofv3 --kernel-trace` and read
Stream_Id / Queue_Id of the fill kernels in launch order (tools/collect: see profiles/r03/queue_map.txt).
usage: GPU_MAX_HW_QUEUES=4 rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/probe_queue_map.py 12 [destroy-first-k]"""
import ctypes
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
drop = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.init()
x = torch.zeros(1, device="cuda")  # the null stream's queue first
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
streams = []
for i in range(n):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0  # hipStreamNonBlocking
    streams.append(s)
for s in streams[:drop]:
    assert hip.hipStreamDestroy(s) == 0
more = []
for i in range(drop):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    more.append(s)
bufs = [torch.empty(1024 + i, device="cuda", dtype=torch.int32) for i in range(n + drop)]
for i, s in enumerate(streams[drop:] + more):
    with torch.cuda.stream(torch.cuda.ExternalStream(s.value)):
        bufs[i].fill_(i)  # the kernel's grid tells the streams apart: 1024 + i elements
torch.cuda.synchronize()
