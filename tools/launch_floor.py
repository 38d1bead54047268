#!/usr/bin/env python3
"""Floor of a chain of dependent launches on this box: N tiny kernels in one stream, eager and as a HIP graph."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
y = torch.zeros(1024, device=dev, dtype=torch.bfloat16)
N = 40


def chain():
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(N):
        lib.mmdeer_convert(x.data_ptr(), 1, y.data_ptr(), 0, 1024, s)


for _ in range(3):
    chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
for mode, fn in (("eager", chain), ("graph", g.replay)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 200
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"{mode}: {N} tiny dependent launches = {dt * 1e6:.1f} us -> {dt * 1e6 / N:.2f} us per launch")
