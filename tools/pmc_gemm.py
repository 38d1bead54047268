#!/usr/bin/env python3
"""A handful of GEMM launches for rocprofv3 --pmc runs (3 launches per configuration, in this order)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = 4096
CONFIGS = [("fwd512_64x64", B, 512, 512, 0, 0, 0, 1), ("fwd_inproj_256x256", 2 * B, 1536, 512, 0, 0, 3, 1),
           ("fwd_inproj_128x64", 2 * B, 1536, 512, 0, 0, 1, 1), ("dw_inproj_256x256_sk8", 1536, 512, 2 * B, 1, 1, 3, 8),
           ("dw512_256x256_sk4", 512, 512, B, 1, 1, 3, 4)]
for tag, M, N, K, ta, tw, tile, sk in CONFIGS:
    dt = torch.bfloat16
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(dt)
    W = torch.randn((K, N) if tw else (N, K), device=dev).to(dt)
    Cm = torch.empty(M, N, device=dev, dtype=torch.float32 if ta else dt)
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, A.shape[1], W.shape[1], N
    a.c_f32 = int(Cm.dtype == torch.float32)
    a.trans_a, a.trans_w, a.tile = ta, tw, tile
    a.drop_site = a.regen_site = -1
    a.mask_scale = 1.0
    if sk > 1:
        slab = torch.empty(sk * (M * N + M + 4), device=dev)
        a.splitk, a.slab = sk, slab.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(lib.mmdeer_gemm(C.byref(a)))
    torch.cuda.synchronize()
print("order:", [c[0] for c in CONFIGS])
