#!/usr/bin/env python3
"""Micro-benchmark of the grouped GEMM on the layer shapes of the path (B=4096), per tile config.
Back-to-back launches between two events: accurate when the kernel is longer than the host launch cost;
run under `rocprofv3 --kernel-trace` for the short ones."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4096))
ITER = 30


def run(tag, M, N, K, trans_a, trans_w, dt, tile, splitk=1):
    f32 = dt == torch.float32
    A = torch.randn((K, M) if trans_a else (M, K), device=dev).to(dt)
    W = torch.randn((K, N) if trans_w else (N, K), device=dev).to(dt)
    Cm = torch.empty(M, N, device=dev, dtype=torch.float32 if (trans_a or f32) else dt)
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw, a.ldc = A.shape[1], W.shape[1], N
    a.a_f32 = a.w_f32 = int(f32)
    a.c_f32 = int(Cm.dtype == torch.float32)
    a.trans_a, a.trans_w, a.compute_f32, a.tile = trans_a, trans_w, int(f32), tile
    a.drop_site = a.regen_site = -1
    a.mask_scale = 1.0
    if splitk > 1:
        slab = torch.empty(splitk * ((M * N + M + 3) // 4 * 4), device=dev)
        a.splitk, a.slab = splitk, slab.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(lib.mmdeer_gemm(C.byref(a)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITER):
        lib.mmdeer_gemm(C.byref(a))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / ITER
    tf = 2.0 * M * N * K / us / 1e6
    tag = f"{tag} sk={splitk}"
    print(f"{tag:28s} M={M:5d} N={N:5d} K={K:5d} ta={trans_a} tw={trans_w} {str(dt)[6:]:8s} tile={tile}: {us:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)


shapes = [
    ("fwd in_proj", 2 * B, 1536, 512, 0, 0),
    ("fwd text_proj", B, 512, 768, 0, 0),
    ("fwd 512x512", B, 512, 512, 0, 0),
    ("fwd 256<-512", B, 256, 512, 0, 0),
    ("dX in_proj", 2 * B, 512, 1536, 0, 1),
    ("dX 512x512", B, 512, 512, 0, 1),
    ("dW in_proj", 1536, 512, 2 * B, 1, 1),
    ("dW 512x512", 512, 512, B, 1, 1),
    ("dW text", 512, 768, B, 1, 1),
]
for dt in (torch.bfloat16, torch.float32):
    for s in shapes:
        for tile in (0, 1, 2):
            if s[4]:   # weight-gradient shapes: with and without split-K (the timing includes the slab reduction)
                for sk in (1, 4, 8):
                    run(*s, dt, tile, sk)
            else:
                run(*s, dt, tile)
