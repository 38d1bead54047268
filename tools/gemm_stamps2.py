#!/usr/bin/env python3
"""Whole-kernel timeline of the LDS-DMA NT kernel from s_memtime stamps (diagnostic library only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib, build  # noqa: E402

lib = C.CDLL(os.path.join(build.PKG_DIR, "libmmdeer_stamps.so"))
lib.mmdeer_gemm.restype = C.c_int
lib.mmdeer_gemm.argtypes = [C.POINTER(_lib.GemmArgs)]
lib.mmdeer_last_error.restype = C.c_char_p
dev = torch.device("cuda:0")


def run(tag, M, N, K, tile, bias):
    dt = torch.bfloat16
    A = torch.randn(M, K, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    b = torch.randn(N, device=dev) if bias else None
    Cm = torch.empty(M, N, device=dev, dtype=dt)
    st = torch.zeros(4096, dtype=torch.int64, device=dev)   # slots [0,16): workgroup 0 phases; [256, ...): per-workgroup begin / end
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.bias = b.data_ptr() if bias else None
    a.relu = int(bias)
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    a.tile = tile
    a.drop_site = 4 if bias else -1
    a.dropout_p = 0.3
    a.regen_site = -1
    a.mask_scale = 1.0
    a.debug = st.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.mmdeer_gemm(C.byref(a)) == 0, lib.mmdeer_last_error()
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    names = [("setup (descriptor, pointers)", 0, 1), ("prologue issue", 1, 2), ("first tile wait+barrier", 2, 3), ("iter0", 3, 8),
             ("iter1", 8, 9), ("rest of loop", 9, 4), ("epilogue", 4, 5), ("TOTAL", 0, 5)]
    print(f"--- {tag} M={M} N={N} K={K} tile={tile} bias/relu/dropout={bias}: " + "  ".join(f"{n}={int(s[j] - s[i])}" for n, i, j in names))


run("fwd512", 4096, 512, 512, 0, False)
run("fwd512", 4096, 512, 512, 0, True)
run("fwd512 128x64", 4096, 512, 512, 1, True)
run("in_proj 128x64", 8192, 1536, 512, 1, False)
