#!/usr/bin/env python3
"""In-wave window of the small LDS-DMA GEMMs: per-workgroup begin / end on the 100 MHz real-time counter (diagnostic
library), to set against the rocprofv3 duration of the same launch -- the difference is time in which no wave runs
(dispatch, end-of-kernel cache write-back)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmdeer import _lib, build  # noqa: E402

lib = C.CDLL(os.path.join(build.PKG_DIR, "libmmdeer_stamps.so"))
lib.mmdeer_gemm.restype = C.c_int
lib.mmdeer_gemm.argtypes = [C.POINTER(_lib.GemmArgs)]
lib.mmdeer_last_error.restype = C.c_char_p
dev = torch.device("cuda:0")


def run(tag, M, N, K, tile, bm, bn):
    dt = torch.bfloat16
    A = torch.randn(M, K, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    b = torch.randn(N, device=dev)
    Cm = torch.empty(M, N, device=dev, dtype=dt)
    st = torch.zeros(256 + 2048, dtype=torch.int64, device=dev)
    a = _lib.GemmArgs()
    a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), Cm.data_ptr(), b.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    a.tile, a.relu = tile, 1
    a.drop_site = a.regen_site = -1
    a.mask_scale = 1.0
    a.debug = st.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        assert lib.mmdeer_gemm(C.byref(a)) == 0, lib.mmdeer_last_error()
    torch.cuda.synchronize()
    n = min(1024, (-(-M // bm)) * (-(-N // bn)))
    w = st.cpu().numpy()[256:256 + 2 * n].reshape(n, 2).astype(np.int64)
    t0 = w[:, 0].min()
    beg, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
    print(f"{tag} M={M} N={N} K={K} ({n} workgroups): begin median/max = {np.median(beg):.2f}/{beg.max():.2f} us, "
          f"workgroup duration median/max = {np.median(end - beg):.2f}/{(end - beg).max():.2f} us, window = {end.max():.2f} us", flush=True)


run("128x64 (8 waves)", 4096, 512, 512, 1, 128, 64)
run("64x64", 4096, 256, 256, 0, 64, 64)
run("64x64", 4096, 256, 512, 0, 64, 64)
run("128x64 (4 waves)", 8192, 512, 1536, 1, 128, 64)
