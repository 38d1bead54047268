#!/usr/bin/env python3
"""Timeline of the 256x256 forward kernel (workgroup 0, wave 0) from s_memtime stamps (diagnostic library)."""
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


def run(tag, M, N, K, bias):
    dt = torch.bfloat16
    A = torch.randn(M, K, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    b = torch.randn(N, device=dev) if bias else None
    Cm = torch.empty(M, N, device=dev, dtype=dt)
    st = torch.zeros(1024, dtype=torch.int64, device=dev)
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.bias = b.data_ptr() if bias else None
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    a.tile = 3
    a.drop_site = -1
    a.regen_site = -1
    a.mask_scale = 1.0
    a.debug = st.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.mmdeer_gemm(C.byref(a)) == 0, lib.mmdeer_last_error()
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    print(f"--- {tag} M={M} N={N} K={K}: setup+prologue issue={s[1]-s[0]} prologue wait+loop={s[2]-s[1]} epilogue={s[3]-s[2]} total={s[3]-s[0]}")
    for kt in range(6):
        b0 = 8 + kt * 4
        if s[b0] == 0:
            break
        print(f"   kt={kt}: load phase (reads, dma issue, waits, barrier)={s[b0+1]-s[b0]} mfma phase={s[b0+2]-s[b0+1]} barrier={s[b0+3]-s[b0+2]} iter={s[b0+3]-s[b0]}")
    if s[8]:
        print(f"   prologue wait (start of kt=0 - end of prologue issue) = {s[8]-s[1]}")
    # per-workgroup begin / end on the 100 MHz real-time counter
    import numpy as np
    n = (M // 256) * (N // 192 if N % 192 == 0 else -(-N // 256))
    w = s[256:256 + 2 * n].reshape(n, 2).astype(np.int64)
    t0 = w[:, 0].min()
    beg, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0     # us
    print(f"   {n} workgroups: begin min/median/max = {beg.min():.2f}/{np.median(beg):.2f}/{beg.max():.2f} us, "
          f"end min/median/max = {end.min():.2f}/{np.median(end):.2f}/{end.max():.2f} us, "
          f"duration min/median/max = {(end-beg).min():.2f}/{np.median(end-beg):.2f}/{(end-beg).max():.2f} us")
    order = np.argsort(beg)
    print("   begin times (sorted, every 16th):", " ".join(f"{beg[i]:.2f}" for i in order[::16]))
    print("   end times of the same workgroups :", " ".join(f"{end[i]:.2f}" for i in order[::16]))


run("in_proj", 8192, 1536, 512, True)
