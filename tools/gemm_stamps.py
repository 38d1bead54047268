#!/usr/bin/env python3
"""Per-phase cycle breakdown of the GEMM K loop from in-kernel s_memtime stamps (diagnostic library only)."""
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


def run(tag, M, N, K, ta, tw, tile, dt=torch.bfloat16):
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(dt)
    W = torch.randn((K, N) if tw else (N, K), device=dev).to(dt)
    Cm = torch.empty(M, N, device=dev, dtype=torch.float32 if ta else dt)
    st = torch.zeros(4096, dtype=torch.int64, device=dev)   # room for the per-workgroup slots [256, ...) of the LDS-DMA kernels
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, A.shape[1], W.shape[1], N
    a.a_f32 = a.w_f32 = int(dt == torch.float32)
    a.c_f32 = int(Cm.dtype == torch.float32)
    a.trans_a, a.trans_w, a.compute_f32, a.tile = ta, tw, int(dt == torch.float32), tile
    a.drop_site = a.regen_site = -1
    a.mask_scale = 1.0
    a.debug = st.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.mmdeer_gemm(C.byref(a)) == 0, lib.mmdeer_last_error()
    torch.cuda.synchronize()
    s = st.cpu().view(17, 8).numpy()
    names = ["issue loads", "lds read+mfma", "wait vmcnt", "lds store", "barrier", "loop back"]
    print(f"--- {tag}  M={M} N={N} K={K} ta={ta} tw={tw} tile={tile}")
    for it in range(2, 7):
        d = [int(s[it, j + 1] - s[it, j]) for j in range(5)] + [int(s[it + 1, 0] - s[it, 5])]
        print(f"  iter {it}: " + "  ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  | total={int(s[it + 1, 0] - s[it, 0])}")


run("dW lone workgroups", 512, 512, 4096, 1, 1, 0)
run("dW lone workgroups 128x128", 512, 512, 4096, 1, 1, 2)
run("fwd 512x512 (2 WG/CU)", 4096, 512, 512, 0, 0, 0)
run("fwd in_proj 128x128", 8192, 1536, 512, 0, 0, 2)
run("dX 512x512", 4096, 512, 512, 0, 1, 0)
