#!/usr/bin/env python3
"""Timeline of the 256x256 weight-gradient kernel (workgroup 0, wave 0) from s_memtime stamps (diagnostic library)."""
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


def run(tag, Bt, Nl, Kl, splitk):
    dt = torch.bfloat16
    dY = torch.randn(Bt, Nl, device=dev).to(dt)
    X = torch.randn(Bt, Kl, device=dev).to(dt)
    Cm = torch.zeros(Nl, Kl, device=dev, dtype=torch.float32)
    slab = torch.zeros(splitk, Nl * Kl + Nl, device=dev, dtype=torch.float32)
    db = torch.zeros(Nl, device=dev)
    st = torch.zeros(256, dtype=torch.int64, device=dev)
    a = _lib.GemmArgs()
    a.A, a.W, a.C = dY.data_ptr(), X.data_ptr(), Cm.data_ptr()
    a.bias_grad = db.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = Nl, Kl, Bt, Nl, Kl, Kl
    a.trans_a, a.trans_w = 1, 1
    a.c_f32 = 1
    a.tile = 3
    a.drop_site = -1
    a.regen_site = -1
    a.mask_scale = 1.0
    a.splitk = splitk
    a.slab = slab.data_ptr()
    a.debug = st.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.mmdeer_gemm(C.byref(a)) == 0, lib.mmdeer_last_error()
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    print(f"--- {tag} Bt={Bt} Nl={Nl} Kl={Kl} splitk={splitk}: prologue issue={s[1]-s[0]} loop={s[2]-s[1]} store issue={s[3]-s[2]} store drain={s[4]-s[3]} total={s[4]-s[0]}")
    for kt in range(8):
        b = 8 + kt * 5
        if s[b] == 0:
            break
        print(f"   kt={kt}: read issue={s[b+1]-s[b]} dma issue+waits+barrier={s[b+2]-s[b+1]} mfma phase={s[b+3]-s[b+2]} barrier={s[b+4]-s[b+3]} iter={s[b+4]-s[b]}")


run("512x512", 4096, 512, 512, 4)
run("in_proj", 8192, 1536, 512, 8)
run("256x256 alone", 4096, 256, 256, 4)
