#!/usr/bin/env python3
"""Interleaved A/B timing of build variants of the fused projection + attention kernel IN ONE PROCESS (clock and box
differences between separate runs are larger than most kernel changes): usage  ab_fused.py <lib A> <lib B> [...] [--bwd] [--B n].
Each round launches every variant REPS times back to back between two events; prints median / min microseconds per launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
bwd = "--bwd" in sys.argv
B = int(sys.argv[sys.argv.index("--B") + 1]) if "--B" in sys.argv else 4096
if "--B" in sys.argv:
    args.remove(str(B))
vp, ci, cf, u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
libs = []
for path in args:
    lib = C.CDLL(os.path.abspath(path))
    lib.mmdeer_pack_qkv_headmajor.argtypes = [vp] * 3
    lib.mmdeer_trimodal_fused_fwd.argtypes = [vp] * 8 + [ci, ci, cf, u64, u64, vp]
    lib.mmdeer_trimodal_fused_bwd.argtypes = [vp] * 6 + [ci, ci, cf, u64, u64, vp]
    libs.append((os.path.basename(path), lib))
dev = torch.device("cuda:0")
x = torch.randn(2 * B, 512, device=dev).bfloat16()
w = torch.randn(1536, 512, device=dev) * 0.05
bias = torch.randn(1536, device=dev) * 0.1
whm = torch.empty(1536 * 512, dtype=torch.bfloat16, device=dev)
obar = torch.empty(B, 512, dtype=torch.bfloat16, device=dev)
probs = torch.empty(B, 8, 4, device=dev)
dob = torch.randn(B, 512, device=dev).bfloat16()
dqkv = torch.empty(2 * B, 1536, dtype=torch.bfloat16, device=dev)
s = torch.cuda.current_stream().cuda_stream
assert libs[0][1].mmdeer_pack_qkv_headmajor(w.data_ptr(), whm.data_ptr(), s) == 0


def launch(lib):
    if bwd:
        rc = lib.mmdeer_trimodal_fused_bwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), dob.data_ptr(), probs.data_ptr(), dqkv.data_ptr(), B, 1, 0.3, 7, 1, s)
    else:
        rc = lib.mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar.data_ptr(), probs.data_ptr(), None, None, None, B, 1, 0.3, 7, 1, s)
    assert rc == 0


libs[0][1].mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar.data_ptr(), probs.data_ptr(), None, None, None, B, 1, 0.3, 7, 1, s)
REPS, ROUNDS = 50, 15
res = {n: [] for n, _ in libs}
for _, lib in libs:
    for _ in range(100):
        launch(lib)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for n, lib in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            launch(lib)
        e1.record()
        torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) * 1e3 / REPS)
for n, _ in libs:
    v = np.array(res[n])
    print(f"{n:36s} {'bwd' if bwd else 'fwd'} B={B}: median {np.median(v):7.2f} us  min {v.min():7.2f} us  (back-to-back launches, incl. ~1.5 us launch gap)")
