#!/usr/bin/env python3
"""Timeline of the fused projection + attention kernel (tri_fused.hip) from in-kernel stamps (diagnostic library
libmmdeer_stamps.so; the product library carries no stamp).  Prints, for waves 0 and 4 of workgroup 0, the prologue, every
L / M phase of the K loop and the epilogue in shader cycles, and the per-workgroup begin / end times and clock."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmdeer import build  # noqa: E402

lib = C.CDLL(os.path.join(build.PKG_DIR, "libmmdeer_stamps.so"))
vp, ci, cf, u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
lib.mmdeer_pack_qkv_headmajor.argtypes = [vp] * 3
lib.mmdeer_trimodal_fused_fwd.argtypes = [vp] * 8 + [ci, ci, cf, u64, u64, vp]
lib.mmdeer_trimodal_fused_bwd.argtypes = [vp] * 6 + [ci, ci, cf, u64, u64, vp]
lib.mmdeer_debug_tf_stamps.argtypes = [vp]
lib.mmdeer_last_error.restype = C.c_char_p
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
x = torch.randn(2 * B, 512, device=dev).bfloat16()
w = torch.randn(1536, 512, device=dev) * 0.05
bias = torch.randn(1536, device=dev) * 0.1
whm = torch.empty(1536 * 512, dtype=torch.bfloat16, device=dev)
obar = torch.empty(B, 512, dtype=torch.bfloat16, device=dev)
probs = torch.empty(B, 8, 4, device=dev)
dob = torch.randn(B, 512, device=dev).bfloat16()
dqkv = torch.empty(2 * B, 1536, dtype=torch.bfloat16, device=dev)
st = torch.zeros(8192, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
assert lib.mmdeer_pack_qkv_headmajor(w.data_ptr(), whm.data_ptr(), s) == 0
lib.mmdeer_debug_tf_stamps(st.data_ptr())


def launch():
    if mode == "fwd":
        rc = lib.mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar.data_ptr(), probs.data_ptr(), None, None, None,
                                           B, 1, 0.3, 7, 1, s)
    else:
        rc = lib.mmdeer_trimodal_fused_bwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), dob.data_ptr(), probs.data_ptr(), dqkv.data_ptr(),
                                           B, 1, 0.3, 7, 1, s)
    assert rc == 0, lib.mmdeer_last_error()


lib.mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar.data_ptr(), probs.data_ptr(), None, None, None, B, 1, 0.3, 7, 1, s)
# keep the chip busy for a while (clock state), then take the stamps of the last launch
for _ in range(300):
    launch()
torch.cuda.synchronize()
v = st.cpu().numpy().astype(np.int64)
for half in (0, 1):
    t = v[64 * half:64 * half + 64]
    print(f"--- {mode} B={B} wave {4 * half}: issue done +{t[1]-t[0]}  loop start +{t[2]-t[0]}  loop end +{t[3]-t[0]}  after final barrier +{t[4]-t[0]}  "
          f"exchange done +{t[5]-t[0]}  epilogue math done +{t[6]-t[0]}  end +{t[7]-t[0]}")
    rows = []
    if not t[8:56].any():          # per-phase stamps exist only in the MMDEER_STAMPS_LOOP build (they perturb the loop)
        continue
    for kt in range(16):
        a, b, c = t[8 + 3 * kt: 11 + 3 * kt]
        nxt = t[8 + 3 * (kt + 1)] if kt < 15 else t[3]
        rows.append((kt, b - a, c - b, nxt - c))
    print("    kt: L issue+lgkm | L vmcnt+barrier | M (+barrier)")
    for kt, li, lw, m in rows:
        print(f"    {kt:2d}: {li:5d} {lw:5d} {m:5d}   step {li + lw + m}")
n = (2 * B + 255) // 256 * 8
rt = v[256:256 + 2 * n].reshape(n, 2)
ck = v[2304:2304 + 2 * n].reshape(n, 2)
t0 = rt[:, 0].min()
beg, end = (rt[:, 0] - t0) / 100.0, (rt[:, 1] - t0) / 100.0
clk = (ck[:, 1] - ck[:, 0]) / np.maximum(rt[:, 1] - rt[:, 0], 1) * 100.0   # MHz
print(f"{n} workgroups: begin min/med/max {beg.min():.2f}/{np.median(beg):.2f}/{beg.max():.2f} us; end {end.min():.2f}/{np.median(end):.2f}/{end.max():.2f} us; "
      f"duration min/med/max {(end-beg).min():.2f}/{np.median(end-beg):.2f}/{(end-beg).max():.2f} us; shader clock med {np.median(clk):.0f} MHz")
