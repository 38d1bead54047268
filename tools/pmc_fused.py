#!/usr/bin/env python3
"""A handful of launches of the fused projection + attention kernels (tri_fused.hip) for rocprofv3 --pmc passes:
forward and backward at B = 4096 (configs[2]) with attention dropout on, through the C ABI of the product library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(2 * B, 512, device=dev).bfloat16()
w = torch.randn(1536, 512, device=dev) * 0.05
bias = torch.randn(1536, device=dev) * 0.1
whm = torch.empty(1536 * 512, dtype=torch.bfloat16, device=dev)
obar = torch.empty(B, 512, dtype=torch.bfloat16, device=dev)
probs = torch.empty(B, 8, 4, device=dev)
dob = torch.randn(B, 512, device=dev).bfloat16()
dqkv = torch.empty(2 * B, 1536, dtype=torch.bfloat16, device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mmdeer_pack_qkv_headmajor(w.data_ptr(), whm.data_ptr(), s))
for _ in range(6):
    _lib.check(lib.mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar.data_ptr(), probs.data_ptr(), None, None, None,
                                             B, 1, 0.3, 7, 1, s))
for _ in range(6):
    _lib.check(lib.mmdeer_trimodal_fused_bwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), dob.data_ptr(), probs.data_ptr(), dqkv.data_ptr(),
                                             B, 1, 0.3, 7, 1, s))
torch.cuda.synchronize()
print("done")
