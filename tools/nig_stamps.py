#!/usr/bin/env python3
"""Timeline of nig_bwd_kernel (workgroup (0,0), thread 0) inside a real train step, from the diagnostic library."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib, build, synth  # noqa: E402

lib = C.CDLL(os.path.join(build.PKG_DIR, "libmmdeer_stamps.so"))
for name, res, args in _lib.SYMBOLS:
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
_lib._LIB = lib                                  # the model below runs on the diagnostic build
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4096"))
m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=1)).to(dev).train()
d = synth.make_batch(B, seed=1)
a, v, t, y = (torch.from_numpy(d[k]).to(dev) for k in ("audio", "video", "text", "targets"))
for _ in range(3):
    m.train_step(a.bfloat16(), v.bfloat16(), t.bfloat16(), y)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.mmdeer_debug_nig_stamps.restype = C.c_int
assert lib.mmdeer_debug_nig_stamps(out) == 0
s = list(out)
names = ["entry -> independent loads issued + targets", "statistics reduced + finals (compute_finals)", "loss terms + gradient (lgamma, digamma)",
         "dE, dz2 row, LDS staging", "weight-gradient partial loop + stores"]
for i, n in enumerate(names):
    print(f"{n:60s} {s[i + 1] - s[i]:7d} cycles")
print(f"{'total':60s} {s[5] - s[0]:7d} cycles")
print(f"inside compute_finals: 105-sum loop {s[8] - s[1]} | barrier {s[9] - s[8]} | thread-0 finals {s[10] - s[9]} | barrier {s[2] - s[10]}")
