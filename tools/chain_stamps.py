#!/usr/bin/env python3
"""Timeline of the two head chain launches (workgroup 0, thread 0) from s_memtime stamps (diagnostic library)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import _lib, build, synth  # noqa: E402

_lib._build.LIB_PATH = os.path.join(build.PKG_DIR, "libmmdeer_stamps.so")   # load the diagnostic library instead
_lib._build.needs_build = lambda: False
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.zeros(96, dtype=torch.int64, device=dev)
lib.mmdeer_debug_chain_stamps.argtypes = [C.c_void_p]
lib.mmdeer_debug_chain_stamps(st.data_ptr())
B = 4096
model = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=42)).to(dev).train()
d = synth.make_batch(B, seed=42)
a, v, t, y = (torch.from_numpy(d[k]).to(dev) for k in ("audio", "video", "text", "targets"))
a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
for _ in range(3):
    model.train_step(a, v, t, y)
torch.cuda.synchronize()
s = st.cpu().numpy()
for name, o in (("forward chain", 0), ("backward chain", 16)):
    x = s[o:o + 16]
    print(f"--- {name}: input tile {x[1]-x[0]}  " + "  ".join(f"L{l}: compute {x[2+2*l]-x[1+2*l]} barrier {x[3+2*l]-x[2+2*l]}" for l in range(4)) + f"  total {x[9]-x[0]}")

k = s[32:62].reshape(10, 3)
print("forward L0, wave 0, per K-step: [wait for slot | reads + 8 MFMA | (next step start - reload issue)]")
for t in range(9):
    print(f"   t={t}: wait {k[t,1]-k[t,0]}  compute {k[t,2]-k[t,1]}  reload issue {k[t+1,0]-k[t,2]}")
