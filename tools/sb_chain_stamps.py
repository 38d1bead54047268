"""Cycle stamps of workgroup 0 of ONE Stack B layer chain (an encoder's forward: stem + 3 residual blocks + output projection, 256-wide
layers) from the diagnostic library libmmdeer_stamps.so -- where the time of a chain of SMALL layers goes.
usage: python tools/sb_chain_stamps.py [batch] [modality 0|1|2]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmdeer import build
build.LIB_PATH = os.path.join(build.PKG_DIR, "libmmdeer_stamps.so")   # the diagnostic build of the same sources
build.needs_build = lambda: False
from mmdeer import _lib, synth, chainops  # noqa: E402
from mmdeer.stackb import CompleteDEERModel, ModelConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
m = CompleteDEERModel(ModelConfig(), compute_dtype="bf16").to(dev).train()
b = synth.make_batch(B, seed=2)
a, v, t, y = (torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text", "targets"))
stamps = torch.zeros(512, dtype=torch.int64, device=dev)
orig = chainops.Chain.launch
count = [0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1

def launch(self):
    if count[0] == which:
        self.a.debug = stamps.data_ptr()
    count[0] += 1
    orig(self)

for _ in range(2):
    m.train_step_fused(a, v, t, y)
torch.cuda.synchronize()
chainops.Chain.launch = launch
m.train_step_fused(a, v, t, y)
torch.cuda.synchronize()
raw = stamps.cpu().numpy()
t0 = raw[0]
print(f"chain launch #{which} of the step, B = {B}")
print("stamp  cycles-from-start  delta   (1: tables in LDS, 2: prologue done; 3+3s: segment s decoded, 4+3s: its tiles done, 5+3s: its layer end done; 130+4s / 131+4s: layer-end barrier passed / body done; 200+8s+w: wave w at the barrier)")
prev = t0
for i, x in enumerate(raw[:512]):
    if x == 0 or abs(int(x) - int(t0)) > 10**9:
        continue
    print(f"{i:3d} {x - t0:10d} {x - prev:8d}")
    prev = x
