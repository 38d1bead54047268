"""Cycle-counter samples of workgroup 0 of the layer-chain kernel (chain.hip) from the diagnostic library
libmmdeer_stamps.so (python -c "from mmdeer import build; build.build_stamps()"; the product library carries no stamp):
where the time of the chain goes -- prologue, every column tile, every layer end.
usage: python tools/chain_stamps.py [batch] [bwd]     (bwd: the backward chain, stamps of mmdeer_backward(phase = 1))"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmdeer import build
build.LIB_PATH = os.path.join(build.PKG_DIR, "libmmdeer_stamps.so")   # the diagnostic build of the same sources
build.needs_build = lambda: False
from mmdeer import _lib, synth  # noqa: E402
from mmdeer.model import MultimodalDEER, ModelConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
lib = _lib.load()
m = MultimodalDEER(ModelConfig(compute_dtype="bf16", seed=1)).to(dev).train()
b = synth.make_batch(B, seed=2)
a, v, t, y = (torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text", "targets"))
for _ in range(3):
    m.train_step(a, v, t, y)
torch.cuda.synchronize()
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
o = m._launch_forward(a, v, t, y, want_features=False)
if bwd:
    meta = o["_meta"]
    m._launch_backward(meta, meta["targets"], loss_out=torch.empty(20, device=dev), flat=torch.zeros_like(m.flat_grad()), want_views=False, phase=1)
torch.cuda.synchronize()
ws = m._workspace(B, torch.device(dev))
off = lib.mmdeer_workspace_offset(B, 0, b"davin" if bwd else b"slab")
raw = ws.view(torch.uint8)[off:off + 8 * 512].cpu().numpy().view(np.uint64).astype(np.int64)
t0 = raw[0]
print("(130+4s: segment s's layer-end barrier passed, 131+4s: its layer-end body done; 200+8s+w: wave w arrives at that barrier; 100-104: LayerNorm backward phases)")
print("stamp  cycles-from-start  delta   (1: tables in LDS, 2: prologue done; 3+3s: segment s decoded, 4+3s: its tiles done, 5+3s: its layer end done)")
prev = t0
for i, x in enumerate(raw[:512]):
    if x == 0 or abs(int(x) - int(t0)) > 10**9:
        continue
    print(f"{i:3d} {x - t0:10d} {x - prev:8d}")
    prev = x
