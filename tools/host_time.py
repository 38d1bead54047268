#!/usr/bin/env python3
"""Host-side cost of one train_step (enqueue only) vs its GPU time."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from mmdeer.optim import FusedAdamW  # noqa: E402

dev = torch.device("cuda:0")
B = 4096
model = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=42)).to(dev).train()
d = synth.make_batch(B, seed=42)
a, v, t, y = (torch.from_numpy(d[k]).to(dev) for k in ("audio", "video", "text", "targets"))
a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
opt = FusedAdamW(model, lr=1e-4)
for _ in range(10):
    model.train_step(a, v, t, y); opt.step()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    model.train_step(a, v, t, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step: {(t1 - t0) / K * 1e6:.1f} us   wall per step incl. drain: {(t2 - t0) / K * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    model.train_step(a, v, t, y)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
