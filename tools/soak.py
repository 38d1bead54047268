#!/usr/bin/env python3
"""Soak: a few thousand replayed training steps with the fused optimiser; loss stays finite, memory stays flat."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from mmdeer.optim import FusedAdamW  # noqa: E402

dev = torch.device("cuda:0")
B = 4096
m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=1)).to(dev).train()
d = synth.make_batch(B, seed=3)
a, v, t, y = (torch.from_numpy(d[k]).to(dev) for k in ("audio", "video", "text", "targets"))
a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
opt = FusedAdamW(m, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
replay = m.capture_train_step(a, v, t, y)
opt.step()
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
first = last = None
N = 3000
for i in range(N):
    ld = replay()
    opt.step()
    if i % 500 == 0 or i == N - 1:
        l = float(ld["total_loss"])
        first = l if first is None else first
        last = l
        print(f"step {i:5d}  loss {l:.5f}  grad_norm {float(opt.last_grad_norm):.4f}  mem {torch.cuda.memory_allocated() / 2**20:.0f} MiB", flush=True)
        assert l == l
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{N} steps in {dt:.2f} s = {dt / N * 1e3:.3f} ms per step (fwd + bwd + optimiser); memory delta {torch.cuda.memory_allocated() - mem0} B; loss {first:.4f} -> {last:.4f}")
assert torch.cuda.memory_allocated() == mem0 and last < first
