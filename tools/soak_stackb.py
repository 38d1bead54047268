#!/usr/bin/env python3
"""Soak, Stack B: a few thousand replayed training steps of the layer-chain plan with FlatAdamW; loss stays finite and falls, memory
stays flat, and a second model stepped through the launch-by-launch plan on the same masks follows the same loss curve."""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import stackb, synth  # noqa: E402
from mmdeer.optim import FlatAdamW  # noqa: E402

dev = torch.device("cuda:0")
B = 4096
m = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype="bf16").to(dev).train()
twin = copy.deepcopy(m)
twin.train_plan = "ops"
d = synth.make_batch(B, seed=3)
a, v, t, y = (torch.from_numpy(d[k]).to(dev) for k in ("audio", "video", "text", "targets"))
a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
runs = []
for model in (m, twin):
    opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
    replay = model.capture_train_step_fused(a, v, t, y)
    opt.step()
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    N = 3000 if model is m else 300
    curve = []
    for i in range(N):
        ld = replay()
        opt.step()
        if i % 100 == 0 or i == N - 1:
            curve.append(float(ld["total_loss"]))
            assert curve[-1] == curve[-1]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'chains' if model is m else 'launch by launch'}: {N} steps in {dt:.2f} s = {dt / N * 1e3:.3f} ms per step (fwd + bwd + optimiser); "
          f"memory delta {torch.cuda.memory_allocated() - mem0} B; loss {curve[0]:.4f} -> {curve[-1]:.4f}", flush=True)
    assert curve[-1] < curve[0] and (model is twin or torch.cuda.memory_allocated() == mem0)      # (the twin's run frees the first graph's pool)
    runs.append(curve)
same = [abs(x - z) for x, z in zip(runs[0][:3], runs[1][:3])]
print("first 300 steps, chains against launch by launch, |loss difference| at steps 0 / 100 / 200:", same)
assert same[0] == 0.0 and max(same) < 5e-3
