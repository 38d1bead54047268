#!/usr/bin/env python3
"""Stack B training step, fused form (train_step_fused + FlatAdamW): eager and as a HIP graph, against the autograd form."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import stackb, synth  # noqa: E402
from mmdeer.optim import FlatAdamW  # noqa: E402

dev = torch.device("cuda:0")
for B, dtype in ((4096, "bf16"), (256, "bf16"), (4096, "fp32")):
    m = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype=dtype).to(dev).train()
    m.train_plan = os.environ.get("SB_PLAN", "auto")       # "ops": the launch-by-launch sequence instead of the layer chains
    b = synth.make_batch(B, seed=1)
    a, v, t, y = (torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text", "targets"))
    if dtype == "bf16":          # bf16 feature blocks resident in HBM, as bench.py's Stack C line
        a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
    opt = FlatAdamW(m, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
    for _ in range(3):
        ld = m.train_step_fused(a, v, t, y)
        opt.step()
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        ld = m.train_step_fused(a, v, t, y)
        opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"workload": f"Stack B fused train step (library launches only + FlatAdamW), B={B}, {dtype}, eager", "ms_per_step": round(dt * 1e3, 3),
                      "samples_per_s": round(B / dt, 1), "loss": round(float(ld["total_loss"]), 5)}), flush=True)
    rep = m.capture_train_step_fused(a, v, t, y)
    for _ in range(5):
        rep(); opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        ld = rep()
    torch.cuda.synchronize()
    dt_g = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        ld = rep(); opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"workload": f"the same as one HIP graph (fwd + loss + bwd) + eager FlatAdamW, B={B}, {dtype}", "ms_per_step": round(dt * 1e3, 3),
                      "ms_fwd_bwd_only": round(dt_g * 1e3, 3), "samples_per_s": round(B / dt, 1), "loss": round(float(ld["total_loss"]), 5)}), flush=True)
