#!/usr/bin/env python3
"""Stack B training step as it runs today: operator sequence issued from Python (eager), B = 4096 and 256, both dtypes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import stackb, synth  # noqa: E402

dev = torch.device("cuda:0")
for B, dtype in ((256, "fp32"), (4096, "fp32"), (256, "bf16"), (4096, "bf16")):
    m = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype=dtype).to(dev).train()
    b = synth.make_batch(B, seed=1)
    a, v, t, y = (torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text", "targets"))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.compute_loss(m(a, v, t), y)["total_loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"workload": f"Stack B train step (fwd with dropout + MultiTaskDEERLoss + bwd + clip + torch AdamW), B={B}, {dtype}, eager operator sequence",
                      "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1), "loss": round(float(loss), 5)}))
    # the same step captured into one HIP graph (capture_train_step with a capturable AdamW inside)
    m2 = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype=dtype).to(dev).train()
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-4, weight_decay=1e-5, capturable=True)
    rep = m2.capture_train_step(a, v, t, y, optimizer=opt2, max_grad_norm=1.0)
    for _ in range(5):
        rep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        ld = rep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"workload": f"the same, HIP-graph replay (capture_train_step), B={B}, {dtype}",
                      "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1), "loss": round(float(ld["total_loss"]), 5)}))
