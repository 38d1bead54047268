#!/usr/bin/env python3
"""20 graph replays of the Stack B training step (B = 4096 bf16) for rocprofv3 --kernel-trace --stats: which kernels a step is made of."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import stackb, synth  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype="bf16").to(dev).train()
b = synth.make_batch(B, seed=1)
a, v, t, y = (torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text", "targets"))
a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()      # bf16 feature blocks resident in HBM
if len(sys.argv) > 2 and sys.argv[2] == "fused":
    from mmdeer.optim import FlatAdamW
    opt = FlatAdamW(m, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
    graph = m.capture_train_step_fused(a, v, t, y)

    def rep():
        graph()
        opt.step()
else:
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5, capturable=True)
    rep = m.capture_train_step(a, v, t, y, optimizer=opt, max_grad_norm=1.0)
for _ in range(20):
    rep()
torch.cuda.synchronize()
print("done")
