#!/usr/bin/env python3
"""A few EAGER training steps of the bench configuration (B = 4096 bf16, dropout 0.3) for rocprofv3 --pmc passes over every
kernel of the step (tools/gpu_pmc_step.sh): each launch is its own dispatch, so counters come per kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=42)).to(dev).train()
d = synth.make_batch(B, seed=42)
a, v, t = (torch.from_numpy(d[k]).to(dev).bfloat16() for k in ("audio", "video", "text"))
y = torch.from_numpy(d["targets"]).to(dev)
for _ in range(6):
    m.train_step(a, v, t, y)
torch.cuda.synchronize()
print("done")
