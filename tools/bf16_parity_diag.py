#!/usr/bin/env python3
"""Where the bf16 HIP step and the bf16-emulating oracle differ: error statistics at the feature checkpoints (LayerNorm
outputs: fp32 copies from the kernels, rounded here), the NIG outputs, the loss and every gradient tensor.
usage: bf16_parity_diag.py [B] [dropout]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from mmdeer.spec import DIM_NAMES, param_table  # noqa: E402
from oracle import deer_oracle as O  # noqa: E402
from tests.test_gpu_model import dump_masks  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
dev = "cuda:0"
b = {k: torch.from_numpy(v) for k, v in synth.make_batch(B, seed=42).items()}
m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=p, seed=43)).to(dev).train()
a, v, t = (b[k].to(dev).bfloat16() for k in ("audio", "video", "text"))
ld = m.train_step(a, v, t, b["targets"].to(dev), return_features=True)
masks = dump_masks(m, B, m._step) if p > 0 else None
P = O.to_params({k: w.detach().cpu() for k, w in m.state_dict().items()}, torch.float32, requires_grad=True)
fo, ho, ldo, grads = O.train_step(P, a.float().cpu(), v.float().cpu(), t.float().cpu(), b["targets"], masks=masks, p=p, emulate_bf16=True)
out = ld["_outputs"]


def stats(name, got, ref):
    d = (got.double() - ref.double()).abs().flatten()
    ulp = (d > 0).double().mean().item()
    print(f"{name:40s} max {d.max().item():.3e}  mean {d.mean().item():.3e}  p99.9 {np.quantile(d.numpy(), 0.999):.3e}  differing {ulp:.4f}  (ref max {ref.abs().max().item():.3f})")


r = lambda x: x.bfloat16().float()
for k, ok in (("audiovisual_features", "audiovisual_features"), ("trimodal_features", "trimodal_features"), ("fused_features", "fused_features")):
    stats(k + " (bf16)", r(out[k].cpu()), fo[ok].detach())
stats("trimodal_attention", out["trimodal_attention"].cpu(), fo["trimodal_attention_weights"].detach())
nig = out["_nig"].cpu()
for i, k in enumerate(("mu", "nu", "alpha", "beta")):
    stats(k, nig[i], torch.cat([ho[f"{d}_{k}"] for d in DIM_NAMES], dim=1).detach())
print("loss", float(ld["total_loss"]), float(ldo["total_loss"]))
named = dict(m.named_parameters())
for name, shape, _ in param_table():
    g, rr = named[name].grad.cpu().double(), grads[name].double()
    if float(rr.abs().max()) == 0:
        continue
    cos = float((g.flatten() @ rr.flatten()) / (g.norm() * rr.norm() + 1e-300))
    print(f"{name:70s} rel-max {float((g - rr).abs().max() / rr.abs().max()):.2e}  rel-l2 {float((g - rr).norm() / rr.norm()):.2e}  cos {cos:.6f}")
