#!/usr/bin/env python3
"""Large-batch sanity: one fused train step at B = 65536 (fp32 and bf16) -- loss against the CPU oracle in fp32, finite
gradients, and a bit-identical rerun; then B = 262144 in bf16 for index arithmetic (finite, reproducible)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402

DEV = "cuda:0"


def run(B, dtype, check_oracle):
    m = MultimodalDEER(ModelConfig(compute_dtype=dtype, dropout=0.0, seed=5)).to(DEV).train()
    b = synth.make_batch(B, seed=9)
    a, v, t, y = (torch.from_numpy(b[k]).to(DEV) for k in ("audio", "video", "text", "targets"))
    ld = m.train_step(a, v, t, y)
    g1 = m.flat_grad().clone()
    l1 = float(ld["total_loss"])
    ld = m.train_step(a, v, t, y)
    torch.cuda.synchronize()
    same = torch.equal(g1, m.flat_grad()) and l1 == float(ld["total_loss"])
    finite = bool(torch.isfinite(g1).all())
    msg = f"B={B} {dtype}: loss {l1:.6f} grad-norm {float(g1.norm()):.5f} finite={finite} rerun-identical={same} bins={int(ld['ece_bin_counts'].sum())}"
    if check_oracle:
        from oracle import deer_oracle as O   # checker only
        P = O.to_params({k: p.detach().cpu() for k, p in m.state_dict().items()})
        t0 = time.time()
        with torch.no_grad():
            fo, ho = O.model_forward(P, *(torch.from_numpy(b[k]) for k in ("audio", "video", "text")))
            pred = {f"{d}_{k}": ho[f"{d}_{k}"] for d in ("valence", "arousal", "dominance") for k in ("mu", "nu", "alpha", "beta")}
            ref = float(O.multitask_loss(pred, torch.from_numpy(b["targets"]))["total_loss"])
        msg += f" | oracle loss {ref:.6f} (|d|={abs(ref - l1):.2e}, {time.time() - t0:.1f} s on the CPU)"
    print(msg, flush=True)
    assert finite and same and int(ld["ece_bin_counts"].sum()) == 3 * B


run(65536, "fp32", True)
run(65536, "bf16", False)
run(262144, "bf16", False)
