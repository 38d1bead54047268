"""Stack B fused training step: layer-chain plan against the launch-by-launch plan on the same model / batch / dropout step --
tape tensors, loss and gradients (tests/test_gpu_stackb.py holds the asserted version; this prints the differences)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mmdeer import synth                      # noqa: E402
from mmdeer.stackb import CompleteDEERModel, ModelConfig    # noqa: E402


def flat_tape(T, prefix=""):
    out = {}
    for k, v in T.items():
        if k in ("P", "ex", "xs", "attn_args", "cal_keep", "chain"):
            continue
        if torch.is_tensor(v):
            out[prefix + k] = v
        elif isinstance(v, dict):
            out.update(flat_tape(v, prefix + k + "."))
        elif isinstance(v, (list, tuple)):
            for i, w in enumerate(v):
                if torch.is_tensor(w):
                    out[f"{prefix}{k}[{i}]"] = w
                elif isinstance(w, dict):
                    out.update(flat_tape(w, f"{prefix}{k}[{i}]."))
                elif isinstance(w, (list, tuple)):
                    for j, u in enumerate(w):
                        if torch.is_tensor(u):
                            out[f"{prefix}{k}[{i}][{j}]"] = u
    return out


def main():
    dev = torch.device("cuda")
    for B in [int(a) for a in sys.argv[1:]] or [1024, 100, 4096]:
        torch.manual_seed(0)
        m1 = CompleteDEERModel(ModelConfig(), compute_dtype="bf16").to(dev)
        m1.train()
        m2 = copy.deepcopy(m1)
        m1.train_plan, m2.train_plan = "ops", "auto"
        b = synth.make_batch(B, seed=5)
        xs = [torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text")]
        y = torch.from_numpy(b["targets"]).to(dev)
        m1._train_step = m2._train_step = 7
        l1 = m1.train_step_fused(*xs, y)
        l2 = m2.train_step_fused(*xs, y)
        torch.cuda.synchronize()
        T1, T2 = flat_tape(l1["_keep"][0]), flat_tape(l2["_keep"][0])
        bad = 0
        for k in T1:
            if k not in T2:
                print("  missing in chain tape:", k)
                continue
            a, c = T1[k].float(), T2[k].float()
            if a.shape != c.shape:
                print("  shape", k, a.shape, c.shape)
                continue
            d = float((a - c).abs().max()) if a.numel() else 0.0
            if d != 0.0:
                bad += 1
                print(f"  tape {k}: max|d| {d:.3e} of {float(a.abs().max()):.3e}; mismatching {int((a != c).sum())} / {a.numel()}")
        print(f"B={B}: loss ops {float(l1['total_loss']):.8f} chain {float(l2['total_loss']):.8f}; chain used: {l2['_keep'][0].get('chain')}; tape tensors differing: {bad} of {len(T1)}")
        diffs = []
        for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            sc = max(float(p1.grad.abs().max()), 1e-12)
            d = float((p1.grad - p2.grad).abs().max()) / sc
            diffs.append((d if d == d else float("inf"), n))
        coss = []
        for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            g1, g2 = p1.grad.double().flatten(), p2.grad.double().flatten()
            if float(g1.norm()) > 0:
                coss.append((float((g1 @ g2) / (g1.norm() * g2.norm())), float(g2.norm() / g1.norm()), n, g1.numel()))
        coss.sort()
        print("  lowest gradient cosines (cos, norm ratio, name, numel):", [(round(c, 6), round(r, 5), n, k) for c, r, n, k in coss[:4]])
        nz = sum(d != 0.0 for d, _ in diffs)
        diffs.sort(reverse=True)
        print(f"  gradients: worst relative-to-max difference {diffs[0][0]:.3e} ({diffs[0][1]}); {nz} parameters differ")
        for d, n in diffs[:40]:
            if d > 2e-2:
                print(f"    {d:.3e}  {n}")


if __name__ == "__main__":
    main()
