"""SURVEY 8a row a5 on the GPU through the C-ABI: AttentionFusion, BilinearFusion, AdaptiveFusionGating and the factory's
non-hierarchical branches (reference src/models/fusion.py:421-592) -- forward, parameter gradients and input gradients against
the vectors captured from the imported reference (tests/golden/fusion_alt.npz) and against the oracle on other sizes."""
import os

import numpy as np
import pytest
import torch

from mmdeer import fusions, synth
from mmdeer.model import create_fusion_module
from tests.test_oracle_golden import FUSION_ALT_TAGS, check_fusion_alt_grads, fusion_alt_oracle, fusion_alt_params

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DIMS = synth.FUSION_ALT_DIMS


def _oracle():
    from oracle import deer_oracle as O   # test infrastructure only
    return O


def _module(tag, compute="fp32"):
    if tag == "attention":
        m = fusions.AttentionFusion(DIMS, 256, compute_dtype=compute)
    elif tag == "bilinear":
        m = fusions.BilinearFusion(DIMS, 256, compute_dtype=compute)
    elif tag == "bilinear2":
        m = fusions.BilinearFusion(DIMS[:2], 256, compute_dtype=compute)
    elif tag == "adaptive":
        m = fusions.AdaptiveFusionGating(DIMS, ["attention", "bilinear"], 256, compute_dtype=compute)
    elif tag == "factory_attention":
        m = create_fusion_module("attention", {"compute_dtype": compute})
    else:
        m = create_fusion_module("concatenation", {"input_dims": DIMS, "compute_dtype": compute})
    P = fusion_alt_params(GOLDEN, tag)
    m.load_state_dict(P)            # strict: the reference's state_dict keys and shapes
    return m.to("cuda:0").eval(), P


def _run(tag, m, xs):
    if tag == "adaptive":
        o = m(*xs)
        return o["fused_features"], {"strategy_weights": o["strategy_weights"]}
    if tag == "factory_concat":
        return m(torch.cat(xs, dim=-1)), {}
    return m(xs), {}


@pytest.mark.parametrize("tag", FUSION_ALT_TAGS)
def test_alternative_fusions_match_reference_golden(tag):
    g = np.load(os.path.join(GOLDEN, "fusion_alt.npz"))
    m, _ = _module(tag)
    xs_np, c = synth.fusion_alt_inputs(tag, g[tag + ".out"].shape)
    xs = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs_np]
    y, extra = _run(tag, m, xs)
    ref = g[tag + ".out"]
    assert y.dtype == torch.float32 and tuple(y.shape) == ref.shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=3e-4, atol=3e-4 * float(np.abs(ref).max()))
    if tag == "adaptive":
        np.testing.assert_allclose(extra["strategy_weights"].cpu().numpy(), g["adaptive.strategy_weights"], rtol=3e-4, atol=3e-5)
    (y * torch.from_numpy(c).cuda()).sum().backward()
    check_fusion_alt_grads(g, tag, {n: p.grad for n, p in m.named_parameters()}, [x.grad for x in xs], rtol=2e-3, atol_frac=2e-3)


@pytest.mark.parametrize("tag", ["attention", "bilinear", "adaptive", "factory_concat"])
@pytest.mark.parametrize("B", [1, 130])
def test_alternative_fusions_match_oracle_on_other_batches(tag, B):
    O = _oracle()
    m, P = _module(tag)
    xs_np = [synth.normal(40 + i + B, B * d).reshape(B, d).astype(np.float32) for i, d in enumerate(DIMS)]
    xs = [torch.from_numpy(x).cuda().requires_grad_(True) for x in xs_np]
    y, _ = _run(tag, m, xs)
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xo = [torch.from_numpy(x).requires_grad_(True) for x in xs_np]
    yo, _ = fusion_alt_oracle(O, tag, Po, xo)
    scale = float(yo.detach().abs().max())
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), rtol=3e-4, atol=3e-4 * scale)
    c = torch.from_numpy(synth.normal(5, yo.numel()).reshape(yo.shape).astype(np.float32))
    (y * c.cuda()).sum().backward()
    (yo * c).sum().backward()
    for n, p in m.named_parameters():
        ref = Po[n].grad
        if ref is None:
            assert p.grad is None, n
            continue
        s = max(float(ref.abs().max()), 1e-12)
        if n.endswith("attention.bias"):
            s = float(Po[n[:-4] + "weight"].grad.abs().max())
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=3e-3 if not n.endswith("attention.bias") else 0, atol=3e-3 * s, err_msg=f"{tag} {n} B={B}")
    for a, b in zip(xs, xo):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=3e-3, atol=3e-3 * float(b.grad.abs().max()))


def test_bf16_tracks_fp32_and_training_dropout():
    xs_np = [synth.normal(60 + i, 64 * d).reshape(64, d).astype(np.float32) for i, d in enumerate(DIMS)]
    outs = {}
    for compute in ("fp32", "bf16"):
        m, _ = _module("adaptive", compute)
        outs[compute] = m(*[torch.from_numpy(x).cuda() for x in xs_np])["fused_features"]
    a, b = outs["fp32"].double().flatten(), outs["bf16"].double().flatten()
    assert float((a @ b) / (a.norm() * b.norm())) > 0.999
    # training mode: the feature encoder's Dropout(0.3) (fusion.py:440) is live -> strategy weights move, eval does not
    m, _ = _module("adaptive")
    xs = [torch.from_numpy(x).cuda() for x in xs_np]
    w_eval = m(*xs)["strategy_weights"]
    m.train()
    w1, w2 = m(*xs)["strategy_weights"], m(*xs)["strategy_weights"]
    assert not torch.equal(w1, w2) and not torch.equal(w1, w_eval)
    np.testing.assert_allclose(w1.sum(1).cpu().numpy(), np.ones(64, np.float32), rtol=1e-5)
    cf = create_fusion_module("concat", {"input_dims": DIMS}).to("cuda:0").train()
    x = torch.cat(xs, dim=-1)
    y1, y2 = cf(x), cf(x)
    assert not torch.equal(y1, y2) and torch.equal(cf.eval()(x), cf(x))
    y1.sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in cf.parameters())


def test_interface_and_error_behaviour():
    m, _ = _module("adaptive")
    assert sorted(m.state_dict()) == sorted(fusion_alt_params(GOLDEN, "adaptive"))
    xs = [torch.from_numpy(synth.normal(3 + i, 4 * d).reshape(4, d).astype(np.float32)).cuda() for i, d in enumerate(DIMS)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(*(x.cpu() for x in xs))
    with pytest.raises(TypeError, match="must be Tensor, not tuple"):     # the reference's own failure for this strategy (fusion.py:479)
        fusions.AdaptiveFusionGating(DIMS, ["concatenation", "attention"], 256).to("cuda:0")(*xs)
    unknown = fusions.AdaptiveFusionGating(DIMS, ["foo", "bar"], 256).to("cuda:0").eval()     # nothing matches: concatenation Linear fallback
    o = unknown(*xs)
    assert tuple(o["fused_features"].shape) == (4, 256) and tuple(o["strategy_weights"].shape) == (4, 2)
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        fusions.AttentionFusion(DIMS, 256).to("cuda:0")([xs[0], xs[0], xs[2]])
    # empty batch: forward and backward run, gradients are zeros
    att = fusions.AttentionFusion(DIMS, 256).to("cuda:0")
    e = att([x[:0] for x in xs])
    assert tuple(e.shape) == (0, 256)
    e.sum().backward()
    assert all(p.grad is not None and not float(p.grad.abs().sum()) for p in att.parameters())
    one = fusions.BilinearFusion([256], 128).to("cuda:0")
    assert tuple(one([xs[1]]).shape) == (4, 128)
    assert isinstance(create_fusion_module("hierarchical", {"audio_dim": 84, "video_dim": 256, "text_dim": 768}), torch.nn.Module)
    assert isinstance(create_fusion_module("attention", {}), fusions.AttentionFusion)
    seq = create_fusion_module("anything else", {})
    assert isinstance(seq, torch.nn.Sequential) and sorted(seq.state_dict()) == ["0.bias", "0.weight", "3.bias", "3.weight"]
