"""fusion.HierarchicalMultimodalFusion on a geometry other than the default (audio 40 / video 128 / text 300)
through the operator path (mmdeer/generic_fusion.py): state_dict keys and shapes of the reference at that geometry, eval outputs
and train-mode gradients against the vectors captured from the imported reference (tests/golden/fusion_geom.npz), against the
oracle on other batch sizes, bf16 tracking fp32, and training with dropout."""
import json
import os

import numpy as np
import pytest
import torch

from mmdeer import synth
from mmdeer.model import HierarchicalMultimodalFusion, create_fusion_module
from tests.test_oracle_golden import FUSION_GEOM, check_fusion_geom_grads, fusion_geom_case

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _module(compute="fp32", dropout=0.3, **kw):
    m = HierarchicalMultimodalFusion(**dict(FUSION_GEOM), dropout=dropout, compute_dtype=compute, **kw)
    P, xs, cs = fusion_geom_case(GOLDEN)
    m.load_state_dict(P)                 # strict: exactly the reference's keys and shapes at this geometry
    return m.to(DEV), P, xs, cs


def test_state_dict_is_the_references_at_that_geometry():
    m, _, _, _ = _module()
    shapes = json.load(open(os.path.join(GOLDEN, "fusion_geom_state_dict_names.json")))
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == shapes


def test_eval_outputs_and_gradients_match_the_reference():
    g = np.load(os.path.join(GOLDEN, "fusion_geom.npz"))
    m, _, xs, cs = _module()
    m.eval()
    with torch.no_grad():
        o = m(*[x.to(DEV) for x in xs])
    for k in ("fused_features", "audiovisual_features", "trimodal_features", "trimodal_attention_weights"):
        ref = g["eval." + k]
        assert tuple(o[k].shape) == ref.shape and o[k].dtype == torch.float32
        np.testing.assert_allclose(o[k].cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), err_msg=k)
    for k in ("audio_to_video", "video_to_audio"):
        np.testing.assert_allclose(o["av_attention_weights"][k].cpu().numpy(), g["eval.av_attention." + k])
    assert o["uncertainty_weights"] is None
    # train mode, dropout 0: gradients of sum(fused c0 + audiovisual c1 + trimodal c2)
    m, _, xs, cs = _module(dropout=0.0)
    m.train()
    xg = [x.to(DEV).requires_grad_(True) for x in xs]
    o = m(*xg)
    loss = sum((o[k] * c.to(DEV)).sum() for k, c in zip(("fused_features", "audiovisual_features", "trimodal_features"), cs))
    assert float(loss) == pytest.approx(float(g["train.loss"]), rel=2e-4)
    loss.backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    E = 256
    qk = grads["audio_visual_fusion.cross_attention.in_proj_weight"][:2 * E]
    assert float(qk.abs().sum()) == 0.0                      # one key per query: the q / k rows are dead, exact zeros as in the reference
    check_fusion_geom_grads(g, grads, [x.grad for x in xg], rtol=3e-3, atol_frac=3e-3)


@pytest.mark.parametrize("B", [1, 130])
def test_other_batch_sizes_against_the_oracle(B):
    from oracle import deer_oracle as O   # test infrastructure only
    m, P, _, _ = _module(dropout=0.0)
    m.train()
    xs_np = [synth.normal(90 + i + B, B * d).reshape(B, d).astype(np.float32) for i, (_k, d) in enumerate(FUSION_GEOM)]
    xg = [torch.from_numpy(x).to(DEV).requires_grad_(True) for x in xs_np]
    o = m(*xg)
    Pf = {"fusion." + k: v.clone().requires_grad_(True) for k, v in P.items()}
    xo = [torch.from_numpy(x).requires_grad_(True) for x in xs_np]
    oo = O.fusion_forward(Pf, *xo, masks=None)
    for k in ("fused_features", "audiovisual_features", "trimodal_features", "trimodal_attention_weights"):
        np.testing.assert_allclose(o[k].detach().cpu().numpy(), oo[k].detach().numpy(), rtol=1e-4, atol=1e-4 * float(oo[k].detach().abs().max()))
    c = torch.from_numpy(synth.normal(7, B * 512).reshape(B, 512).astype(np.float32))
    (o["fused_features"] * c.to(DEV)).sum().backward()
    (oo["fused_features"] * c).sum().backward()
    for n, p in m.named_parameters():
        ref = Pf["fusion." + n].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n
            continue
        s = max(float(ref.abs().max()), 1e-12)
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=3e-3, atol=3e-3 * s, err_msg=n)
    for a, b in zip(xg, xo):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=3e-3, atol=3e-3 * float(b.grad.abs().max()))


def test_bf16_tracks_fp32_and_dropout_training_runs():
    outs = {}
    for compute in ("fp32", "bf16"):
        m, _, xs, _ = _module(compute)
        m.eval()
        with torch.no_grad():
            outs[compute] = m(*[x.to(DEV) for x in xs])["fused_features"]
    d = (outs["bf16"] - outs["fp32"]).abs().max() / outs["fp32"].abs().max()
    assert float(d) < 5e-2, float(d)
    m, _, xs, _ = _module("bf16", dropout=0.3)
    m.train()
    xg = [x.to(DEV).requires_grad_(True) for x in xs]
    o1 = m(*xg)
    o1["fused_features"].square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    w = o1["av_attention_weights"]["audio_to_video"]
    assert tuple(w.shape) == (9, 1) and float(w.min()) >= 0.0 and float(w.max()) <= 1.0 / 0.7 + 1e-6
    o2 = m(*[x.to(DEV) for x in xs])
    assert not torch.equal(o1["fused_features"], o2["fused_features"])      # fresh masks per training forward
    with pytest.raises(TypeError):
        m(*[x.to(DEV) for x in xs], uncertainties={"audio": torch.ones(9, 1)})
    # use_uncertainty_weighting=False: the reference never enters the broken branch (fusion.py:147) and ignores `uncertainties`
    m2 = HierarchicalMultimodalFusion(**dict(FUSION_GEOM), use_uncertainty_weighting=False).to(DEV).eval()
    assert "uncertainty_gate.gating_network.0.weight" not in m2.state_dict()
    with torch.no_grad():
        o = m2(*[x.to(DEV) for x in xs], uncertainties={"audio": torch.ones(9, 1)})
    assert torch.isfinite(o["fused_features"]).all()


def test_factory_builds_other_geometries():
    m = create_fusion_module("hierarchical", {"audio_dim": 40, "video_dim": 128, "text_dim": 300})
    assert m.audio_visual_fusion.audio_projection.weight.shape == (256, 40)
    with pytest.raises(NotImplementedError):
        HierarchicalMultimodalFusion(40, 128, 300, fusion_dim=256)
