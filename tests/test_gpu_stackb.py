"""Stack B (complete_project.CompleteDEERModel, eval forward) on the GPU through the C-ABI, against the
golden vectors captured from the reference and against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from mmdeer import stackb, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TENSOR_KEYS = ("mu_all", "uncertainty_all", "calibrated_uncertainty", "attention_weights", "modality_uncertainties", "fused_features")


def _shapes():
    with open(os.path.join(GOLDEN, "stackb_state_dict_names.json")) as fh:
        return json.load(fh)


def _model(compute="fp32", tag="stackb"):
    m = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype=compute)
    P = {k: torch.from_numpy(v) for k, v in synth.module_fill(tag, _shapes()).items()}
    m.load_state_dict(P)
    return m.to("cuda:0").eval(), P


def _inputs(B, seed):
    b = synth.make_batch(B, seed=seed)
    return [torch.from_numpy(b[k]) for k in ("audio", "video", "text")]


def _oracle():
    from oracle import deer_oracle as O   # test infrastructure only
    return O


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_stackb_matches_reference_golden(compute):
    g = np.load(os.path.join(GOLDEN, "stackb_B9.npz"))
    m, _ = _model(compute)
    out = m(*(x.to("cuda:0") for x in _inputs(9, 78)))
    keys = [k[4:] for k in g.files]
    assert sorted(out) == sorted(keys)
    # fp32: the exact-fp32 MFMA path; bf16: operands rounded to bf16 at every layer (12 layers deep)
    tol = dict(rtol=2e-4, atol=2e-5) if compute == "fp32" else dict(rtol=0.08, atol=0.06)
    for k in keys:
        assert out[k].dtype == torch.float32 and tuple(out[k].shape) == g["out." + k].shape, k
        np.testing.assert_allclose(out[k].cpu().numpy(), g["out." + k], err_msg=k, **tol)


@pytest.mark.parametrize("B", [1, 1027])
def test_stackb_matches_oracle_other_parameters_and_ragged_batch(B):
    O = _oracle()
    m, P = _model("fp32", tag="stackb2")
    xs = _inputs(B, 5)
    out = m(*(x.to("cuda:0") for x in xs))
    with torch.no_grad():
        ref = O.stackb_forward(P, *xs)
    for k in TENSOR_KEYS + ("valence_nu", "arousal_alpha", "dominance_beta", "valence_aleatoric_uncertainty", "arousal_epistemic_uncertainty"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=3e-4, atol=3e-5, err_msg=k)
    p, u = m.get_predictions_and_uncertainties(out)
    assert p is out["mu_all"] and u is out["calibrated_uncertainty"]
    # softmax rows sum to one; sigmoid outputs stay in (0, 1)
    np.testing.assert_allclose(out["attention_weights"].sum(1).cpu().numpy(), np.ones(B, np.float32), rtol=1e-5)
    for k in ("modality_uncertainties", "calibrated_uncertainty"):
        v = out[k].cpu().numpy()
        assert (v > 0).all() and (v < 1).all()


def test_stackb_missing_modality_and_empty_batch():
    O = _oracle()
    m, P = _model("fp32")
    a, v, t = _inputs(6, 11)
    v = torch.zeros_like(v)                     # a missing modality arrives as zeros (SURVEY 8c edge cases)
    out = m(a.cuda(), v.cuda(), t.cuda())
    with torch.no_grad():
        ref = O.stackb_forward(P, a, v, t)
    for k in TENSOR_KEYS:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=3e-4, atol=3e-5, err_msg=k)
    out = m(a[:0].cuda(), v[:0].cuda(), t[:0].cuda())
    assert out["mu_all"].shape == (0, 3) and out["fused_features"].shape == (0, 512) and out["valence_mu"].shape == (0,)


def test_stackb_other_encoder_depths_and_input_widths():
    """encoder_layers and the input widths are ModelConfig fields (complete_project.py:33-56), not constants."""
    O = _oracle()
    for layers, dims, compute, tol in ((1, (40, 64, 128), "fp32", dict(rtol=3e-4, atol=3e-5)), (0, (84, 256, 768), "fp32", dict(rtol=3e-4, atol=3e-5)),
                                       (4, (200, 256, 768), "bf16", dict(rtol=0.08, atol=0.06))):
        cfg = stackb.ModelConfig(audio_dim=dims[0], video_dim=dims[1], text_dim=dims[2], encoder_layers=layers)
        m = stackb.CompleteDEERModel(cfg, compute_dtype=compute)
        P = {k: torch.from_numpy(v) for k, v in synth.module_fill(f"stackb.l{layers}", {k: tuple(v.shape) for k, v in m.state_dict().items()}).items()}
        m.load_state_dict(P)
        m = m.to("cuda:0").eval()
        xs = [torch.from_numpy(synth.normal(700 + i, 37 * d).reshape(37, d).astype(np.float32)) for i, d in enumerate(dims)]
        out = m(*(x.cuda() for x in xs))
        with torch.no_grad():
            ref = O.stackb_forward(P, *xs, layers=layers)
        for k in TENSOR_KEYS:
            np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), err_msg=f"layers={layers} {k}", **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [256, 512])
def test_residual_layer_norm_operator(dtype, N):
    from mmdeer import ops
    M = 133
    y = torch.from_numpy(synth.normal(810, M * N).reshape(M, N).astype(np.float32)).cuda().to(dtype)
    x = torch.from_numpy(synth.normal(811, M * 2 * N).reshape(M, 2 * N).astype(np.float32)).cuda().to(dtype)[:, N:]   # a column block
    g = torch.from_numpy((1 + 0.1 * synth.normal(812, N)).astype(np.float32)).cuda()
    b = torch.from_numpy((0.1 * synth.normal(813, N)).astype(np.float32)).cuda()
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    for res in (None, x):
        out = ops.residual_layer_norm(y, res, g, b, torch.empty(M, N, dtype=dtype, device="cuda:0"))
        ref = torch.nn.functional.layer_norm(y.float(), (N,), g, b, 1e-5) + (res.float() if res is not None else 0)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), **tol)
    with pytest.raises(RuntimeError, match="256 or 512"):
        ops.residual_layer_norm(y[:, :128], None, g, b, torch.empty(M, 128, dtype=dtype, device="cuda:0"))


def test_stackb_interface():
    m, _ = _model("fp32")
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == _shapes()
    xs = [x.to("cuda:0") for x in _inputs(4, 3)]
    with pytest.raises(NotImplementedError, match="inference-only"):
        m.train()(*xs)
    m.eval()
    with pytest.raises(ValueError, match="expected features"):
        m(xs[0], xs[1][:, :128], xs[2])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(*(x.cpu() for x in xs))
    # parameters edited in place are picked up (the packed operand images are keyed on parameter versions)
    before = m(*xs)["mu_all"].clone()
    with torch.no_grad():
        m.prediction_heads["valence"].evidence_network[6].bias.add_(1.0)
    after = m(*xs)["mu_all"]
    np.testing.assert_allclose((after - before).cpu().numpy(), np.tile(np.float32([1, 0, 0]), (4, 1)), atol=1e-5)
    with pytest.raises(NotImplementedError):
        stackb.CompleteDEERModel(stackb.ModelConfig(encoder_dim=128))


def test_stackb_graph_replay_equals_eager():
    m, _ = _model("bf16")
    xs = [x.to("cuda:0") for x in _inputs(64, 21)]
    replay = m.capture(*xs)
    ys = [x.to("cuda:0") for x in _inputs(64, 22)]
    got = {k: v.clone() for k, v in replay(*ys).items()}
    want = m(*ys)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    again = replay(*xs)
    assert torch.equal(again["mu_all"], m(*xs)["mu_all"])


def test_stackb_bf16_agrees_with_fp32_by_ccc():
    """bf16 operands through ~14 layers: per-dimension concordance (metrics.py:85-101 formula) of the bf16 outputs with the
    fp32 outputs at B = 1024, reference-style (Xavier) weights."""
    torch.manual_seed(11)
    m32 = stackb.CompleteDEERModel(compute_dtype="fp32")
    m16 = stackb.CompleteDEERModel(compute_dtype="bf16")
    m16.load_state_dict(m32.state_dict())
    m32, m16 = m32.to("cuda:0").eval(), m16.to("cuda:0").eval()
    xs = [x.to("cuda:0") for x in _inputs(1024, 17)]
    o32, o16 = m32(*xs), m16(*xs)

    def ccc(x, y):
        x, y = x.double(), y.double()
        mx, my = x.mean(0), y.mean(0)
        vx, vy = x.var(0, unbiased=False), y.var(0, unbiased=False)
        cov = ((x - mx) * (y - my)).mean(0)
        return 2 * cov / (vx + vy + (mx - my) ** 2)

    for k, floor in (("mu_all", 0.999), ("uncertainty_all", 0.99), ("calibrated_uncertainty", 0.99), ("fused_features", 0.995)):
        c = ccc(o16[k], o32[k])
        assert float(c.min()) > floor, (k, c.cpu().numpy())
    assert float((o16["attention_weights"] - o32["attention_weights"]).abs().max()) < 0.05
