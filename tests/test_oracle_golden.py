"""Pin the CPU oracle against vectors captured from the imported reference
(tests/golden/make_golden.py).  CPU-only; runs in the build container and on the
GPU box alike (no /root/reference access)."""
import os

import numpy as np
import pytest
import torch

from mmdeer import synth
from mmdeer.spec import DIM_NAMES, param_table
from oracle import deer_oracle as O

GRAD_SLICE = 48


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _batch(B, seed, zero=()):
    b = synth.make_batch(B, seed=seed)
    for z in zero:
        b[z] = np.zeros_like(b[z])
    return [torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets")]


@pytest.mark.parametrize("B", [1, 7, 32])
def test_stackc_eval_forward(golden_dir, B):
    g = _load(golden_dir, f"stackc_B{B}.npz")
    P = O.to_params(synth.closed_form_state())
    a, v, t, _ = _batch(B, 42)
    with torch.no_grad():
        fo, ho = O.model_forward(P, a, v, t)
    for k in ("fused_features", "audiovisual_features", "trimodal_features", "trimodal_attention_weights"):
        np.testing.assert_allclose(fo[k].numpy(), g["eval." + k], rtol=1e-5, atol=2e-5, err_msg=k)
    np.testing.assert_array_equal(fo["av_attention_weights"]["audio_to_video"].numpy(),
                                  g["eval.av_attention.audio_to_video"])
    np.testing.assert_array_equal(fo["av_attention_weights"]["video_to_audio"].numpy(),
                                  g["eval.av_attention.video_to_audio"])
    for k, val in ho.items():
        np.testing.assert_allclose(val.numpy(), g["eval." + k], rtol=2e-5, atol=1e-5, err_msg=k)
    assert fo["uncertainty_weights"] is None


@pytest.mark.parametrize("B", [1, 7, 32])
def test_stackc_loss_and_grads(golden_dir, B):
    g = _load(golden_dir, f"stackc_B{B}.npz")
    P = O.to_params(synth.closed_form_state(), requires_grad=True)
    a, v, t, y = _batch(B, 42)
    _, _, ld, grads = O.train_step(P, a, v, t, y)
    for k, val in ld.items():
        if k.startswith("_") or k.endswith("_bin_counts"):
            continue
        assert float(val) == pytest.approx(float(g["loss." + k]), rel=2e-5, abs=2e-6), k
    for name, _, _ in param_table():
        gr = grads[name].detach()
        ref_norm = float(g["gnorm." + name])
        assert float(gr.double().norm()) == pytest.approx(ref_norm, rel=2e-4, abs=1e-7), name
        flat = gr.numpy().reshape(-1)
        tol = 2e-5 * max(ref_norm, 1e-3)
        np.testing.assert_allclose(flat[:GRAD_SLICE], g["ghead." + name], rtol=1e-4, atol=tol, err_msg=name)
        np.testing.assert_allclose(flat[-GRAD_SLICE:], g["gtail." + name], rtol=1e-4, atol=tol, err_msg=name)
    assert bool(g["gate_grads_all_none"])
    # dead AV q/k rows: exact zeros
    w = grads["fusion.audio_visual_fusion.cross_attention.in_proj_weight"]
    assert float(w[:512].abs().max()) == 0.0


@pytest.mark.parametrize("tag,zero", [("audio_only", ("video", "text")), ("text_only", ("audio", "video"))])
def test_stackc_missing_modalities(golden_dir, tag, zero):
    g = _load(golden_dir, "stackc_missing.npz")
    P = O.to_params(synth.closed_form_state(), requires_grad=True)
    a, v, t, y = _batch(8, 43, zero)
    fo, ho, ld, grads = O.train_step(P, a, v, t, y)
    assert float(ld["total_loss"]) == pytest.approx(float(g[f"{tag}/loss.total_loss"]), rel=2e-5)
    with torch.no_grad():
        fo, ho = O.model_forward(P, a, v, t)
    np.testing.assert_allclose(ho["mu_all"].numpy(), g[f"{tag}/eval.mu_all"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ho["uncertainty_all"].numpy(), g[f"{tag}/eval.uncertainty_all"], rtol=1e-5, atol=2e-6)


def _nig(g):
    e = torch.from_numpy(g["evidence"])
    return O.nig_activations(e), torch.from_numpy(g["targets"])


def test_nig_activations_and_edges(golden_dir):
    g = _load(golden_dir, "loss_cases.npz")
    (mu, nu, alpha, beta, *_), _ = _nig(g)
    for nm, val in (("mu", mu), ("nu", nu), ("alpha", alpha), ("beta", beta)):
        np.testing.assert_array_equal(val.numpy(), g["nig." + nm])
    np.testing.assert_array_equal(np.asarray(O.ECE_EDGES_10, dtype=np.float32), g["linspace11"])
    np.testing.assert_array_equal(np.asarray(O.CAL_EDGES_15, dtype=np.float32), g["linspace16"])
    # alpha - 1 underflow row gives an infinite uncertainty, as in the reference
    assert torch.isinf((beta / (alpha - 1))[2, 0])


@pytest.mark.parametrize("tag,sl", [("reg", slice(5, 64)), ("all", slice(0, 64)), ("one", slice(7, 8)), ("two", slice(9, 11))])
def test_loss_variants(golden_dir, tag, sl):
    g = _load(golden_dir, "loss_cases.npz")
    (mu, nu, alpha, beta, *_), y = _nig(g)
    pred = {}
    for i, d in enumerate(DIM_NAMES):
        pred[f"{d}_mu"], pred[f"{d}_nu"] = mu[sl, i:i + 1], nu[sl, i:i + 1]
        pred[f"{d}_alpha"], pred[f"{d}_beta"] = alpha[sl, i:i + 1], beta[sl, i:i + 1]
    ld = O.multitask_loss(pred, y[sl])

    def close(val, ref, what):
        ref = float(ref)
        if np.isnan(ref) or np.isinf(ref):
            assert (np.isnan(float(val)) and np.isnan(ref)) or float(val) == ref, what
        else:
            assert float(val) == pytest.approx(ref, rel=3e-5, abs=3e-6), what

    for k, val in ld.items():
        if k.endswith("_bin_counts"):
            continue
        close(val, g[f"multitask.{tag}.{k}"], k)
    # CombinedDEERLoss == MultiTaskDEERLoss on per-dimension keys (both extras return 0)
    close(ld["total_loss"], g[f"combined.{tag}.combined_total_loss"], "combined")
    lb = O.deer_loss_v2(mu[sl], nu[sl], alpha[sl], beta[sl], y[sl])
    for k in ("total_loss", "nll_loss", "reg_loss", "kl_loss", "ece_loss"):
        close(lb[k], g[f"basic.{tag}.{k}"], "basic." + k)
    assert lb["batch_size"] == int(g[f"basic.{tag}.batch_size"])
    l1 = O.deer_loss_v2(mu[sl, 0], nu[sl, 0], alpha[sl, 0], beta[sl, 0], y[sl, 0:1])
    close(l1["total_loss"], g[f"basic1d.{tag}.total_loss"], "basic1d")
    if sl.stop - sl.start > 1:
        ur = O.uncertainty_reg_loss(alpha[sl], beta[sl])
        for k, val in ur.items():
            close(val, g[f"unc_reg.{tag}.{k}"], k)
    close(O.calibration_loss(mu[sl], alpha[sl], beta[sl], y[sl]), g[f"calibration.{tag}"], "calibration")
    for kw in (1.0, 0.1):
        l0 = O.deer_loss_v1(mu[sl, 0:1], nu[sl, 0:1], alpha[sl, 0:1], beta[sl, 0:1], y[sl, 0], kl_weight=kw)
        for k, val in l0.items():
            close(val, g[f"v1.kw{kw}.{tag}.{k}"], f"v1.{kw}.{k}")


def test_loss_gradient_wrt_evidence(golden_dir):
    g = _load(golden_dir, "loss_cases.npz")
    e = torch.from_numpy(g["evidence"])[5:64].clone().requires_grad_(True)
    y = torch.from_numpy(g["targets"])[5:64]
    mu, nu, alpha, beta, *_ = O.nig_activations(e)
    pred = {}
    for i, d in enumerate(DIM_NAMES):
        pred[f"{d}_mu"], pred[f"{d}_nu"] = mu[:, i:i + 1], nu[:, i:i + 1]
        pred[f"{d}_alpha"], pred[f"{d}_beta"] = alpha[:, i:i + 1], beta[:, i:i + 1]
    O.multitask_loss(pred, y)["total_loss"].backward()
    np.testing.assert_allclose(e.grad.numpy(), g["multitask.reg.devidence"], rtol=2e-4, atol=1e-7)


def test_side_kernels(golden_dir):
    g = _load(golden_dir, "side_kernels.npz")

    def fill(shapes, tag):
        P = {}
        for name, shape in shapes.items():
            n = int(np.prod(shape))
            u = synth.uniform01(synth._stream_of(tag + "." + name), n) * 2.0 - 1.0
            if len(shape) >= 2:
                w = u * np.sqrt(6.0 / (shape[0] + shape[1]))
            elif name.endswith("weight"):
                w = 1.0 + 0.1 * u
            else:
                w = 0.05 * u
            P[name] = torch.from_numpy(w.reshape(shape).astype(np.float32))
        return P

    lin = lambda n, o, i: {n + ".weight": (o, i), n + ".bias": (o,)}
    shapes = {}
    for n in ("query_proj", "key_proj", "value_proj", "output_proj"):
        shapes.update(lin(n, 256, 256))
    shapes.update(lin("uncertainty_gate.0", 256, 768))
    shapes.update(lin("uncertainty_gate.2", 2, 256))
    P = fill(shapes, "cma")
    wa, wv = O.cross_modal_attention(P, *(torch.from_numpy(g["cma.in." + k]) for k in ("audio", "video", "text")))
    np.testing.assert_allclose(wa.numpy(), g["cma.audio"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(wv.numpy(), g["cma.video"], rtol=1e-5, atol=1e-6)
    assert wa.shape == (9, 32)

    # HierarchicalDEERFusion state_dict order: encoders first (fill is name-keyed anyway)
    shapes = {}
    shapes.update(lin("audio_encoder", 256, 84)); shapes.update(lin("video_encoder", 256, 256))
    shapes.update(lin("text_encoder", 256, 768))
    P = fill(shapes, "hdf")
    b = synth.make_batch(9, seed=77)
    ea, ev, et = O.modality_encoders(P, *(torch.from_numpy(b[k]) for k in ("audio", "video", "text")))
    np.testing.assert_allclose(ea.numpy(), g["hdf.audio_encoded"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ev.numpy(), g["hdf.video_encoded"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(et.numpy(), g["hdf.text_encoded"], rtol=1e-5, atol=1e-6)

    shapes = {}
    for l, inp in ((0, 84), (1, 512)):
        for sfx in ("", "_reverse"):
            shapes[f"lstm.weight_ih_l{l}{sfx}"] = (1024, inp)
            shapes[f"lstm.weight_hh_l{l}{sfx}"] = (1024, 256)
            shapes[f"lstm.bias_ih_l{l}{sfx}"] = (1024,)
            shapes[f"lstm.bias_hh_l{l}{sfx}"] = (1024,)
    shapes.update(lin("attention.0", 256, 512)); shapes.update(lin("attention.2", 1, 256))
    shapes.update(lin("output_projection.0", 512, 512)); shapes.update(lin("output_projection.3", 512, 512))
    shapes.update({"output_projection.4.weight": (512,), "output_projection.4.bias": (512,)})
    P = fill(shapes, "aenc")
    out = O.audio_encoder_features(P, torch.from_numpy(b["audio"]))
    np.testing.assert_allclose(out.numpy(), g["aenc.out"], rtol=1e-4, atol=5e-6)


def test_ccc_formula():
    x = torch.tensor([0.1, 0.4, -0.3, 0.9, 0.0])
    y = torch.tensor([0.2, 0.35, -0.1, 0.7, 0.05])
    # independent evaluation of metrics.py:85-101 with numpy
    xn, yn = x.double().numpy(), y.double().numpy()
    ref = 2 * np.corrcoef(xn, yn)[0, 1] * np.sqrt(xn.var()) * np.sqrt(yn.var()) / (
        xn.var() + yn.var() + (xn.mean() - yn.mean()) ** 2)
    assert O.ccc(x, y) == pytest.approx(ref, rel=1e-12)
    assert O.ccc(x, x) == pytest.approx(1.0)


def _stackb_params(golden_dir, dtype=torch.float32):
    import json
    with open(os.path.join(golden_dir, "stackb_state_dict_names.json")) as fh:
        shapes = json.load(fh)
    return {k: torch.from_numpy(v).to(dtype) for k, v in synth.module_fill("stackb", shapes).items()}


def test_stackb_eval_forward(golden_dir):
    """complete_project.CompleteDEERModel: every output of the reference's eval forward."""
    g = _load(golden_dir, "stackb_B9.npz")
    P = _stackb_params(golden_dir)
    b = synth.make_batch(9, seed=78)
    with torch.no_grad():
        o = O.stackb_forward(P, *(torch.from_numpy(b[k]) for k in ("audio", "video", "text")))
    keys = [k[4:] for k in g if k.startswith("out.")]
    assert len(keys) == 27
    for k in keys:
        np.testing.assert_allclose(o[k].numpy(), g["out." + k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_stackb_single_key_attention_is_two_linears(golden_dir):
    """With one key the softmax is exactly 1, so MultiHeadAttention == output_proj(value_proj(value)) bit for bit --
    the identity the HIP path of mmdeer.stackb relies on."""
    P = _stackb_params(golden_dir)
    x = torch.from_numpy(synth.normal(901, 5 * 256).reshape(5, 1, 256).astype(np.float32))
    q = torch.from_numpy(synth.normal(902, 5 * 256).reshape(5, 1, 256).astype(np.float32))
    full = O._stackb_mha(P, "attention_module.cross_attention", q, x, x)
    short = O._lin(O._lin(x, P, "attention_module.cross_attention.value_proj"), P, "attention_module.cross_attention.output_proj")
    assert torch.equal(full, short)


def stackb_oracle_gradients(O, P, batch):
    """MultiTaskDEERLoss gradients of the oracle's Stack B (dropout sites open), by autograd.  Shared with the GPU tests."""
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    o = O.stackb_forward(Pg, *(torch.from_numpy(batch[k]) for k in ("audio", "video", "text")))
    pred = {k: v.unsqueeze(1) for k, v in o.items() if k.split("_")[-1] in ("mu", "nu", "alpha", "beta") and v.dim() == 1}
    loss = O.multitask_loss(pred, torch.from_numpy(batch["targets"]))
    loss["total_loss"].backward()
    return loss["total_loss"].detach(), {k: v.grad for k, v in Pg.items()}


def check_gradient_digests(g, grads, rtol, atol_frac):
    """`grads` against the digests of tests/golden/stackb_B9.npz (sum, l2 norm, 256 evenly spaced elements per parameter)."""
    names = [k[len("gradnorm."):] for k in g if k.startswith("gradnorm.")]
    assert len(names) == 112          # every parameter but the calibration layer's (query / key projections: exact zeros)
    for n in names:
        v = grads[n].detach().double().cpu().reshape(-1)
        norm = float(g["gradnorm." + n])
        assert float(v.norm()) == pytest.approx(norm, rel=rtol, abs=1e-12), n
        idx = torch.linspace(0, v.numel() - 1, min(256, v.numel())).round().long()
        scale = max(float(np.abs(g["gradsample." + n]).max()), norm / max(v.numel(), 1) ** 0.5)
        np.testing.assert_allclose(v[idx].numpy(), g["gradsample." + n], rtol=rtol, atol=atol_frac * scale, err_msg=n)
        assert float(v.sum()) == pytest.approx(float(g["gradsum." + n]), rel=rtol, abs=atol_frac * norm * max(v.numel(), 1) ** 0.5), n
    for k in g:
        if k.startswith("gradnone."):
            assert grads.get(k[len("gradnone."):]) is None, k


def test_stackb_gradients(golden_dir):
    """Backward of complete_project.CompleteDEERModel under MultiTaskDEERLoss, captured from the reference in eval mode
    (SURVEY 8f-1 training): the oracle's autograd reproduces loss and every parameter gradient digest."""
    g = _load(golden_dir, "stackb_B9.npz")
    loss, grads = stackb_oracle_gradients(O, _stackb_params(golden_dir), synth.make_batch(9, seed=78))
    assert float(loss) == pytest.approx(float(g["grad.total_loss"]), rel=1e-5)
    check_gradient_digests(g, grads, rtol=2e-4, atol_frac=2e-5)


# ---------------------------------------------------------------------------------------------- a5: alternative fusions
FUSION_ALT_TAGS = ("attention", "bilinear", "bilinear2", "adaptive", "factory_attention", "factory_concat")


def fusion_alt_params(golden_dir, tag, dtype=torch.float32):
    import json
    with open(os.path.join(golden_dir, "fusion_alt_state_dict_names.json")) as fh:
        shapes = json.load(fh)[tag]
    return {k: torch.from_numpy(v).to(dtype) for k, v in synth.module_fill("fa_" + tag, shapes).items()}


def fusion_alt_oracle(O, tag, P, xs):
    """(output, extras) of the oracle's restatement of golden case `tag`."""
    if tag in ("attention", "factory_attention"):
        y, w = O.attention_fusion(P, xs)
        return y, {"attention_weights": w}
    if tag in ("bilinear", "bilinear2"):
        return O.bilinear_fusion(P, xs), {}
    if tag == "adaptive":
        y, w = O.adaptive_fusion(P, xs, ["attention", "bilinear"])
        return y, {"strategy_weights": w}
    return O.concat_fusion(P, torch.cat(xs, dim=-1)), {}


def check_fusion_alt_grads(g, tag, grads, dxs, rtol, atol_frac):
    """Parameter and input gradients of case `tag` against tests/golden/fusion_alt.npz (whole tensors, digests for the large ones)."""
    seen = 0
    for k in g:
        if not k.startswith(tag + "."):
            continue
        kind, _, name = k[len(tag) + 1:].partition(".")
        if kind == "grad":
            ref = g[k]
            scale = max(float(np.abs(ref).max()), 1e-12)
            if name.endswith("attention.bias"):      # softmax is shift invariant: this gradient is zero up to rounding -- scale by its weight's
                scale = float(np.abs(g[k[:-len("bias")] + "weight"]).max())
                rtol_k = 0.0
            else:
                rtol_k = rtol
            np.testing.assert_allclose(grads[name].detach().cpu().numpy(), ref, rtol=rtol_k, atol=max(atol_frac, 1e-5) * scale, err_msg=k)
        elif kind == "gradnorm":
            v = grads[name].detach().double().cpu().reshape(-1)
            assert float(v.norm()) == pytest.approx(float(g[k]), rel=rtol), k
            ref = g[f"{tag}.gradsample.{name}"]
            idx = torch.linspace(0, v.numel() - 1, 1024).round().long()
            np.testing.assert_allclose(v[idx].numpy(), ref, rtol=rtol, atol=atol_frac * float(np.abs(ref).max()), err_msg=k)
        elif kind == "gradnone":
            assert grads[name] is None, k
        else:
            continue
        seen += 1
    assert seen >= 2, tag
    for i, dx in enumerate(dxs):
        ref = g[f"{tag}.dx{i}"]
        np.testing.assert_allclose(dx.detach().cpu().numpy(), ref, rtol=rtol, atol=atol_frac * float(np.abs(ref).max()), err_msg=f"{tag}.dx{i}")


@pytest.mark.parametrize("tag", FUSION_ALT_TAGS)
def test_alternative_fusions(golden_dir, tag):
    """AttentionFusion / BilinearFusion / AdaptiveFusionGating and the factory's non-hierarchical branches (fusion.py:421-592):
    eval outputs, parameter gradients and input gradients of the imported reference."""
    g = _load(golden_dir, "fusion_alt.npz")
    P = {k: v.requires_grad_(True) for k, v in fusion_alt_params(golden_dir, tag).items()}
    xs_np, c = synth.fusion_alt_inputs(tag, g[tag + ".out"].shape)
    xs = [torch.from_numpy(x).requires_grad_(True) for x in xs_np]
    y, extra = fusion_alt_oracle(O, tag, P, xs)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + ".out"], rtol=2e-5, atol=2e-5 * float(np.abs(g[tag + ".out"]).max()))
    if tag == "adaptive":
        np.testing.assert_allclose(extra["strategy_weights"].detach().numpy(), g["adaptive.strategy_weights"], rtol=2e-5, atol=1e-6)
    (y * torch.from_numpy(c)).sum().backward()
    check_fusion_alt_grads(g, tag, {k: v.grad for k, v in P.items()}, [x.grad for x in xs], rtol=2e-4, atol_frac=2e-5)


# ---------------------------------------------------------------------------------------------- a3 on another geometry
FUSION_GEOM = (("audio_dim", 40), ("video_dim", 128), ("text_dim", 300))


def fusion_geom_case(golden_dir, dtype=torch.float32):
    """(parameters under the oracle's 'fusion.' prefix, inputs, the three cotangents) of tests/golden/fusion_geom.npz: everything is
    closed-form (synth.module_fill / synth.normal), only the expected values come from the file."""
    import json
    with open(os.path.join(golden_dir, "fusion_geom_state_dict_names.json")) as fh:
        shapes = json.load(fh)
    P = {k: torch.from_numpy(v).to(dtype) for k, v in synth.module_fill("fgeom", shapes).items()}
    B = 9
    xs = [torch.from_numpy(synth.normal(810 + i, B * d).reshape(B, d).astype(np.float32)).to(dtype) for i, (_k, d) in enumerate(FUSION_GEOM)]
    cs = [torch.from_numpy(synth.normal(830 + j, B * w).reshape(B, w).astype(np.float32)).to(dtype) for j, w in enumerate((512, 256, 512))]
    return P, xs, cs


def check_fusion_geom_grads(g, grads, dxs, rtol, atol_frac):
    seen = 0
    for k in (g.files if hasattr(g, "files") else list(g)):
        if not k.startswith("train."):
            continue
        kind, _, name = k[len("train."):].partition(".")
        if kind == "grad":
            ref = g[k]
            scale = max(float(np.abs(ref).max()), 1e-12)
            np.testing.assert_allclose(grads[name].detach().cpu().numpy(), ref, rtol=rtol, atol=atol_frac * scale + 1e-7, err_msg=k)
        elif kind == "gradnorm":
            v = grads[name].detach().double().cpu().reshape(-1)
            assert float(v.norm()) == pytest.approx(float(g[k]), rel=rtol), k
            ref = g[f"train.gradsample.{name}"]
            idx = torch.linspace(0, v.numel() - 1, 1024).round().long()
            np.testing.assert_allclose(v[idx].numpy(), ref, rtol=rtol, atol=atol_frac * float(np.abs(ref).max()) + 1e-7, err_msg=k)
        elif kind == "gradnone":
            assert grads.get(name) is None or float(grads[name].abs().sum()) == 0.0, k
        else:
            continue
        seen += 1
    assert seen >= 20
    for (key, _d), dx in zip(FUSION_GEOM, dxs):
        ref = g[f"train.dx.{key}"]
        np.testing.assert_allclose(dx.detach().cpu().numpy(), ref, rtol=rtol, atol=atol_frac * float(np.abs(ref).max()), err_msg=key)


def test_hierarchical_fusion_on_another_geometry(golden_dir):
    """fusion.HierarchicalMultimodalFusion(audio_dim=40, video_dim=128, text_dim=300) (fusion.py:47-50 takes any widths): the oracle's
    fusion_forward is shape-generic -- eval outputs and the gradients of the imported reference at that geometry."""
    g = _load(golden_dir, "fusion_geom.npz")
    P, xs, cs = fusion_geom_case(golden_dir)
    Pf = {"fusion." + k: v.clone().requires_grad_(True) for k, v in P.items()}
    with torch.no_grad():
        o = O.fusion_forward(Pf, *xs, masks=None)
    for k in ("fused_features", "audiovisual_features", "trimodal_features", "trimodal_attention_weights"):
        np.testing.assert_allclose(o[k].numpy(), g["eval." + k], rtol=2e-5, atol=2e-5 * float(np.abs(g["eval." + k]).max()), err_msg=k)
    np.testing.assert_allclose(o["av_attention_weights"]["audio_to_video"].numpy(), g["eval.av_attention.audio_to_video"])
    xg = [x.clone().requires_grad_(True) for x in xs]
    o = O.fusion_forward(Pf, *xg, masks=None)
    loss = sum((o[k] * c).sum() for k, c in zip(("fused_features", "audiovisual_features", "trimodal_features"), cs))
    assert float(loss) == pytest.approx(float(g["train.loss"]), rel=1e-5)
    loss.backward()
    check_fusion_geom_grads(g, {k[len("fusion."):]: v.grad for k, v in Pf.items()}, [x.grad for x in xg], rtol=3e-4, atol_frac=3e-5)


# ---------------------------------------------------------------------------------------------- side rows: backward (a8, a14)
def side_shapes(tag):
    """state_dict shapes of deer.CrossModalAttention(256, 8) ('cma') / encoders.EnhancedAudioEncoder() ('aenc')."""
    lin = lambda n, o, i: {n + ".weight": (o, i), n + ".bias": (o,)}     # noqa: E731
    shapes = {}
    if tag == "cma":
        for n in ("query_proj", "key_proj", "value_proj", "output_proj"):
            shapes.update(lin(n, 256, 256))
        shapes.update(lin("uncertainty_gate.0", 256, 768)); shapes.update(lin("uncertainty_gate.2", 2, 256))
        return shapes
    for l, inp in ((0, 84), (1, 512)):
        for sfx in ("", "_reverse"):
            shapes[f"lstm.weight_ih_l{l}{sfx}"] = (1024, inp)
            shapes[f"lstm.weight_hh_l{l}{sfx}"] = (1024, 256)
            shapes[f"lstm.bias_ih_l{l}{sfx}"] = (1024,)
            shapes[f"lstm.bias_hh_l{l}{sfx}"] = (1024,)
    shapes.update(lin("attention.0", 256, 512)); shapes.update(lin("attention.2", 1, 256))
    shapes.update(lin("output_projection.0", 512, 512)); shapes.update(lin("output_projection.3", 512, 512))
    shapes.update({"output_projection.4.weight": (512,), "output_projection.4.bias": (512,)})
    return shapes


def synth_state(golden_dir, tag):
    return {k: torch.from_numpy(v) for k, v in synth.module_fill(tag, side_shapes(tag)).items()}


def check_side_grads(g, tag, grads, dxs, rtol, atol_frac):
    """Gradients of case `tag` ('cma' / 'aenc') against tests/golden/side_kernels.npz.  A golden zero tensor (recurrent weights at
    T = 1, the attention pool over one step) accepts None or zeros; a golden None (CrossModalAttention.output_proj) needs None."""
    seen = 0
    for k in g:
        if not k.startswith(tag + "."):
            continue
        kind, _, name = k[len(tag) + 1:].partition(".")
        if kind == "grad":
            ref = g[k]
            if not np.abs(ref).max():
                assert grads.get(name) is None or not float(grads[name].abs().max()), k
            else:
                np.testing.assert_allclose(grads[name].detach().cpu().numpy(), ref, rtol=rtol, atol=atol_frac * float(np.abs(ref).max()), err_msg=k)
        elif kind == "gradnorm":
            if float(g[k]) == 0.0:
                assert grads.get(name) is None or not float(grads[name].abs().max()), k
            else:
                v = grads[name].detach().double().cpu().reshape(-1)
                assert float(v.norm()) == pytest.approx(float(g[k]), rel=rtol), k
                ref = g[f"{tag}.gradsample.{name}"]
                idx = torch.linspace(0, v.numel() - 1, 1024).round().long()
                np.testing.assert_allclose(v[idx].numpy(), ref, rtol=rtol, atol=atol_frac * float(np.abs(ref).max()), err_msg=k)
        elif kind == "gradnone":
            assert grads.get(name) is None, k
        elif kind == "dx":
            np.testing.assert_allclose(dxs[name].detach().cpu().numpy(), g[k], rtol=rtol, atol=atol_frac * float(np.abs(g[k]).max()), err_msg=k)
        else:
            continue
        seen += 1
    assert seen >= 10, (tag, seen)


def side_loss_weights(tag, B=9):
    """The closed-form loss weights of the golden backward passes (make_golden.py: capture_side)."""
    if tag == "cma":
        return [torch.from_numpy(synth.normal(520 + i, B * 32).reshape(B, 32).astype(np.float32)) for i in range(2)]
    return [torch.from_numpy(synth.normal(530, B * 512).reshape(B, 512).astype(np.float32))]


def test_side_rows_backward(golden_dir):
    """CrossModalAttention (deer.py:379-425) and the EnhancedAudioEncoder feature branch (encoders.py:356-389): parameter and
    input gradients of the imported reference against the oracle's autograd."""
    g = _load(golden_dir, "side_kernels.npz")
    P = {k: v.requires_grad_(True) for k, v in synth_state(golden_dir, "cma").items()}
    xs = {k: torch.from_numpy(g["cma.in." + k]).requires_grad_(True) for k in ("audio", "video", "text")}
    wa, wv = O.cross_modal_attention(P, xs["audio"], xs["video"], xs["text"])
    ca, cv = side_loss_weights("cma")
    ((wa * ca).sum() + (wv * cv).sum()).backward()
    check_side_grads(g, "cma", {k: v.grad for k, v in P.items()}, {k: v.grad for k, v in xs.items()}, rtol=2e-4, atol_frac=2e-5)
    P = {k: v.requires_grad_(True) for k, v in synth_state(golden_dir, "aenc").items()}
    x = torch.from_numpy(synth.make_batch(9, seed=77)["audio"]).requires_grad_(True)
    y = O.audio_encoder_features(P, x)
    (y * side_loss_weights("aenc")[0]).sum().backward()
    check_side_grads(g, "aenc", {k: v.grad for k, v in P.items()}, {"audio": x.grad}, rtol=3e-4, atol_frac=3e-5)
