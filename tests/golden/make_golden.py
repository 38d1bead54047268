#!/usr/bin/env python3
"""Capture golden vectors from the IMPORTED reference (build container only).

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only); writes small .npz fixtures next to this file.
Nothing here runs on the GPU box: the fixtures + the closed-form generators of
``mmdeer.synth`` are all that travels.

What is captured (SURVEY 8c golden-vector plan):
  * stackc_B{1,7,32}.npz  -- HierarchicalMultimodalFusion -> MultiDimensionalDEER ->
    MultiTaskDEERLoss with closed-form parameters: eval-mode outputs, and
    dropout=0.0 train-mode loss components + gradient digests.
  * stackc_missing.npz    -- audio-only / text-only batches (other modalities zeroed).
  * loss_cases.npz        -- losses.DEERLoss / MultiTaskDEERLoss / CombinedDEERLoss /
    UncertaintyRegularizationLoss / CalibrationLoss / deer.DEERLoss on hand-made
    NIG parameters incl. extreme pre-activations and empty / single-element ECE bins.
  * side_kernels.npz      -- deer.CrossModalAttention, HierarchicalDEERFusion's three
    encoders, EnhancedAudioEncoder feature branch (eval).
  * metrics_cases.npz     -- utils/metrics.py: DEERMetrics CCC / MAE / RMSE per dimension and the quantile-binned
    uncertainty_calibration_error on closed-form (predictions, targets, uncertainties), incl. NaN / inf rows, ties and a
    too-small set (`python tests/golden/make_golden.py metrics` regenerates only this one).
  * stackb_B9.npz         -- complete_project.CompleteDEERModel, eval-mode forward with
    closed-form parameters (`python tests/golden/make_golden.py stackb` regenerates only this one).
  * fusion_geom.npz       -- fusion.HierarchicalMultimodalFusion at audio 40 / video 128 / text 300: eval outputs and train-mode
    (dropout 0) gradients (`python tests/golden/make_golden.py fusion_geom`).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("MMDEER_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(REF, "src", "models"), os.path.join(REF, "src", "utils")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import deer as ref_deer  # noqa: E402  (reference)
import fusion as ref_fusion  # noqa: E402  (reference)
import losses as ref_losses  # noqa: E402  (reference)

from mmdeer import synth  # noqa: E402
from mmdeer.spec import DIM_NAMES, param_table  # noqa: E402

torch.set_num_threads(4)
GRAD_SLICE = 48


def build_reference(dropout):
    f = ref_fusion.HierarchicalMultimodalFusion(84, 256, 768, fusion_dim=512, intermediate_dim=256,
                                                num_attention_heads=8, dropout=dropout,
                                                use_uncertainty_weighting=True)
    h = ref_deer.MultiDimensionalDEER(512, emotion_dims=3, hidden_dim=256, dropout=dropout)
    sd = synth.closed_form_state(include_gate=True)
    f.load_state_dict({k[len("fusion."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("fusion.")})
    h.load_state_dict({k[len("head."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("head.")})
    return f, h


def tnp(t):
    return t.detach().cpu().numpy()


def capture_stackc(B, seed, zero=()):
    batch = synth.make_batch(B, seed=seed)
    for z in zero:
        batch[z] = np.zeros_like(batch[z])
    a, v, t, y = (torch.from_numpy(batch[k]) for k in ("audio", "video", "text", "targets"))
    out = {}
    # eval-mode forward
    f, h = build_reference(0.3)
    f.eval(); h.eval()
    with torch.no_grad():
        fo = f(a, v, t)
        ho = h(fo["fused_features"])
    for k in ("fused_features", "audiovisual_features", "trimodal_features", "trimodal_attention_weights"):
        out["eval." + k] = tnp(fo[k])
    out["eval.av_attention.audio_to_video"] = tnp(fo["av_attention_weights"]["audio_to_video"])
    out["eval.av_attention.video_to_audio"] = tnp(fo["av_attention_weights"]["video_to_audio"])
    for k, val in ho.items():
        out["eval." + k] = tnp(val)
    # train-mode, dropout disabled: loss + gradients
    f, h = build_reference(0.0)
    f.train(); h.train()
    loss_fn = ref_losses.MultiTaskDEERLoss()
    fo = f(a, v, t)
    ho = h(fo["fused_features"])
    ld = loss_fn(ho, y)
    ld["total_loss"].backward()
    for k, val in ld.items():
        out["loss." + k] = np.asarray(float(val), dtype=np.float64)
    named = {"fusion." + n: p for n, p in f.named_parameters()}
    named.update({"head." + n: p for n, p in h.named_parameters()})
    for name, _, _ in param_table():
        g = named[name].grad
        g = torch.zeros_like(named[name]) if g is None else g
        out["gnorm." + name] = np.asarray(float(g.double().norm()), dtype=np.float64)
        out["gsum." + name] = np.asarray(float(g.double().sum()), dtype=np.float64)
        flat = tnp(g).reshape(-1)
        out["ghead." + name] = flat[:GRAD_SLICE].copy()
        out["gtail." + name] = flat[-GRAD_SLICE:].copy()
    gate = [p.grad for n, p in f.named_parameters() if n.startswith("uncertainty_gate.")]
    out["gate_grads_all_none"] = np.asarray(all(g is None for g in gate))
    return out


def nig_cases():
    """Hand-made raw evidence (B,3,4) incl. extremes; targets (B,3)."""
    rng_u = synth.uniform01(991, 64 * 12).reshape(64, 3, 4) * 6.0 - 3.0
    e = rng_u.copy()
    e[0, :, 1:] = 25.0       # softplus threshold (x > 20 -> identity)
    e[1, :, 1:] = -25.0      # tiny evidence
    e[2, 0, 2] = -104.0      # alpha-1 underflows to 0 -> inf uncertainty
    e[3, :, 2] = 19.999      # just under the softplus threshold
    e[4, :, 2] = 20.001
    y = np.tanh(synth.normal(992, 64 * 3).reshape(64, 3))
    return e.astype(np.float32), y.astype(np.float32)


def capture_losses():
    import torch.nn.functional as F
    out = {}
    e, y = nig_cases()
    out["evidence"] = e
    out["targets"] = y
    et = torch.from_numpy(e)
    yt = torch.from_numpy(y)
    mu = et[..., 0]
    nu = F.softplus(et[..., 1]) + 1e-6
    alpha = F.softplus(et[..., 2]) + 1.0
    beta = F.softplus(et[..., 3]) + 1e-6
    for nm, val in (("mu", mu), ("nu", nu), ("alpha", alpha), ("beta", beta)):
        out["nig." + nm] = tnp(val)
    # regular rows only (5..63) for the finite-loss cases; extremes get their own entry
    for tag, sl in (("reg", slice(5, 64)), ("all", slice(0, 64)), ("one", slice(7, 8)), ("two", slice(9, 11))):
        pred = {}
        for i, d in enumerate(DIM_NAMES):
            pred[f"{d}_mu"] = mu[sl, i:i + 1]
            pred[f"{d}_nu"] = nu[sl, i:i + 1]
            pred[f"{d}_alpha"] = alpha[sl, i:i + 1]
            pred[f"{d}_beta"] = beta[sl, i:i + 1]
        ld = ref_losses.MultiTaskDEERLoss()(pred, yt[sl])
        for k, val in ld.items():
            out[f"multitask.{tag}.{k}"] = np.asarray(float(val), dtype=np.float64)
        lc = ref_losses.CombinedDEERLoss()(pred, yt[sl])
        out[f"combined.{tag}.combined_total_loss"] = np.asarray(float(lc["combined_total_loss"]), dtype=np.float64)
        # basic losses.DEERLoss on (B,3) tensors and 1-D targets path
        lb = ref_losses.DEERLoss()({"gamma": mu[sl], "nu": nu[sl], "alpha": alpha[sl], "beta": beta[sl]}, yt[sl])
        for k, val in lb.items():
            out[f"basic.{tag}.{k}"] = np.asarray(float(val), dtype=np.float64)
        l1 = ref_losses.DEERLoss()({"mu": mu[sl, 0], "lambda": nu[sl, 0], "alpha": alpha[sl, 0], "beta": beta[sl, 0]},
                                   yt[sl, 0:1])
        out[f"basic1d.{tag}.total_loss"] = np.asarray(float(l1["total_loss"]), dtype=np.float64)
        # flat-key extras
        flat = {"gamma": mu[sl], "alpha": alpha[sl], "beta": beta[sl]}
        if sl.stop - sl.start > 1:
            ur = ref_losses.UncertaintyRegularizationLoss()(flat, yt[sl])
            for k, val in ur.items():
                out[f"unc_reg.{tag}.{k}"] = np.asarray(float(val), dtype=np.float64)
        out[f"calibration.{tag}"] = np.asarray(float(ref_losses.CalibrationLoss()(flat, yt[sl])), dtype=np.float64)
        for nb in (10, 7):      # other bin counts: torch.linspace(0, 1, nb + 1) boundaries (losses.py:459)
            out[f"calibration{nb}.{tag}"] = np.asarray(float(ref_losses.CalibrationLoss(n_bins=nb)(flat, yt[sl])), dtype=np.float64)
        # variant-1 loss (deer.py) per dimension 0, defaults and the self-test's kw=0.1
        for kw in (1.0, 0.1):
            l0 = ref_deer.DEERLoss(kl_weight=kw)({"mu": mu[sl, 0:1], "nu": nu[sl, 0:1], "alpha": alpha[sl, 0:1],
                                                   "beta": beta[sl, 0:1]}, yt[sl, 0])
            for k, val in l0.items():
                out[f"v1.kw{kw}.{tag}.{k}"] = np.asarray(float(val), dtype=np.float64)
    # gradient of the multitask loss wrt raw evidence on the regular rows
    er = et[5:64].clone().requires_grad_(True)
    pred = {}
    for i, d in enumerate(DIM_NAMES):
        pred[f"{d}_mu"] = er[:, i, 0:1]
        pred[f"{d}_nu"] = F.softplus(er[:, i, 1:2]) + 1e-6
        pred[f"{d}_alpha"] = F.softplus(er[:, i, 2:3]) + 1.0
        pred[f"{d}_beta"] = F.softplus(er[:, i, 3:4]) + 1e-6
    ref_losses.MultiTaskDEERLoss()(pred, yt[5:64])["total_loss"].backward()
    out["multitask.reg.devidence"] = tnp(er.grad)
    out["linspace11"] = tnp(torch.linspace(0, 1, 11))
    out["linspace16"] = tnp(torch.linspace(0, 1, 16))
    return out


def fill_module(mod, tag):
    """Closed-form fill of an arbitrary reference module (synth.module_fill); returns the numpy state."""
    sd = synth.module_fill(tag, {k: tuple(v.shape) for k, v in mod.state_dict().items()})
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd


def store_grads(out, tag, mod, inputs):
    """Gradients after a backward: inputs whole; parameters whole up to 20000 elements, else l2 norm + 1024 evenly spaced elements."""
    for k, x in inputs.items():
        out[f"{tag}.dx.{k}"] = tnp(x.grad)
    for name, prm in mod.named_parameters():
        if prm.grad is None:
            out[f"{tag}.gradnone.{name}"] = np.zeros(0, np.float32)
        elif prm.numel() > 20000:
            g = prm.grad.detach().double().reshape(-1)
            idx = torch.linspace(0, g.numel() - 1, 1024).round().long()
            out[f"{tag}.gradnorm.{name}"], out[f"{tag}.gradsample.{name}"] = np.float64(g.norm().item()), g[idx].numpy().astype(np.float32)
        else:
            out[f"{tag}.grad.{name}"] = tnp(prm.grad)


def capture_side():
    out = {}
    B = 9
    x = {k: synth.normal(500 + i, B * 256).reshape(B, 256).astype(np.float32)
         for i, k in enumerate(("audio", "video", "text"))}
    cma = ref_deer.CrossModalAttention(256, num_heads=8).eval()
    fill_module(cma, "cma")
    xt = {k: torch.from_numpy(x[k]).requires_grad_(True) for k in ("audio", "video", "text")}
    wa, wv = cma(xt["audio"], xt["video"], xt["text"])
    out["cma.audio"], out["cma.video"] = tnp(wa), tnp(wv)
    # backward of sum(wa * ca + wv * cv), ca / cv closed-form (synth.normal(520 / 521)): parameter and input gradients
    ca, cv = (torch.from_numpy(synth.normal(520 + i, B * 32).reshape(B, 32).astype(np.float32)) for i in range(2))
    ((wa * ca).sum() + (wv * cv).sum()).backward()
    store_grads(out, "cma", cma, xt)
    for k in x:
        out["cma.in." + k] = x[k]
    hd = ref_deer.HierarchicalDEERFusion().eval()
    fill_module(hd, "hdf")
    batch = synth.make_batch(B, seed=77)
    with torch.no_grad():
        out["hdf.audio_encoded"] = tnp(torch.relu(hd.audio_encoder(torch.from_numpy(batch["audio"]))))
        out["hdf.video_encoded"] = tnp(torch.relu(hd.video_encoder(torch.from_numpy(batch["video"]))))
        out["hdf.text_encoded"] = tnp(torch.relu(hd.text_encoder(torch.from_numpy(batch["text"]))))
    # EnhancedAudioEncoder feature branch: encoders.py imports librosa / cv2 at module
    # level; neither is installed and neither is touched by the feature branch, so empty
    # placeholder modules are enough to import the file.
    for missing in ("librosa", "cv2"):
        if missing not in sys.modules:
            m = types.ModuleType(missing)
            m.__spec__ = __import__("importlib.machinery").machinery.ModuleSpec(missing, None)
            sys.modules[missing] = m
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import encoders as ref_enc  # (reference)
        enc = ref_enc.EnhancedAudioEncoder().eval()
    fill_module(enc, "aenc")
    xa = torch.from_numpy(batch["audio"]).requires_grad_(True)
    y = enc(xa)
    out["aenc.out"] = tnp(y)
    (y * torch.from_numpy(synth.normal(530, y.numel()).reshape(y.shape).astype(np.float32))).sum().backward()
    store_grads(out, "aenc", enc, {"audio": xa})
    return out


def capture_stackb():
    """complete_project.CompleteDEERModel, eval forward."""
    import json
    import complete_project as ref_cp  # (reference)
    model = ref_cp.CompleteDEERModel(ref_cp.ModelConfig()).eval()
    fill_module(model, "stackb")
    B = 9
    batch = synth.make_batch(B, seed=78)
    with torch.no_grad():
        o = model(*(torch.from_numpy(batch[k]) for k in ("audio", "video", "text")))
    out = {"out." + k: tnp(v) for k, v in o.items()}
    # gradients of MultiTaskDEERLoss through the eval-mode model (every dropout site open): per parameter the sum, the
    # l2 norm and 256 evenly spaced elements of the flattened gradient -- digests, so the fixture stays small
    model.zero_grad()
    o = model(*(torch.from_numpy(batch[k]) for k in ("audio", "video", "text")))
    loss = ref_losses.MultiTaskDEERLoss()(o, torch.from_numpy(batch["targets"]))
    loss["total_loss"].backward()
    out["grad.total_loss"] = tnp(loss["total_loss"])
    for name, prm in model.named_parameters():
        if prm.grad is None:
            out["gradnone." + name] = np.zeros(0, np.float32)
            continue
        g = prm.grad.detach().double().reshape(-1)
        idx = torch.linspace(0, g.numel() - 1, min(256, g.numel())).round().long()
        out["gradsum." + name] = np.float64(g.sum().item())
        out["gradnorm." + name] = np.float64(g.norm().item())
        out["gradsample." + name] = g[idx].numpy().astype(np.float32)
    with open(os.path.join(HERE, "stackb_state_dict_names.json"), "w") as fh:
        json.dump({k: list(v.shape) for k, v in model.state_dict().items()}, fh, indent=0)
    return out


def metric_cases():
    """(predictions, targets, uncertainties) triples for the evaluation metrics: closed-form generators, so the test side
    rebuilds the same inputs from the stored arrays only."""
    cases = {}
    n = 777
    t = np.tanh(synth.normal(1301, n * 3).reshape(n, 3)).astype(np.float32)
    p = (t + 0.35 * synth.normal(1302, n * 3).reshape(n, 3) + np.array([0.05, -0.1, 0.0])).astype(np.float32)
    u = (0.05 + 0.6 * np.abs(synth.normal(1303, n * 3).reshape(n, 3))).astype(np.float32)
    cases["main"] = (p, t, u)
    p2, t2, u2 = p.copy(), t.copy(), u.copy()
    p2[5, 1] = np.nan; t2[17, 0] = np.nan; p2[40:44, 2] = np.nan; u2[77, 1] = np.inf; u2[300, 0] = np.nan
    cases["nan"] = (p2, t2, u2)
    cases["small"] = (p[:9].copy(), t[:9].copy(), u[:9].copy())            # fewer valid samples than bins: ECE = 1.0
    p3 = p[:200].copy(); p3[:, 2] = 0.25                                   # a constant column: CCC = 0.0
    u3 = u[:200].copy(); u3[:, :] = np.round(u3 * 8) / 8                   # heavily tied uncertainties: quantile edges on ties
    cases["ties"] = (p3, t[:200].copy(), u3)
    return cases


def capture_metrics():
    """DEERMetrics.concordance_correlation_coefficient / mean_absolute_error / root_mean_squared_error (reference
    src/utils/metrics.py:59-125) per emotion dimension and uncertainty_calibration_error (:214-279)."""
    import metrics as ref_metrics  # (reference)
    m = ref_metrics.DEERMetrics()
    out = {}
    for tag, (p, t, u) in metric_cases().items():
        out[f"{tag}.predictions"], out[f"{tag}.targets"], out[f"{tag}.uncertainties"] = p, t, u
        for i, d in enumerate(DIM_NAMES):
            out[f"{tag}.ccc_{d}"] = np.float64(m.concordance_correlation_coefficient(t[:, i], p[:, i]))
            out[f"{tag}.mae_{d}"] = np.float64(m.mean_absolute_error(t[:, i], p[:, i]))
            out[f"{tag}.rmse_{d}"] = np.float64(m.root_mean_squared_error(t[:, i], p[:, i]))
        out[f"{tag}.ece"] = np.float64(ref_metrics.uncertainty_calibration_error(p, t, u))
        out[f"{tag}.ece_15"] = np.float64(ref_metrics.uncertainty_calibration_error(p, t, u, n_bins=15))
    return out


FUSION_DIMS = [84, 256, 768]


def fusion_alt_modules():
    """(tag, reference module, state shapes) of the a5 row: the alternative fusions of fusion.py:421-554 and the two non-hierarchical
    branches of its factory (:579-592)."""
    return [("attention", ref_fusion.AttentionFusion(FUSION_DIMS, 256)),
            ("bilinear", ref_fusion.BilinearFusion(FUSION_DIMS, 256)),
            ("bilinear2", ref_fusion.BilinearFusion(FUSION_DIMS[:2], 256)),
            ("adaptive", ref_fusion.AdaptiveFusionGating(FUSION_DIMS, ["attention", "bilinear"], 256)),
            ("factory_attention", ref_fusion.create_fusion_module("attention", {})),
            ("factory_concat", ref_fusion.create_fusion_module("concatenation", {"input_dims": FUSION_DIMS}))]


def capture_fusion_alt():
    """Eval-mode outputs of every module above at B = 9 with closed-form parameters, and the gradients of sum(out * c) (c a
    fixed closed-form tensor) with respect to every parameter and every input, stored whole (the modules are small except for
    the bilinear weight, which is stored as digests)."""
    import json
    out, shapes = {}, {}
    B = 9
    for tag, mod in fusion_alt_modules():
        mod.eval()
        sd = fill_module(mod, "fa_" + tag)
        shapes[tag] = {k: list(v.shape) for k, v in sd.items()}
        dims = [256, 256, 256] if tag == "factory_attention" else (FUSION_DIMS[:2] if tag == "bilinear2" else FUSION_DIMS)
        xs = [torch.from_numpy(synth.normal(700 + 10 * len(tag) + i, B * d).reshape(B, d).astype(np.float32)).requires_grad_(True)
              for i, d in enumerate(dims)]
        if tag == "adaptive":
            o = mod(*xs)
            y, extra = o["fused_features"], {"strategy_weights": o["strategy_weights"]}
        elif tag == "factory_concat":
            y, extra = mod(torch.cat(xs, dim=-1)), {}
        else:
            y, extra = mod(xs), {}
        c = torch.from_numpy(synth.normal(990 + len(tag), y.numel()).reshape(y.shape).astype(np.float32))
        (y * c).sum().backward()
        out[f"{tag}.out"] = tnp(y)          # inputs and c are closed-form: the tests rebuild them (fusion_alt_inputs)
        for k, v in extra.items():
            out[f"{tag}.{k}"] = tnp(v)
        for i, x in enumerate(xs):
            out[f"{tag}.dx{i}"] = tnp(x.grad)
        for name, prm in mod.named_parameters():
            if prm.grad is None:
                out[f"{tag}.gradnone.{name}"] = np.zeros(0, np.float32)
            elif prm.numel() > 20000:
                g = prm.grad.detach().double().reshape(-1)
                idx = torch.linspace(0, g.numel() - 1, 1024).round().long()
                out[f"{tag}.gradnorm.{name}"], out[f"{tag}.gradsample.{name}"] = np.float64(g.norm().item()), g[idx].numpy().astype(np.float32)
            else:
                out[f"{tag}.grad.{name}"] = tnp(prm.grad)
    with open(os.path.join(HERE, "fusion_alt_state_dict_names.json"), "w") as fh:
        json.dump(shapes, fh, indent=0)
    return out


FUSION_GEOM = dict(audio_dim=40, video_dim=128, text_dim=300)


def capture_fusion_geom():
    """HierarchicalMultimodalFusion on ANOTHER geometry than the default: audio 40 / video 128 / text 300
    (fusion 512, intermediate 256, 8 heads), closed-form parameters, B = 9.  Eval-mode outputs; and with dropout = 0.0 in train mode
    the gradients of sum(fused * c0 + audiovisual * c1 + trimodal * c2) with respect to every parameter (digests for the large
    matrices) and every input."""
    import json
    out = {}
    B = 9
    xs = {k: synth.normal(810 + i, B * d).reshape(B, d).astype(np.float32) for i, (k, d) in enumerate(FUSION_GEOM.items())}
    shapes = None
    for mode in ("eval", "train"):
        mod = ref_fusion.HierarchicalMultimodalFusion(fusion_dim=512, intermediate_dim=256, num_attention_heads=8,
                                                      dropout=0.3 if mode == "eval" else 0.0, use_uncertainty_weighting=True, **FUSION_GEOM)
        sd = fill_module(mod, "fgeom")
        shapes = {k: list(v.shape) for k, v in sd.items()}
        ins = {k: torch.from_numpy(v).requires_grad_(mode == "train") for k, v in xs.items()}
        if mode == "eval":
            mod.eval()
            with torch.no_grad():
                o = mod(ins["audio_dim"], ins["video_dim"], ins["text_dim"])
            for k in ("fused_features", "audiovisual_features", "trimodal_features", "trimodal_attention_weights"):
                out["eval." + k] = tnp(o[k])
            out["eval.av_attention.audio_to_video"] = tnp(o["av_attention_weights"]["audio_to_video"])
            out["eval.av_attention.video_to_audio"] = tnp(o["av_attention_weights"]["video_to_audio"])
        else:
            mod.train()
            o = mod(ins["audio_dim"], ins["video_dim"], ins["text_dim"])
            loss = 0.0
            for j, k in enumerate(("fused_features", "audiovisual_features", "trimodal_features")):
                c = torch.from_numpy(synth.normal(830 + j, o[k].numel()).reshape(o[k].shape).astype(np.float32))
                loss = loss + (o[k] * c).sum()
            loss.backward()
            out["train.loss"] = np.float64(float(loss))
            store_grads(out, "train", mod, {k: v for k, v in ins.items()})
    with open(os.path.join(HERE, "fusion_geom_state_dict_names.json"), "w") as fh:
        json.dump(shapes, fh, indent=0)
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "fusion_geom":
        np.savez_compressed(os.path.join(HERE, "fusion_geom.npz"), **capture_fusion_geom())
        print("fusion_geom.npz", os.path.getsize(os.path.join(HERE, "fusion_geom.npz")), "bytes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "side":
        np.savez_compressed(os.path.join(HERE, "side_kernels.npz"), **capture_side())
        print("side_kernels.npz", os.path.getsize(os.path.join(HERE, "side_kernels.npz")), "bytes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fusion_alt":
        np.savez_compressed(os.path.join(HERE, "fusion_alt.npz"), **capture_fusion_alt())
        print("fusion_alt.npz", os.path.getsize(os.path.join(HERE, "fusion_alt.npz")), "bytes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "losses":
        np.savez_compressed(os.path.join(HERE, "loss_cases.npz"), **capture_losses())
        print("loss_cases.npz", os.path.getsize(os.path.join(HERE, "loss_cases.npz")), "bytes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **capture_metrics())
        print("metrics_cases.npz", os.path.getsize(os.path.join(HERE, "metrics_cases.npz")), "bytes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "stackb":
        np.savez_compressed(os.path.join(HERE, "stackb_B9.npz"), **capture_stackb())
        print("stackb_B9.npz", os.path.getsize(os.path.join(HERE, "stackb_B9.npz")), "bytes")
        return
    np.savez_compressed(os.path.join(HERE, "stackb_B9.npz"), **capture_stackb())
    for B in (1, 7, 32):
        np.savez_compressed(os.path.join(HERE, f"stackc_B{B}.npz"), **capture_stackc(B, seed=42))
    miss = {}
    for tag, zero in (("audio_only", ("video", "text")), ("text_only", ("audio", "video"))):
        for k, v in capture_stackc(8, seed=43, zero=zero).items():
            if k.startswith("eval.") or k.startswith("loss.") or k.startswith("gnorm."):
                miss[f"{tag}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "stackc_missing.npz"), **miss)
    np.savez_compressed(os.path.join(HERE, "loss_cases.npz"), **capture_losses())
    np.savez_compressed(os.path.join(HERE, "side_kernels.npz"), **capture_side())
    np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **capture_metrics())
    import json
    f, h = build_reference(0.3)
    names = {"fusion": list(f.state_dict().keys()), "head": list(h.state_dict().keys()),
             "shapes": {**{"fusion." + k: list(v.shape) for k, v in f.state_dict().items()},
                        **{"head." + k: list(v.shape) for k, v in h.state_dict().items()}}}
    with open(os.path.join(HERE, "state_dict_names.json"), "w") as fh:
        json.dump(names, fh, indent=0)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
