"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement (torch-CPU tensor arithmetic, fp32 or fp64) of the
reference's multimodal-fusion + DEER forward and loss; gradients come from
torch autograd over this restatement.  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package (``mmdeer``) never does and fails loudly without its HIP library.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's
``fusion.py`` / ``deer.py`` / ``losses.py`` in the build container, loads the
closed-form parameters of ``mmdeer.synth`` through ``load_state_dict`` and stores
the reference's outputs, loss components and gradient digests under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against
them.  Rows the reference itself cannot execute (a4
``UncertaintyAwareGating`` call site, a9 ``HierarchicalDEERFusion`` as a whole)
are "parity unpinned" and are not restated here beyond their working pieces.

All ``file:line`` citations are relative to the reference checkout.
Parameter dictionaries are keyed by the reference's state_dict names with the
``fusion.`` / ``head.`` prefixes of ``mmdeer.spec.param_table``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

DIM_NAMES = ("valence", "arousal", "dominance")

# torch.linspace(0, 1, 11) in fp32 == float32(i)/10 exactly.
ECE_EDGES_10 = [float.fromhex(h) for h in (
    "0x0p+0", "0x1.99999ap-4", "0x1.99999ap-3", "0x1.333334p-2", "0x1.99999ap-2",
    "0x1p-1", "0x1.333334p-1", "0x1.666666p-1", "0x1.99999ap-1", "0x1.ccccccp-1", "0x1p+0")]
# torch.linspace(0, 1, 16) in fp32 -- NOT float32(i)/15.
CAL_EDGES_15 = [float.fromhex(h) for h in (
    "0x0p+0", "0x1.111112p-4", "0x1.111112p-3", "0x1.99999cp-3", "0x1.111112p-2",
    "0x1.555556p-2", "0x1.99999cp-2", "0x1.dddde0p-2", "0x1.111110p-1", "0x1.333332p-1",
    "0x1.555554p-1", "0x1.777778p-1", "0x1.99999ap-1", "0x1.bbbbbcp-1", "0x1.dddddep-1",
    "0x1p+0")]


# --------------------------------------------------------------------------- helpers
def _lin(x, P, name):
    """nn.Linear: y = x W^T + b."""
    return x @ P[name + ".weight"].t() + P[name + ".bias"]


def _layer_norm(x, g, b, eps=1e-5):
    """nn.LayerNorm over the last dim, biased variance, eps inside the sqrt."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _drop(x, mask, p):
    """Inverted dropout with an explicit keep-mask (None = eval mode)."""
    if mask is None:
        return x
    return x * mask.to(x.dtype) / (1.0 - p)


def _relu_drop_ln(x, P, prefix, mask, p):
    """nn.Sequential(Linear, ReLU, Dropout, LayerNorm) -- fusion.py:98-103, 216-221, 301-306."""
    y = _drop(torch.relu(_lin(x, P, prefix + ".0")), mask, p)
    return _layer_norm(y, P[prefix + ".3.weight"], P[prefix + ".3.bias"]), y


# --------------------------------------------------------------------------- fusion
def av_fusion(P, audio, video, masks=None, p=0.3, heads=8):
    """AudioVisualFusion.forward -- fusion.py:223-271.

    The shared nn.MultiheadAttention is called with L = S = 1, so softmax over the
    single key is 1 and each call reduces to out_proj(v_proj(key)); Q and K
    projections are dead (their gradients are exact zeros).  In train mode torch
    applies dropout to the (B*heads, 1, 1) attention weights, i.e. one keep/scale
    factor per (sample, head) on that head's slice of V; the returned weights are
    the head-mean of the post-dropout weights.
    """
    masks = masks or {}
    pre = "fusion.audio_visual_fusion."
    ap = _lin(audio, P, pre + "audio_projection")          # :236
    vp = _lin(video, P, pre + "video_projection")          # :237
    E = ap.shape[1]
    hd = E // heads
    w_in, b_in = P[pre + "cross_attention.in_proj_weight"], P[pre + "cross_attention.in_proj_bias"]
    w_v, b_v = w_in[2 * E:], b_in[2 * E:]                  # packed rows are [q; k; v]

    def attend(kv, mask):                                  # :244-255
        v = kv @ w_v.t() + b_v
        B = v.shape[0]
        wgt = torch.ones(B, heads, dtype=v.dtype)
        wgt = _drop(wgt, mask, p)
        v = (v.view(B, heads, hd) * wgt[:, :, None]).reshape(B, E)
        out = _lin(v, P, pre + "cross_attention.out_proj")
        return out, wgt.mean(dim=1, keepdim=True)

    audio_att, w_a2v = attend(vp, masks.get("av_attn_a2v"))  # query=audio, key=value=video
    video_att, w_v2a = attend(ap, masks.get("av_attn_v2a"))  # query=video, key=value=audio
    cat = torch.cat([audio_att, video_att], dim=-1)        # :262
    fused, _ = _relu_drop_ln(cat, P, pre + "fusion_layers", masks.get("av_fuse"), p)  # :263
    return {"fused_features": fused,
            "attention_weights": {"audio_to_video": w_a2v, "video_to_audio": w_v2a}}


def trimodal_fusion(P, av, text, masks=None, p=0.3, heads=8):
    """TrimodalFusion.forward -- fusion.py:308-343.

    Self-attention over the 2-token sequence [av_proj, text_proj]:
    q is scaled by sqrt(1/head_dim) before the product (torch functional
    multi_head_attention_forward), softmax over the 2 keys, dropout on the
    probabilities, PV, out_proj; returned weights are the head-mean of the
    post-dropout probabilities.
    """
    masks = masks or {}
    pre = "fusion.trimodal_fusion."
    x0 = _lin(av, P, pre + "audiovisual_projection")       # :321
    x1 = _lin(text, P, pre + "text_projection")            # :322
    x = torch.stack([x0, x1], dim=1)                       # (B, 2, E)  :325
    B, T, E = x.shape
    hd = E // heads
    qkv = x @ P[pre + "modality_attention.in_proj_weight"].t() + P[pre + "modality_attention.in_proj_bias"]
    q, k, v = qkv.split(E, dim=-1)
    q = q.view(B, T, heads, hd).transpose(1, 2) * math.sqrt(1.0 / hd)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    prob = torch.softmax(q @ k.transpose(-1, -2), dim=-1)  # (B, H, 2, 2)
    prob = _drop(prob, masks.get("tri_attn"), p)
    o = (prob @ v).transpose(1, 2).reshape(B, T, E)
    y = _lin(o, P, pre + "modality_attention.out_proj")    # :328-332
    pooled = y.mean(dim=1)                                 # :335
    fused, _ = _relu_drop_ln(pooled, P, pre + "final_fusion", masks.get("tri_fuse"), p)  # :338
    return {"fused_features": fused, "attention_weights": prob.mean(dim=1)}


def fusion_forward(P, audio, video, text, masks=None, p=0.3, heads=8):
    """HierarchicalMultimodalFusion.forward with uncertainties=None -- fusion.py:119-171."""
    masks = masks or {}
    av = av_fusion(P, audio, video, masks, p, heads)
    tri = trimodal_fusion(P, av["fused_features"], text, masks, p, heads)
    final, _ = _relu_drop_ln(tri["fused_features"], P, "fusion.output_projection",
                             masks.get("out_proj"), p)    # :162
    return {"fused_features": final,
            "audiovisual_features": av["fused_features"],
            "trimodal_features": tri["fused_features"],
            "av_attention_weights": av["attention_weights"],
            "trimodal_attention_weights": tri["attention_weights"],
            "uncertainty_weights": None}


# --------------------------------------------------------------------------- DEER head
def nig_activations(e):
    """DEERLayer.forward tail -- deer.py:90-98.  e: (..., 4) = [mu, nu^, alpha^, beta^]."""
    mu = e[..., 0]
    nu = F.softplus(e[..., 1]) + 1e-6
    alpha = F.softplus(e[..., 2]) + 1.0
    beta = F.softplus(e[..., 3]) + 1e-6
    alea = beta / (alpha - 1)
    epis = beta / (nu * (alpha - 1))
    return mu, nu, alpha, beta, alea, epis, alea + epis


def deer_head_forward(P, fused, masks=None, p=0.3):
    """MultiDimensionalDEER.forward -- deer.py:233-266 (3 x DEERLayer deer.py:68-108)."""
    masks = masks or {}
    h = _drop(torch.relu(_lin(fused, P, "head.feature_processor.0")), masks.get("fp0"), p)
    h = _drop(torch.relu(_lin(h, P, "head.feature_processor.3")), masks.get("fp1"), p)  # :246
    out: Dict[str, torch.Tensor] = {}
    keys = ("mu", "nu", "alpha", "beta", "aleatoric_uncertainty", "epistemic_uncertainty", "uncertainty")
    m0, m1 = masks.get("ev0"), masks.get("ev1")
    for i, dim in enumerate(DIM_NAMES):
        pre = f"head.deer_heads.{i}.evidence_net"
        e = _drop(torch.relu(_lin(h, P, pre + ".0")), None if m0 is None else m0[:, i], p)
        e = _drop(torch.relu(_lin(e, P, pre + ".3")), None if m1 is None else m1[:, i], p)
        e = _lin(e, P, pre + ".6").view(h.shape[0], 1, 4)   # deer.py:86-87
        for key, val in zip(keys, nig_activations(e)):
            out[f"{dim}_{key}"] = val
    out["mu_all"] = torch.cat([out[f"{d}_mu"] for d in DIM_NAMES], dim=1)           # :258
    out["uncertainty_all"] = torch.cat([out[f"{d}_uncertainty"] for d in DIM_NAMES], dim=1)
    return out


def model_forward(P, audio, video, text, masks=None, p=0.3, heads=8):
    """The Stack-C composite (SURVEY 8b "Definition of MultimodalDEER")."""
    fo = fusion_forward(P, audio, video, text, masks, p, heads)
    ho = deer_head_forward(P, fo["fused_features"], masks, p)
    return fo, ho


# --------------------------------------------------------------------------- losses
def ece_term(gamma, alpha, beta, targets, eps=1e-8, edges: Sequence[float] = ECE_EDGES_10):
    """losses.DEERLoss._compute_ece_loss -- losses.py:187-226.  Bin i is (lo, hi]."""
    err = (targets - gamma).abs().flatten()
    conf = (1.0 / (1.0 + beta / (alpha - 1 + eps))).flatten()
    n = conf.numel()
    total = torch.zeros((), dtype=gamma.dtype)
    counts = []
    for i in range(len(edges) - 1):
        lo = torch.tensor(edges[i], dtype=torch.float32).to(conf.dtype)
        hi = torch.tensor(edges[i + 1], dtype=torch.float32).to(conf.dtype)
        in_bin = (conf > lo) & (conf <= hi)
        c = int(in_bin.sum())
        counts.append(c)
        if c > 0:
            total = total + (c / n) * (conf[in_bin].mean() - (1.0 - err[in_bin].mean())).abs()
    return total, counts


def deer_loss_v2(gamma, nu, alpha, beta, targets, reg_w=0.1, kl_w=0.01, ece_w=0.05, eps=1e-8):
    """losses.DEERLoss.forward -- losses.py:72-130 (terms :132-226)."""
    if targets.dim() == 1 and gamma.dim() == 2:           # :98-104
        targets = targets.unsqueeze(-1)
    elif targets.dim() == 2 and gamma.dim() == 1:
        gamma, nu, alpha, beta = (t.unsqueeze(-1) for t in (gamma, nu, alpha, beta))
    err = targets - gamma
    log_prob = (0.5 * torch.log(nu / (2 * math.pi + eps))
                + alpha * torch.log(beta + eps)
                - torch.lgamma(alpha + eps)
                - (alpha + 0.5) * torch.log(beta + 0.5 * nu * err.pow(2) + eps))   # :144-150
    nll = -log_prob.mean()
    e = err.abs()
    reg = (e.pow(2) * (2 * beta + nu * e.pow(2))).mean()                          # :165-167
    kl = ((alpha - 1).pow(2)).mean() + 0.1 * ((torch.log(beta + eps) - math.log(1 + eps)).pow(2)).mean()
    if ece_w > 0:
        ece, counts = ece_term(gamma, alpha, beta, targets, eps)
    else:
        ece, counts = torch.zeros((), dtype=gamma.dtype), []
    total = nll + reg_w * reg + kl_w * kl + ece_w * ece                           # :121
    return {"total_loss": total, "nll_loss": nll, "reg_loss": reg, "kl_loss": kl,
            "ece_loss": ece, "batch_size": gamma.shape[0], "_bin_counts": counts}


def multitask_loss(pred, targets, task_weights=(1.0, 1.0, 1.0), cross_w=0.05, **kw):
    """losses.MultiTaskDEERLoss.forward -- losses.py:268-348."""
    out: Dict[str, torch.Tensor] = {}
    total = 0.0
    for i, dim in enumerate(DIM_NAMES):
        g = pred[f"{dim}_gamma"] if f"{dim}_gamma" in pred else pred[f"{dim}_mu"]
        n = pred[f"{dim}_nu"] if f"{dim}_nu" in pred else pred[f"{dim}_lambda"]
        d = deer_loss_v2(g, n, pred[f"{dim}_alpha"], pred[f"{dim}_beta"], targets[:, i:i + 1], **kw)
        total = total + task_weights[i] * d["total_loss"]
        for k, v in d.items():
            out[f"{dim}_{k}"] = v
    if cross_w > 0:
        u = [(pred[f"{d}_beta"] / (pred[f"{d}_alpha"] - 1 + 1e-8)).mean(dim=0) for d in DIM_NAMES]
        cross, pairs = 0.0, 0
        for i in range(3):
            for j in range(i + 1, 3):
                cross = cross + ((u[i] - u[j]) ** 2).mean()
                pairs += 1
        cross = cross / pairs                                                      # :339-346
        total = total + cross_w * cross
        out["cross_dim_loss"] = cross
    out["total_loss"] = total / 3                                                  # :314
    return out


def deer_loss_v1(mu, nu, alpha, beta, targets, evidence_weight=1.0, kl_weight=1.0):
    """deer.DEERLoss.forward -- deer.py:125-195 (loss variant 1)."""
    if targets.dim() == 1:
        targets = targets.unsqueeze(-1)
    se = (targets - mu) ** 2
    nll = (0.5 * torch.log(math.pi / nu) - alpha * torch.log(2 * beta) + torch.lgamma(alpha)
           - torch.lgamma(alpha + 0.5) + (alpha + 0.5) * torch.log(beta + nu * se / 2))
    reg = (nu * se + 2 * beta * (1 + nu)) / (2 * nu * (1 + nu))
    kl = (0.5 * (nu - 1) + alpha * torch.log(beta) - torch.lgamma(alpha)
          + torch.lgamma(alpha + 0.5) - 0.5 * torch.log(2 * math.pi * beta)).clamp(min=0)
    return {"total_loss": nll.mean() + evidence_weight * reg.mean() + kl_weight * kl.mean(),
            "nll_loss": nll.mean(), "evidence_reg": reg.mean(), "kl_reg": kl.mean(), "mse": se.mean()}


def uncertainty_reg_loss(alpha, beta, diversity_weight=0.1, sparsity_weight=0.01):
    """losses.UncertaintyRegularizationLoss with flat keys -- losses.py:363-416."""
    u = beta / (alpha - 1 + 1e-8)
    div = -torch.log(torch.var(u, dim=0).mean() + 1e-8)      # unbiased variance over the batch
    spars = u.mean()
    return {"reg_loss": diversity_weight * div + sparsity_weight * spars,
            "diversity_loss": div, "sparsity_loss": spars}


def calibration_loss(gamma, alpha, beta, targets, edges: Sequence[float] = CAL_EDGES_15):
    """losses.CalibrationLoss.forward, uniform bins -- losses.py:431-497.
    Bin i is [lo, hi), the last bin [lo, hi]."""
    err = (targets - gamma).abs().flatten()
    conf = (1.0 / (1.0 + beta / (alpha - 1 + 1e-8))).flatten()
    acc = 1.0 - (err / 2.0).clamp(0, 1)
    n = conf.numel()
    nb = len(edges) - 1
    total = torch.zeros((), dtype=gamma.dtype)
    for i in range(nb):
        lo = torch.tensor(edges[i], dtype=torch.float32).to(conf.dtype)
        hi = torch.tensor(edges[i + 1], dtype=torch.float32).to(conf.dtype)
        in_bin = (conf >= lo) & ((conf <= hi) if i == nb - 1 else (conf < hi))
        c = int(in_bin.sum())
        if c > 0:
            total = total + (c / n) * (conf[in_bin].mean() - acc[in_bin].mean()).abs()
    return total


# --------------------------------------------------------------------------- side kernels
def cross_modal_attention(P, audio, video, text, heads=8, prefix=""):
    """deer.CrossModalAttention.forward -- deer.py:379-425.  Softmax runs over the
    HEAD axis (dim=1) and the weighted sum collapses heads, giving (B, head_dim)."""
    B, E = audio.shape
    hd = E // heads
    lin = lambda x, n: x @ P[prefix + n + ".weight"].t() + P[prefix + n + ".bias"]
    q = lin(text, "query_proj").view(B, heads, hd)
    outs = []
    for m in (audio, video):
        k = lin(m, "key_proj").view(B, heads, hd)
        v = lin(m, "value_proj").view(B, heads, hd)
        s = (q * k).sum(dim=2) / math.sqrt(hd)
        a = torch.softmax(s, dim=1)
        outs.append((a.unsqueeze(2) * v).sum(dim=1))
    ctx = torch.cat([audio, video, text], dim=1)
    g = torch.relu(ctx @ P[prefix + "uncertainty_gate.0.weight"].t() + P[prefix + "uncertainty_gate.0.bias"])
    g = torch.softmax(g @ P[prefix + "uncertainty_gate.2.weight"].t() + P[prefix + "uncertainty_gate.2.bias"], dim=1)
    return outs[0] * g[:, 0:1], outs[1] * g[:, 1:2]


def modality_encoders(P, audio, video, text, prefix=""):
    """The three ReLU(Linear) encoders of deer.HierarchicalDEERFusion -- deer.py:330-332."""
    enc = lambda x, n: torch.relu(x @ P[prefix + n + ".weight"].t() + P[prefix + n + ".bias"])
    return enc(audio, "audio_encoder"), enc(video, "video_encoder"), enc(text, "text_encoder")


def lstm_t1_bidir(x, P, prefix, layers=2, hidden=256):
    """nn.LSTM(bidirectional, batch_first) evaluated at T = 1 with zero initial state
    (encoders.py:82-89, 380): gates = W_ih x + b_ih + b_hh (W_hh multiplies h0 = 0),
    c = sigmoid(i) * tanh(g), h = sigmoid(o) * tanh(c); gate row order is [i; f; g; o]."""
    inp = x
    for l in range(layers):
        outs = []
        for sfx in ("", "_reverse"):
            g = (inp @ P[f"{prefix}weight_ih_l{l}{sfx}"].t()
                 + P[f"{prefix}bias_ih_l{l}{sfx}"] + P[f"{prefix}bias_hh_l{l}{sfx}"])
            i, _f, gg, o = g.split(hidden, dim=-1)
            c = torch.sigmoid(i) * torch.tanh(gg)
            outs.append(torch.sigmoid(o) * torch.tanh(c))
        inp = torch.cat(outs, dim=-1)
    return inp


def audio_encoder_features(P, x, prefix="", hidden=256):
    """EnhancedAudioEncoder.forward, pre-extracted-feature branch, eval mode --
    encoders.py:368-389.  x: (B, 84) -> (B, 512).  With T = 1 the attention pool's
    softmax over time is 1, so pooled == lstm_out[:, 0]."""
    h = lstm_t1_bidir(x, P, prefix + "lstm.", 2, hidden)
    y = torch.relu(h @ P[prefix + "output_projection.0.weight"].t() + P[prefix + "output_projection.0.bias"])
    y = y @ P[prefix + "output_projection.3.weight"].t() + P[prefix + "output_projection.3.bias"]
    return _layer_norm(y, P[prefix + "output_projection.4.weight"], P[prefix + "output_projection.4.bias"])


# --------------------------------------------------------------------------- Stack B
# complete_project.CompleteDEERModel, eval-mode forward.  Parameter dict keyed by the reference's state_dict names
# (audio_encoder.*, attention_module.*, fusion_module.*, prediction_heads.<dim>.*, calibration_layer.*).
def _stackb_encoder(P, x, prefix, layers=3):
    """EnhancedModalityEncoder -- complete_project.py:76-117: Linear-ReLU-LN, `layers` x (x + LN(ReLU(Lin x))), Linear."""
    h = _layer_norm(torch.relu(_lin(x, P, prefix + ".input_projection.0")),
                    P[prefix + ".input_projection.2.weight"], P[prefix + ".input_projection.2.bias"])
    for i in range(layers):
        q = f"{prefix}.encoder_layers.{i}.layers"          # ResidualBlock, complete_project.py:60-73
        h = h + _layer_norm(torch.relu(_lin(h, P, q + ".0")), P[q + ".3.weight"], P[q + ".3.bias"])
    return _lin(h, P, prefix + ".output_projection")


def _stackb_mha(P, prefix, query, key, value, heads=8):
    """MultiHeadAttention on (B, S, D) tensors -- complete_project.py:120-184 (eval: no dropout, no mask)."""
    Bq, _, D = query.shape
    hd = D // heads
    split = lambda t: t.view(Bq, -1, heads, hd).transpose(1, 2)
    Q, K, V = split(_lin(query, P, prefix + ".query_proj")), split(_lin(key, P, prefix + ".key_proj")), \
        split(_lin(value, P, prefix + ".value_proj"))
    w = torch.softmax(Q @ K.transpose(-2, -1) / math.sqrt(hd), dim=-1)
    o = (w @ V).transpose(1, 2).contiguous().view(Bq, -1, D)
    return _lin(o, P, prefix + ".output_proj")


def _stackb_uncertainty(P, x, prefix):
    """UncertaintyEstimator -- complete_project.py:187-213: D -> D/2 -> D/4 -> 1, sigmoid."""
    h = torch.relu(_lin(x, P, prefix + ".0"))
    h = torch.relu(_lin(h, P, prefix + ".3"))
    return torch.sigmoid(_lin(h, P, prefix + ".5"))


def stackb_attention(P, audio, video, text, heads=8, prefix="attention_module"):
    """UncertaintyAwareAttention -- complete_project.py:216-304."""
    a3, v3, t3 = audio.unsqueeze(1), video.unsqueeze(1), text.unsqueeze(1)
    ue = prefix + ".uncertainty_estimator.estimator"
    ua, uv, ut = (_stackb_uncertainty(P, x, ue) for x in (audio, video, text))
    sa = prefix + ".self_attention"
    a_self, v_self, t_self = (_stackb_mha(P, sa, x, x, x, heads).squeeze(1) for x in (a3, v3, t3))
    ca = prefix + ".cross_attention"      # text is the query, the modality itself key and value (:270-273)
    a_cross, v_cross, t_cross = (_stackb_mha(P, ca, t3, x, x, heads).squeeze(1) for x in (a3, v3, t3))
    wi = torch.cat([a_self, v_self, t_self, ua, uv, ut], dim=1)
    wn = prefix + ".weight_network"
    w = torch.softmax(_lin(torch.relu(_lin(wi, P, wn + ".0")), P, wn + ".3"), dim=1)
    return {"audio": w[:, 0:1] * a_self + (1 - ua) * a_cross,
            "video": w[:, 1:2] * v_self + (1 - uv) * v_cross,
            "text": w[:, 2:3] * t_self + (1 - ut) * t_cross,
            "attention_weights": w, "modality_uncertainties": torch.cat([ua, uv, ut], dim=1)}


def stackb_fusion(P, audio, video, text, prefix="fusion_module"):
    """HierarchicalFusionModule -- complete_project.py:307-366."""
    def stage(x, q):   # Linear, ReLU, Dropout, LayerNorm, Linear, ReLU
        h = _layer_norm(torch.relu(_lin(x, P, q + ".0")), P[q + ".3.weight"], P[q + ".3.bias"])
        return torch.relu(_lin(h, P, q + ".4"))
    av = stage(torch.cat([audio, video], dim=1), prefix + ".av_fusion")
    tri_in = torch.cat([av, text], dim=1)
    gate = torch.sigmoid(_lin(tri_in, P, prefix + ".fusion_gate.0"))
    tri = stage(tri_in, prefix + ".trimodal_fusion")
    return gate * tri + (1 - gate) * av


def stackb_head(P, x, prefix):
    """DEERPredictionHead -- complete_project.py:369-418."""
    q = prefix + ".evidence_network"
    e = _lin(torch.relu(_lin(torch.relu(_lin(x, P, q + ".0")), P, q + ".3")), P, q + ".6")
    mu = e[:, 0]
    nu = F.softplus(e[:, 1]) + 1e-6
    alpha = F.softplus(e[:, 2]) + 1.0
    beta = F.softplus(e[:, 3]) + 1e-6
    alea = beta / (alpha - 1)
    epi = beta / (nu * (alpha - 1))
    return {"mu": mu, "nu": nu, "alpha": alpha, "beta": beta, "aleatoric_uncertainty": alea,
            "epistemic_uncertainty": epi, "uncertainty": alea + epi}


def stackb_calibration(P, unc, prefix="calibration_layer"):
    """UncertaintyCalibrationLayer -- complete_project.py:421-459: temperature, then a shared 1-32-16-1 MLP per entry."""
    s = unc / P[prefix + ".temperature"].unsqueeze(0)
    q = prefix + ".calibration_network"
    cols = []
    for i in range(unc.shape[1]):
        h = torch.relu(_lin(s[:, i:i + 1], P, q + ".0"))
        h = torch.relu(_lin(h, P, q + ".2"))
        cols.append(torch.sigmoid(_lin(h, P, q + ".4")))
    return torch.cat(cols, dim=1)


def stackb_forward(P, audio, video, text, heads=8, layers=3):
    """CompleteDEERModel.forward -- complete_project.py:517-589."""
    enc = [_stackb_encoder(P, x, n, layers) for x, n in
           ((audio, "audio_encoder"), (video, "video_encoder"), (text, "text_encoder"))]
    att = stackb_attention(P, *enc, heads=heads)
    fused = stackb_fusion(P, att["audio"], att["video"], att["text"])
    out = {}
    for d in DIM_NAMES:
        for k, v in stackb_head(P, fused, f"prediction_heads.{d}").items():
            out[f"{d}_{k}"] = v
    out["mu_all"] = torch.stack([out[f"{d}_mu"] for d in DIM_NAMES], dim=1)
    out["uncertainty_all"] = torch.stack([out[f"{d}_uncertainty"] for d in DIM_NAMES], dim=1)
    out["calibrated_uncertainty"] = stackb_calibration(P, out["uncertainty_all"])
    out["attention_weights"] = att["attention_weights"]
    out["modality_uncertainties"] = att["modality_uncertainties"]
    out["fused_features"] = fused
    out["encoded"] = enc
    out["attended"] = [att["audio"], att["video"], att["text"]]
    return out


# --------------------------------------------------------------------------- a5: the alternative fusions (fusion.py:421-592)
def attention_fusion(P, feats, prefix=""):
    """AttentionFusion.forward -- fusion.py:515-528.  Returns (weighted sum, attention weights)."""
    st = torch.stack([_lin(f, P, f"{prefix}projections.{i}") for i, f in enumerate(feats)], dim=1)
    w = torch.softmax(_lin(st, P, prefix + "attention").squeeze(-1), dim=-1)
    return (w.unsqueeze(-1) * st).sum(dim=1), w


def bilinear_fusion(P, feats, prefix=""):
    """BilinearFusion.forward -- fusion.py:545-554 (nn.Bilinear: y_o = x1^T W_o x2 + b_o)."""
    if len(feats) < 2:
        return _lin(feats[0], P, prefix + "linear")
    y = torch.einsum("bi,oij,bj->bo", feats[0], P[prefix + "bilinear.weight"], feats[1]) + P[prefix + "bilinear.bias"]
    if len(feats) > 2:
        y = y + _lin(torch.cat(list(feats[2:]), dim=-1), P, prefix + "additional_linear")
    return y


def adaptive_fusion(P, feats, strategies, prefix=""):
    """AdaptiveFusionGating.forward in eval mode -- fusion.py:458-501 (strategies 'attention' / 'bilinear')."""
    cat = torch.cat(list(feats), dim=-1)
    h = torch.relu(_lin(torch.relu(_lin(cat, P, prefix + "feature_encoder.0")), P, prefix + "feature_encoder.3"))
    w = torch.softmax(_lin(h, P, prefix + "strategy_selector.0"), dim=-1)
    outs = []
    for name in strategies:
        if name == "attention":
            outs.append(attention_fusion(P, feats, prefix + "fusion_modules.attention.")[0])
        elif name == "bilinear":
            outs.append(bilinear_fusion(P, feats, prefix + "fusion_modules.bilinear."))
    return (w.unsqueeze(-1) * torch.stack(outs, dim=1)).sum(dim=1), w


def concat_fusion(P, x, prefix=""):
    """The concatenation branch of create_fusion_module in eval mode -- fusion.py:584-592: LayerNorm(ReLU(Linear(x)))."""
    return _layer_norm(torch.relu(_lin(x, P, prefix + "0")), P[prefix + "3.weight"], P[prefix + "3.bias"])


# --------------------------------------------------------------------------- metric
def ccc(x, y):
    """Concordance correlation coefficient, population variance -- metrics.py:85-101."""
    x = x.double().flatten()
    y = y.double().flatten()
    mx, my = x.mean(), y.mean()
    vx, vy = ((x - mx) ** 2).mean(), ((y - my) ** 2).mean()
    cov = ((x - mx) * (y - my)).mean()
    den = vx + vy + (mx - my) ** 2
    return float(2 * cov / den) if float(den) != 0 else 0.0


# --------------------------------------------------------------------------- train step
def to_params(state: Dict[str, "torch.Tensor"], dtype=torch.float32, requires_grad=False):
    P = {}
    for k, v in state.items():
        t = torch.as_tensor(v).detach().to(dtype).clone()
        t.requires_grad_(requires_grad)
        P[k] = t
    return P


# --------------------------------------------------------------------------- bf16 storage emulation
# The bf16 compute path of the HIP kernels keeps fp32 accumulators and fp32 parameters / vectors but STORES packed weight
# matrices, activations and activation-gradients as bf16.  The functions below restate the same arithmetic (the
# reference's, fusion.py / deer.py as cited above) with a round-to-nearest-even to bf16 at exactly those storage points,
# so a bf16 GPU step can be compared with the oracle at a tolerance set by fp32 summation order instead of by bf16's 8
# significant bits.  Rounding points (csrc file: what is stored):
#   * gemm_kernel.inc epilogue_direct: C = bf16(dropout(relu(acc + bias))) forward; dX = bf16(acc * (Y > 0) * 1/(1-p)) or
#     bf16(acc * regenerated keep factor) backward, i.e. the gradient is rounded AFTER it passed the ReLU / dropout mask of
#     the layer below;
#   * rowops.hip ln_fwd / ln_bwd: LayerNorm output bf16; its input gradient (masked like a dX) bf16;
#   * tri_fused.hip: q, k rounded to bf16 for the score products only (the backward multiplies by the unrounded
#     accumulators); v stays fp32; the token-pooled context obar and dqkv are stored bf16; probabilities fp32;
#   * nig.hip: the evidence (64 -> 4 layer output) and everything after it fp32; dz2 bf16;
#   * api.hip pack: every weight MATRIX bf16 (gradients of the fp32 master parameters are NOT rounded: the weight-gradient
#     GEMMs accumulate bf16 operands in fp32 slabs); biases and LayerNorm vectors fp32.
def _bf16_round(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundFwd(torch.autograd.Function):
    """value rounded to bf16, gradient passed through (a stored activation / a packed weight)."""
    @staticmethod
    def forward(ctx, x):
        return _bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """value passed through, gradient rounded to bf16 (a stored activation-gradient)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16_round(g)


class _ScoresBf16(torch.autograd.Function):
    """q k^T on bf16-rounded q, k (tri_fused.hip MODE 0); the backward multiplies by the UNROUNDED q, k (MODE 1
    recomputes the fp32 accumulators)."""
    @staticmethod
    def forward(ctx, q, k):
        ctx.save_for_backward(q, k)
        return _bf16_round(q) @ _bf16_round(k).transpose(-1, -2)

    @staticmethod
    def backward(ctx, g):
        q, k = ctx.saved_tensors
        return g @ k, g.transpose(-1, -2) @ q


def _qf(x):
    return _RoundFwd.apply(x)


def _qb(x):
    return _RoundBwd.apply(x)


def _qfb(x):
    return _RoundBwd.apply(_RoundFwd.apply(x))


def drop_scale(p):
    """1 / (1 - p) as api.hip:make_drop forms it: p arrives as a C float, the quotient is rounded to float."""
    import numpy as np
    keep = 1.0 - float(np.float32(p))
    return float(np.float32(1.0 / keep)) if keep > 0 else 0.0


def _drop_k(x, mask, p):
    """keep ? x * scale : 0 -- the kernels' form of inverted dropout (multiplication by the fp32 reciprocal)."""
    if mask is None:
        return x
    return x * (mask.to(x.dtype) * drop_scale(p))


def model_forward_bf16(P, audio, video, text, masks=None, p=0.3, heads=8):
    """model_forward with the bf16 storage points of the HIP path (see the block comment above); fp32 tensors in, fp32 out.
    Follows the launch plan of csrc/api.hip:mmdeer_forward (F1 .. F18), e.g. the token mean BEFORE the attention out_proj."""
    masks = masks or {}
    W = lambda name: _qf(P[name])
    lin = lambda x, pre: x @ W(pre + ".weight").t() + P[pre + ".bias"]

    def relu_drop_ln(x, pre, mask):          # Linear -> ReLU -> Dropout -> LayerNorm block (F4-5, F10-11, F12-13)
        y = _qb(lin(x, pre + ".0"))          # d(pre-activation) is what ln_bwd stores
        z = _qf(_drop_k(torch.relu(y), mask, p))
        return _qfb(_layer_norm(z, P[pre + ".3.weight"], P[pre + ".3.bias"]))

    a, v, t = _qf(audio), _qf(video), _qf(text)
    pre = "fusion.audio_visual_fusion."
    ap = _qfb(lin(a, pre + "audio_projection"))                      # F1
    vp = _qfb(lin(v, pre + "video_projection"))
    E = ap.shape[1]
    hd = E // heads
    w_v = W(pre + "cross_attention.in_proj_weight")[2 * E:]
    b_v = P[pre + "cross_attention.in_proj_bias"][2 * E:]

    def attend(kv, mask):                                            # F2, F3
        B = kv.shape[0]
        val = _qb(kv @ w_v.t() + b_v)
        wgt = _drop_k(torch.ones(B, heads, dtype=val.dtype), mask, p)
        val = _qf((val.view(B, heads, hd) * wgt[:, :, None]).reshape(B, E))
        return _qfb(lin(val, pre + "cross_attention.out_proj")), wgt.mean(dim=1, keepdim=True)

    audio_att, w_a2v = attend(vp, masks.get("av_attn_a2v"))
    video_att, w_v2a = attend(ap, masks.get("av_attn_v2a"))
    av = relu_drop_ln(torch.cat([audio_att, video_att], dim=-1), pre + "fusion_layers", masks.get("av_fuse"))

    pre = "fusion.trimodal_fusion."
    x0 = _qfb(lin(av, pre + "audiovisual_projection"))               # F6
    x1 = _qfb(lin(t, pre + "text_projection"))                       # F1
    x = torch.stack([x0, x1], dim=1)
    B, T, E = x.shape
    hd = E // heads
    qkv = _qb(x @ W(pre + "modality_attention.in_proj_weight").t() + P[pre + "modality_attention.in_proj_bias"])   # F7
    q, k, vv = (u.view(B, T, heads, hd).transpose(1, 2) for u in qkv.split(E, dim=-1))
    prob = torch.softmax(_ScoresBf16.apply(q, k) * math.sqrt(1.0 / hd), dim=-1)       # F8 (B, H, 2, 2)
    probd = _drop_k(prob, masks.get("tri_attn"), p)
    o = (probd @ vv).transpose(1, 2).reshape(B, T, E)
    obar = _qfb(o.mean(dim=1))                                       # token mean commutes with out_proj: B rows
    pooled = _qfb(lin(obar, pre + "modality_attention.out_proj"))    # F9
    tri = relu_drop_ln(pooled, pre + "final_fusion", masks.get("tri_fuse"))
    fused = relu_drop_ln(tri, "fusion.output_projection", masks.get("out_proj"))
    fo = {"fused_features": fused, "audiovisual_features": av, "trimodal_features": tri,
          "av_attention_weights": {"audio_to_video": w_a2v, "video_to_audio": w_v2a},
          "trimodal_attention_weights": probd.mean(dim=1), "trimodal_probs": prob, "uncertainty_weights": None}

    layer = lambda x, pre, mask: _qf(_drop_k(torch.relu(_qb(lin(x, pre))), mask, p))
    h = layer(fused, "head.feature_processor.0", masks.get("fp0"))   # F14
    h = layer(h, "head.feature_processor.3", masks.get("fp1"))       # F15
    ho: Dict[str, torch.Tensor] = {}
    keys = ("mu", "nu", "alpha", "beta", "aleatoric_uncertainty", "epistemic_uncertainty", "uncertainty")
    m0, m1 = masks.get("ev0"), masks.get("ev1")
    for i, dim in enumerate(DIM_NAMES):
        pre = f"head.deer_heads.{i}.evidence_net"
        e = layer(h, pre + ".0", None if m0 is None else m0[:, i])   # F16
        e = layer(e, pre + ".3", None if m1 is None else m1[:, i])   # F17
        e = lin(e, pre + ".6").view(B, 1, 4)                         # F18: evidence stays fp32
        for key, val in zip(keys, nig_activations(e)):
            ho[f"{dim}_{key}"] = val
    ho["mu_all"] = torch.cat([ho[f"{d}_mu"] for d in DIM_NAMES], dim=1)
    ho["uncertainty_all"] = torch.cat([ho[f"{d}_uncertainty"] for d in DIM_NAMES], dim=1)
    return fo, ho


# --------------------------------------------------------------------------- kernel-level restatements (bf16 path)
# One function per launch of csrc/api.hip's plan, each taking the tensors THAT launch reads (as fp32 tensors holding
# bf16-representable values where the kernel reads bf16) and returning what it stores.  tests/test_gpu_bf16_layers.py feeds
# every launch of a full-size step its own stored inputs ("teacher forcing"), so the chaotic growth of rounding differences
# through the chain (see model_forward_bf16's callers) cannot hide an error inside one kernel.
def k_linear(x, W, b, relu=False, mask=None, p=0.0):
    """gemm epilogue_direct, forward: bf16(dropout(relu(x W^T + b))); W is rounded as the pack kernels do."""
    y = x @ _bf16_round(W).t() + b
    if relu:
        y = torch.relu(y)
    return _bf16_round(_drop_k(y, mask, p))


def k_dx(dy, W, ymask=None, p=0.0, keep=None):
    """gemm epilogue_direct, backward: bf16((dy W) * (Y > 0) / (1 - p)) -- Y the stored output of the layer below -- or
    bf16((dy W) * regenerated keep factor) for the attention-weight dropout."""
    g = dy @ _bf16_round(W)
    if ymask is not None:
        g = g * ((ymask > 0).to(g.dtype) * (drop_scale(p) if p > 0 else 1.0))
    if keep is not None:
        g = g * (keep.to(g.dtype) * drop_scale(p))
    return _bf16_round(g)


def k_dw(dy, x):
    """weight-gradient GEMM + slab fold: dW = dy^T x, db = column sums of dy (fp32 accumulation of bf16 operands;
    accumulated here in fp64 so the reference does not depend on a summation order)."""
    return (dy.double().t() @ x.double()).float(), dy.double().sum(dim=0).float()


def k_ln_fwd(y, gamma, beta, eps=1e-5):
    """rowops.hip ln_fwd_kernel: (bf16 out, mean, rstd)."""
    mu = y.mean(dim=-1, keepdim=True)
    var = ((y - mu) ** 2).mean(dim=-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    return _bf16_round((y - mu) * rstd * gamma + beta), mu.squeeze(-1), rstd.squeeze(-1)


def k_ln_bwd(dout, y, mean, rstd, gamma, p=0.0):
    """rowops.hip ln_bwd_kernel: gradient of LayerNorm(y) w.r.t. the PRE-activation of the Linear-ReLU-Dropout in front of it
    (y = stored dropout(relu(.)): mask (y > 0) / (1 - p)), bf16; plus d gamma, d beta."""
    xhat = (y - mean[:, None]) * rstd[:, None]
    dxhat = dout * gamma
    dz = rstd[:, None] * (dxhat - dxhat.mean(dim=-1, keepdim=True) - xhat * (dxhat * xhat).mean(dim=-1, keepdim=True))
    dz = dz * ((y > 0).to(dz.dtype) * (drop_scale(p) if p > 0 else 1.0))
    return _bf16_round(dz), (dout.double() * xhat.double()).sum(dim=0).float(), dout.double().sum(dim=0).float()


def _tri_split(xtok, w_in, b_in, heads):
    qkv = xtok @ _bf16_round(w_in).t() + b_in
    B, T, E3 = qkv.shape
    E = E3 // 3
    return [u.view(B, T, heads, E // heads).transpose(1, 2) for u in qkv.split(E, dim=-1)]    # q, k, v: (B, H, T, hd)


def k_tri_fwd(xtok, w_in, b_in, mask=None, p=0.0, heads=8, probs=None):
    """tri_fused_kernel<0>: xtok (B, 2, 512) -> (softmax probabilities (B, H, 2, 2) fp32, bf16 token-pooled context (B, 512)).
    The scores are products of bf16-ROUNDED q, k (an internal rounding: an element of q or k that rounds the other way moves
    a probability by up to ~5e-4); with `probs` given, the context is formed from THOSE probabilities, so the second half of
    the kernel can be checked on its own."""
    q, k, v = _tri_split(xtok, w_in, b_in, heads)
    B, H, T, hd = q.shape
    prob = torch.softmax((_bf16_round(q) @ _bf16_round(k).transpose(-1, -2)) * math.sqrt(1.0 / hd), dim=-1)
    o = (_drop_k(prob if probs is None else probs, mask, p) @ v).transpose(1, 2).reshape(B, T, H * hd)
    return prob, _bf16_round(o.mean(dim=1))


def k_tri_bwd(xtok, w_in, b_in, dobar, probs, mask=None, p=0.0, heads=8):
    """tri_fused_kernel<1>: recompute q|k|v (fp32, unrounded), attention backward from the SAVED probabilities -> bf16 dqkv
    (B, 2, 1536) in the reference's column order [q | k | v]."""
    q, k, v = _tri_split(xtok, w_in, b_in, heads)
    B, H, T, hd = q.shape
    keep = 1.0 if mask is None else mask.to(q.dtype) * drop_scale(p)
    do = (0.5 * dobar).view(B, 1, H, hd).transpose(1, 2).expand(B, H, T, hd)        # d o_t = d obar / 2 for both tokens
    pd = probs * keep
    dv = pd.transpose(-1, -2) @ do
    dp = (do @ v.transpose(-1, -2)) * keep
    ds = probs * (dp - (probs * dp).sum(dim=-1, keepdim=True)) * math.sqrt(1.0 / hd)
    dq, dk = ds @ k, ds.transpose(-1, -2) @ q
    back = lambda u: u.transpose(1, 2).reshape(B, T, H * hd)
    return _bf16_round(torch.cat([back(dq), back(dk), back(dv)], dim=-1))


def k_nig_bwd(evid, targets, e2, w3, p=0.0):
    """nig_bwd_kernel in loss mode: d MultiTaskDEERLoss / d evidence (fp32, from the stored evidence (B, 3, 4)) and
    dz2 = bf16((d evid . W3) * (e2 > 0) / (1 - p)); w3: list of the three (4, 64) last-layer weights."""
    ev = evid.detach().clone().requires_grad_(True)
    pred = {}
    for i, dim in enumerate(DIM_NAMES):
        mu, nu, alpha, beta = nig_activations(ev[:, i:i + 1, :])[:4]
        pred.update({f"{dim}_mu": mu, f"{dim}_nu": nu, f"{dim}_alpha": alpha, f"{dim}_beta": beta})
    multitask_loss(pred, targets)["total_loss"].backward()
    dev = ev.grad
    dz = torch.cat([dev[:, i, :] @ _bf16_round(w3[i]) for i in range(3)], dim=1)
    dz = dz * ((e2 > 0).to(dz.dtype) * (drop_scale(p) if p > 0 else 1.0))
    return dev, _bf16_round(dz)


def train_step(P, audio, video, text, targets, masks=None, p=0.3, heads=8, emulate_bf16=False):
    """forward + MultiTaskDEERLoss + backward over every parameter (the bench 'step').
    Returns (fusion_out, head_out, loss_dict, grads keyed like P).  emulate_bf16: round to bf16 where the bf16 HIP path
    stores (model_forward_bf16); P, the inputs and every result stay fp32 tensors."""
    for t in P.values():
        t.grad = None
    if emulate_bf16:
        fo, ho = model_forward_bf16(P, audio, video, text, masks, p, heads)
    else:
        fo, ho = model_forward(P, audio, video, text, masks, p, heads)
    ld = multitask_loss(ho, targets)
    ld["total_loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
    return fo, ho, ld, grads
