"""Stack B: ``CompleteDEERModel`` of the reference's src/models/complete_project.py, inference forward.

Same constructor protocol (``ModelConfig``), ``state_dict()`` keys and shapes, output dictionary and
``get_predictions_and_uncertainties`` as the reference class (complete_project.py:33-56, 462-602), so a reference
checkpoint loads with ``load_state_dict`` and callers of ``model(audio, video, text)`` keep working.  The arithmetic
runs on the HIP library only -- 33 ``mmdeer_gemm`` launches and the four ``mmdeer_stackb_*`` row kernels per batch;
there is no CPU path.

How the batch is laid out (one allocation per intermediate, all row-major in HBM):
  * the three encoders write their (B, 256) outputs into column blocks of one (B, 768) matrix, which read as
    (3B, 256) is the (sample, modality)-interleaved row set that the shared-weight layers -- both attention blocks and
    the uncertainty estimator -- consume in a single GEMM each;
  * with one key per query the reference's MultiHeadAttention softmax is identically 1 (complete_project.py:160-171), so
    self / cross attention are ``output_proj(value_proj(.))``; the two value projections share one N = 512 GEMM, and the
    (3B, 256) self-attention output read as (B, 768) is already ``cat([audio_self, video_self, text_self])``;
  * ``weight_network.0`` has K = 771: its 768 feature columns run as a GEMM, the three uncertainty columns are added
    in ``mmdeer_stackb_attn_mix`` together with the ReLU, the 256 -> 3 layer, the softmax and the final mix;
  * the first layer of the three prediction heads is one N = 768 GEMM.

Training (dropout, backward) is not built for this stack: the reference trains Stack C, which is the path
``mmdeer.model.MultimodalDEER`` accelerates end to end.  ``forward`` raises in training mode instead of silently
skipping dropout.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops

DIM_NAMES = ("valence", "arousal", "dominance")
HEAD_KEYS = ("mu", "nu", "alpha", "beta", "aleatoric_uncertainty", "epistemic_uncertainty", "uncertainty")


@dataclass
class ModelConfig:
    """Field names and defaults of complete_project.ModelConfig (complete_project.py:33-56)."""
    audio_dim: int = 84
    video_dim: int = 256
    text_dim: int = 768
    encoder_dim: int = 256
    fusion_dim: int = 512
    emotion_dims: int = 3
    attention_heads: int = 8
    encoder_layers: int = 3
    dropout: float = 0.3
    evidence_weight: float = 1.0
    kl_weight: float = 0.1
    learning_rate: float = 1e-4
    weight_decay: float = 1e-5
    gradient_clip: float = 1.0


# ---- parameter containers: the module tree (and therefore every state_dict key) of the reference classes.  The
#      activation / dropout slots hold no parameters; they only keep the Sequential indices aligned.
def _slot() -> nn.Module:
    return nn.Identity()


def _residual_block(dim: int) -> nn.Module:
    m = nn.Module()
    m.layers = nn.Sequential(nn.Linear(dim, dim), _slot(), _slot(), nn.LayerNorm(dim))
    return m


def _encoder(in_dim: int, dim: int, layers: int) -> nn.Module:
    m = nn.Module()
    m.input_projection = nn.Sequential(nn.Linear(in_dim, dim), _slot(), nn.LayerNorm(dim))
    m.encoder_layers = nn.ModuleList([_residual_block(dim) for _ in range(layers)])
    m.output_projection = nn.Linear(dim, dim)
    return m


def _mha(dim: int) -> nn.Module:
    m = nn.Module()
    for n in ("query_proj", "key_proj", "value_proj", "output_proj"):
        setattr(m, n, nn.Linear(dim, dim))
    return m


def _attention(dim: int) -> nn.Module:
    m = nn.Module()
    m.self_attention, m.cross_attention = _mha(dim), _mha(dim)
    m.uncertainty_estimator = nn.Module()
    m.uncertainty_estimator.estimator = nn.Sequential(nn.Linear(dim, dim // 2), _slot(), _slot(), nn.Linear(dim // 2, dim // 4),
                                                      _slot(), nn.Linear(dim // 4, 1), _slot())
    m.weight_network = nn.Sequential(nn.Linear(dim * 3 + 3, dim), _slot(), _slot(), nn.Linear(dim, 3), _slot())
    return m


def _fusion(dim: int, fdim: int) -> nn.Module:
    m = nn.Module()
    stage = lambda k: nn.Sequential(nn.Linear(k, fdim), _slot(), _slot(), nn.LayerNorm(fdim), nn.Linear(fdim, fdim), _slot())
    m.av_fusion = stage(2 * dim)
    m.trimodal_fusion = stage(fdim + dim)
    m.fusion_gate = nn.Sequential(nn.Linear(fdim + dim, fdim), _slot())
    return m


def _head(fdim: int, hidden: int = 256) -> nn.Module:
    m = nn.Module()
    m.evidence_network = nn.Sequential(nn.Linear(fdim, hidden), _slot(), _slot(), nn.Linear(hidden, hidden // 2), _slot(), _slot(),
                                       nn.Linear(hidden // 2, 4))
    return m


class _Calibration(nn.Module):
    def __init__(self, dims: int):
        super().__init__()
        self.temperature = nn.Parameter(torch.ones(dims))
        self.calibration_network = nn.Sequential(nn.Linear(1, 32), _slot(), nn.Linear(32, 16), _slot(), nn.Linear(16, 1), _slot())


class CompleteDEERModel(nn.Module):
    """Mirror of ``complete_project.CompleteDEERModel`` (complete_project.py:462-602); see the module docstring."""

    def __init__(self, config: Optional[ModelConfig] = None, compute_dtype: str = "fp32"):
        super().__init__()
        config = config or ModelConfig()
        if (config.encoder_dim, config.fusion_dim, config.emotion_dims) != (256, 512, 3):
            raise NotImplementedError("the HIP row kernels are specialised for encoder_dim=256, fusion_dim=512, emotion_dims=3")
        if config.encoder_dim % config.attention_heads:
            raise ValueError("feature_dim must be divisible by num_heads")          # complete_project.py:126
        for k in (config.audio_dim, config.video_dim, config.text_dim):
            if k % 4:
                raise NotImplementedError("input dimensions must be multiples of 4")
        ops._act_dtype(compute_dtype)
        self.config, self.compute_dtype = config, compute_dtype
        d, f = config.encoder_dim, config.fusion_dim
        self.audio_encoder = _encoder(config.audio_dim, d, config.encoder_layers)
        self.video_encoder = _encoder(config.video_dim, d, config.encoder_layers)
        self.text_encoder = _encoder(config.text_dim, d, config.encoder_layers)
        self.attention_module = _attention(d)
        self.fusion_module = _fusion(d, f)
        self.prediction_heads = nn.ModuleDict({n: _head(f) for n in DIM_NAMES})
        self.calibration_layer = _Calibration(config.emotion_dims)
        self._initialize_weights()
        self._packed = None

    def _initialize_weights(self) -> None:
        """Xavier-uniform Linear weights, zero biases, unit LayerNorm (complete_project.py:503-513)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    # ---- device-side operand images (compute dtype, fused where layers share an input), rebuilt when a parameter changes
    def _pack(self) -> dict:
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self.compute_dtype,)
        if self._packed is not None and self._packed["key"] == key:
            return self._packed
        dt = ops._act_dtype(self.compute_dtype)
        w = lambda lin: lin.weight.detach().to(dt).contiguous()
        b = lambda lin: lin.bias.detach().float().contiguous()
        f32 = lambda t: t.detach().float().contiguous()
        P = {"key": key, "enc": []}
        for enc in (self.audio_encoder, self.video_encoder, self.text_encoder):
            ip = enc.input_projection
            P["enc"].append({
                "in": (w(ip[0]), b(ip[0]), f32(ip[2].weight), f32(ip[2].bias)),
                "res": [(w(r.layers[0]), b(r.layers[0]), f32(r.layers[3].weight), f32(r.layers[3].bias)) for r in enc.encoder_layers],
                "out": (w(enc.output_projection), b(enc.output_projection))})
        att = self.attention_module
        sa, ca = att.self_attention, att.cross_attention
        P["value"] = (torch.cat([w(sa.value_proj), w(ca.value_proj)], 0).contiguous(), torch.cat([b(sa.value_proj), b(ca.value_proj)]))
        P["self_out"], P["cross_out"] = (w(sa.output_proj), b(sa.output_proj)), (w(ca.output_proj), b(ca.output_proj))
        est = att.uncertainty_estimator.estimator
        P["est"] = ((w(est[0]), b(est[0])), (w(est[3]), b(est[3])), f32(est[5].weight).view(-1), f32(est[5].bias))
        wn = att.weight_network
        D3 = 3 * self.config.encoder_dim
        P["wn"] = (wn[0].weight.detach()[:, :D3].to(dt).contiguous(), b(wn[0]), wn[0].weight.detach()[:, D3:].float().contiguous(),
                   f32(wn[3].weight), f32(wn[3].bias))
        fu = self.fusion_module
        for name, seq in (("av", fu.av_fusion), ("tri", fu.trimodal_fusion)):
            P[name] = ((w(seq[0]), b(seq[0])), (f32(seq[3].weight), f32(seq[3].bias)), (w(seq[4]), b(seq[4])))
        P["gate"] = (w(fu.fusion_gate[0]), b(fu.fusion_gate[0]))
        nets = [self.prediction_heads[n].evidence_network for n in DIM_NAMES]
        P["head0"] = (torch.cat([w(n[0]) for n in nets], 0).contiguous(), torch.cat([b(n[0]) for n in nets]))
        P["head3"] = [(w(n[3]), b(n[3])) for n in nets]
        P["head6"] = [(w(n[6]), b(n[6])) for n in nets]
        cal = self.calibration_layer
        cn = cal.calibration_network
        P["cal"] = (f32(cal.temperature), f32(cn[0].weight).view(-1), f32(cn[0].bias), f32(cn[2].weight), f32(cn[2].bias),
                    f32(cn[4].weight).view(-1), f32(cn[4].bias))
        self._packed = P
        return P

    @torch.no_grad()
    def forward(self, audio_features: torch.Tensor, video_features: torch.Tensor, text_features: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self.training:
            raise NotImplementedError("mmdeer.stackb.CompleteDEERModel is inference-only: call .eval() first "
                                      "(training runs on mmdeer.model.MultimodalDEER, the stack the reference trains)")
        cfg = self.config
        xs = (audio_features, video_features, text_features)
        ops._check_dev(*xs)
        B = xs[0].shape[0]
        for x, k in zip(xs, (cfg.audio_dim, cfg.video_dim, cfg.text_dim)):
            if x.dim() != 2 or x.shape != (B, k):
                raise ValueError(f"expected features of shape ({B}, {k}), got {tuple(x.shape)}")
        c, dt = self.compute_dtype, ops._act_dtype(self.compute_dtype)
        dev = xs[0].device
        P = self._pack()
        lib = _lib.load()
        stream = _lib.current_stream()
        new = lambda *s, dtype=dt: torch.empty(*s, dtype=dtype, device=dev)
        D, Fd = cfg.encoder_dim, cfg.fusion_dim

        # -- encoders (complete_project.py:76-117) -> column blocks of E
        E = new(B, 3 * D)
        tmp, h = new(B, D), new(B, D)
        for m, (x, pe) in enumerate(zip(xs, P["enc"])):
            wi, bi, g, be = pe["in"]
            ops.linear_into(x.detach().to(dt).contiguous(), wi, bi, tmp, relu=True, compute=c)
            ops.residual_layer_norm(tmp, None, g, be, h)
            for wr, br, g, be in pe["res"]:
                ops.linear_into(h, wr, br, tmp, relu=True, compute=c)
                ops.residual_layer_norm(tmp, h, g, be, h)
            ops.linear_into(h, pe["out"][0], pe["out"][1], E[:, m * D:(m + 1) * D], compute=c)
        E3 = E.view(3 * B, D)

        # -- UncertaintyAwareAttention (complete_project.py:216-304) on the interleaved rows
        VV = ops.linear_into(E3, *P["value"], new(3 * B, 2 * D), compute=c)
        S = ops.linear_into(VV[:, :D], *P["self_out"], new(3 * B, D), compute=c)
        X = ops.linear_into(VV[:, D:], *P["cross_out"], new(3 * B, D), compute=c)
        (w1, b1), (w2, b2), w3, b3 = P["est"]
        H2 = ops.linear_into(ops.linear_into(E3, w1, b1, new(3 * B, D // 2), relu=True, compute=c), w2, b2, new(3 * B, D // 4),
                             relu=True, compute=c)
        wn_f, wn_b, wn_u, wn_w2, wn_b2 = P["wn"]
        pre = ops.linear_into(S.view(B, 3 * D), wn_f, wn_b, new(B, D), compute=c)
        AV, T = new(B, 2 * D), new(B, Fd + D)
        weights, unc = new(B, 3, dtype=torch.float32), new(B, 3, dtype=torch.float32)
        a = _lib.StackBAttnArgs()
        a.h2, a.pre, a.self_out, a.cross_out = H2.data_ptr(), pre.data_ptr(), S.data_ptr(), X.data_ptr()
        a.est_w3, a.est_b3, a.wn_w1_unc, a.wn_w2, a.wn_b2 = w3.data_ptr(), b3.data_ptr(), wn_u.data_ptr(), wn_w2.data_ptr(), wn_b2.data_ptr()
        a.out_av, a.out_text = AV.data_ptr(), T[:, Fd:].data_ptr()
        a.weights, a.uncertainties = weights.data_ptr(), unc.data_ptr()
        a.ld_w1_unc, a.ld_av, a.ld_text, a.B, a.act_f32 = 3, AV.stride(0), T.stride(0), B, int(dt == torch.float32)
        a.stream = stream
        if B:
            _lib.check(lib.mmdeer_stackb_attn_mix(C.byref(a)))

        # -- HierarchicalFusionModule (complete_project.py:307-366); av_fused lands in T[:, :512] next to the text block
        def stage(x, p, out):
            (w0, b0), (g, be), (w4, b4) = p
            y = ops.linear_into(x, w0, b0, new(B, Fd), relu=True, compute=c)
            ops.residual_layer_norm(y, None, g, be, y)
            return ops.linear_into(y, w4, b4, out, relu=True, compute=c)
        stage(AV, P["av"], T[:, :Fd])
        G = ops.linear_into(T, *P["gate"], new(B, Fd), compute=c)
        R = stage(T, P["tri"], new(B, Fd))
        fused = new(B, Fd)
        _lib.check(lib.mmdeer_stackb_gate_mix(G.data_ptr(), G.stride(0), R.data_ptr(), R.stride(0), T.data_ptr(), T.stride(0),
                                              fused.data_ptr(), fused.stride(0), B, Fd, int(dt == torch.float32), stream))

        # -- prediction heads + calibration (complete_project.py:369-459)
        H0 = ops.linear_into(fused, *P["head0"], new(B, 3 * 256), relu=True, compute=c)
        H3, ev = new(B, 3 * 128), new(B, 12, dtype=torch.float32)
        for d in range(3):
            ops.linear_into(H0[:, d * 256:(d + 1) * 256], *P["head3"][d], H3[:, d * 128:(d + 1) * 128], relu=True, compute=c)
            ops.linear_into(H3[:, d * 128:(d + 1) * 128], *P["head6"][d], ev[:, 4 * d:4 * d + 4], compute=c)
        planes = new(8, B, 3, dtype=torch.float32)
        _lib.check(lib.mmdeer_stackb_head(ev.data_ptr(), ev.stride(0), *(t.data_ptr() for t in P["cal"]), planes.data_ptr(), B, stream))

        out: Dict[str, torch.Tensor] = {}
        for d, name in enumerate(DIM_NAMES):
            for k, key in enumerate(HEAD_KEYS):
                out[f"{name}_{key}"] = planes[k, :, d]
        out["mu_all"], out["uncertainty_all"], out["calibrated_uncertainty"] = planes[0], planes[6], planes[7]
        out["attention_weights"], out["modality_uncertainties"] = weights, unc
        out["fused_features"] = fused.float() if dt != torch.float32 else fused
        return out

    def get_predictions_and_uncertainties(self, outputs: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """(mu_all, calibrated_uncertainty or uncertainty_all) -- complete_project.py:591-602."""
        return outputs["mu_all"], outputs.get("calibrated_uncertainty", outputs["uncertainty_all"])


def create_complete_deer_model(config: Optional[ModelConfig] = None, compute_dtype: str = "fp32") -> CompleteDEERModel:
    """Factory of complete_project.py:605-632 (without its parameter-count printout)."""
    return CompleteDEERModel(config or ModelConfig(), compute_dtype=compute_dtype)
