"""``fusion.HierarchicalMultimodalFusion`` (src/models/fusion.py:35-185) on geometries OTHER than the reference's default
(audio 84 / video 256 / text 768): any ``audio_dim`` / ``video_dim`` / ``text_dim`` that is a multiple of 4, with
``fusion_dim = 512``, ``intermediate_dim = 256`` and 8 heads (the sizes the attention operators are built for).

The default geometry runs as the fused launch plan of ``mmdeer_forward`` / ``mmdeer_backward`` (model.py); this module is the
operator path: every layer is a C-ABI call on the current stream (``mmdeer_gemm`` with bias / ReLU / counter-hash dropout in its
epilogue, ``mmdeer_layernorm_fwd / _bwd``, ``mmdeer_trimodal_attn_fwd / _bwd``), wrapped in ``torch.autograd.Function``s so that
it trains through ``loss.backward()``.  Correct, not tuned: ~25 launches forward.  The parameters live in torch's own module
classes under the reference's attribute names, so ``state_dict()`` has exactly the reference's keys and shapes for the given
dimensions (``uncertainty_gate.*`` included: kept for checkpoints, unreachable in the reference).

  AudioVisualFusion      fusion.py:196-271   audio / video projections, the shared 1-key cross-attention = out_proj(v_proj(.)) with
                                             one dropout decision per (row, head), Linear-ReLU-Dropout-LayerNorm
  TrimodalFusion         fusion.py:281-343   two token projections, packed in_proj, 2-token x 8-head attention, out_proj on the
                                             token-pooled context (the mean over tokens commutes with it), Linear-ReLU-Dropout-LayerNorm
  output_projection      fusion.py:98-103    Linear-ReLU-Dropout-LayerNorm
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch
from torch import nn

from . import _lib
from .fusions import _Drop, _LinActLnFn, _act, _convert, _dt, linear
from .opseq import Exec

_SITE_AV, _SITE_AVF, _SITE_TRI_ATTN, _SITE_TRIF, _SITE_OUT = 81, 82, 83, 84, 85


class _AVAttnFn(torch.autograd.Function):
    """The shared ``nn.MultiheadAttention`` of AudioVisualFusion called with one query and one key (fusion.py:244-255): softmax
    over a single key is 1, so the call is ``out_proj(dropout_heads(v_proj(key)))``; the q / k thirds of ``in_proj_weight``
    receive exact zeros as gradient, as in the reference.  Both calls run stacked: rows [0, B) = key video, [B, 2B) = key audio."""

    @staticmethod
    def forward(ctx, kv, w_in, b_in, w_out, b_out, compute_dtype, heads, drop, site):
        dt = _dt(compute_dtype)
        x = _act(kv, dt)
        E, M = x.shape[1], x.shape[0]
        wv = w_in.detach()[2 * E:].to(dt).contiguous()
        bv = b_in.detach()[2 * E:].float().contiguous()
        wo, bo = w_out.detach().to(dt).contiguous(), b_out.detach().float().contiguous()
        ex = Exec(compute_dtype, drop)
        p = ex.p_of(drop[0]) if drop else 0.0
        shift = (E // heads).bit_length() - 1
        v = torch.empty(M, E, dtype=dt, device=x.device)
        ex.gemm(x, wv, v, M, E, E, E, E, E, bias=bv, drop_site=site if p > 0 else -1, drop_shift=shift, p=p)
        out = torch.empty(M, E, dtype=dt, device=x.device)
        ex.gemm(v, wo, out, M, E, E, E, E, E, bias=bo)
        ctx.save_for_backward(x, wv, v, wo)
        ctx.meta = (compute_dtype, drop, site, shift, p, kv.dtype, w_in.dtype, E)
        return out.float()

    @staticmethod
    def backward(ctx, g):
        x, wv, v, wo = ctx.saved_tensors
        compute_dtype, drop, site, shift, p, xdt, wdt, E = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype, drop)           # the same (seed, step): the keep factors of the forward are regenerated
        M = x.shape[0]
        ga = _convert(ex, g, dt)
        dv = torch.empty(M, E, dtype=dt, device=x.device)
        ex.dx(ga, E, wo, dv, E, M, regen_site=site, shift=shift, p=p)        # d v = (g W_o) * keep / (1 - p)
        gwo, gbo = torch.zeros(E, E, device=x.device), torch.zeros(E, device=x.device)
        ex.dw(ga, E, v, E, gwo, gbo, M, E, E)
        dkv = torch.empty(M, E, dtype=dt, device=x.device)
        ex.dx(dv, E, wv, dkv, E, M)
        gw_in, gb_in = torch.zeros(3 * E, E, device=x.device), torch.zeros(3 * E, device=x.device)
        ex.dw(dv, E, x, E, gw_in[2 * E:], gb_in[2 * E:], M, E, E)
        return dkv.to(xdt), gw_in.to(wdt), gb_in.to(wdt), gwo.to(wdt), gbo.to(wdt), None, None, None, None


class _TriAttnFn(torch.autograd.Function):
    """2-token x 8-head self-attention on the packed q|k|v rows (row 2b + t), token-pooled context out (fusion.py:328-335)."""

    @staticmethod
    def forward(ctx, qkv, compute_dtype, drop, want_weights):
        dt = _dt(compute_dtype)
        x = _act(qkv, dt)
        B = x.shape[0] // 2
        lib, s = _lib.load(), _lib.current_stream()
        obar = torch.empty(B, 512, dtype=dt, device=x.device)
        probs = torch.empty(B, 8, 4, dtype=torch.float32, device=x.device)
        attn_w = torch.empty(B, 2, 2, dtype=torch.float32, device=x.device) if want_weights else None
        p, seed, step = (drop[0], drop[1], drop[2]) if drop else (0.0, 0, 0)
        _lib.check(lib.mmdeer_trimodal_attn_fwd(x.data_ptr(), obar.data_ptr(), probs.data_ptr(), attn_w.data_ptr() if want_weights else None,
                                                None, B, int(dt == torch.float32), int(bool(drop)), p, seed, step, s))
        ctx.save_for_backward(x, probs)
        ctx.meta = (compute_dtype, drop, qkv.dtype)
        ctx.mark_non_differentiable(*( [attn_w] if want_weights else []))
        return (obar.float(), attn_w) if want_weights else (obar.float(), None)

    @staticmethod
    def backward(ctx, g, _gw):
        x, probs = ctx.saved_tensors
        compute_dtype, drop, xdt = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype)
        B = x.shape[0] // 2
        ga = _convert(ex, g, dt)
        dqkv = torch.empty_like(x)
        p, seed, step = (drop[0], drop[1], drop[2]) if drop else (0.0, 0, 0)
        _lib.check(ex.lib.mmdeer_trimodal_attn_bwd(x.data_ptr(), ga.data_ptr(), probs.data_ptr(), dqkv.data_ptr(), B, int(dt == torch.float32),
                                                   int(bool(drop)), p, seed, step, ex.s))
        return dqkv.to(xdt), None, None, None


class _Holder(nn.Module):
    """A parameter container with named children (the reference's AudioVisualFusion / TrimodalFusion attribute names)."""


def _lin_relu_drop_ln(d_in: int, d_out: int, p: float) -> nn.Sequential:
    return nn.Sequential(nn.Linear(d_in, d_out), nn.ReLU(), nn.Dropout(p), nn.LayerNorm(d_out))


class GenericHierarchicalFusion(nn.Module):
    """See the module docstring.  ``forward(audio, video, text)`` returns the reference's output dictionary (fusion.py:164-171)."""

    def __init__(self, audio_dim: int, video_dim: int, text_dim: int, fusion_dim: int = 512, intermediate_dim: int = 256,
                 num_attention_heads: int = 8, dropout: float = 0.3, use_uncertainty_weighting: bool = True,
                 compute_dtype: str = "fp32", seed: int = 0):
        super().__init__()
        if fusion_dim != 512 or intermediate_dim != 256 or num_attention_heads != 8:
            raise NotImplementedError("the operator path is built for fusion_dim = 512, intermediate_dim = 256, 8 heads "
                                      "(mmdeer_trimodal_attn_* works on 64-wide heads); audio / video / text widths are free")
        for name, d in (("audio_dim", audio_dim), ("video_dim", video_dim), ("text_dim", text_dim)):
            if d <= 0 or d % 4:
                raise NotImplementedError(f"{name} = {d}: input widths must be positive multiples of 4 (16-byte fp32 rows)")
        self.audio_dim, self.video_dim, self.text_dim = audio_dim, video_dim, text_dim
        self.fusion_dim, self.intermediate_dim, self.heads = fusion_dim, intermediate_dim, num_attention_heads
        self.dropout, self.compute_dtype = float(dropout), compute_dtype
        self.use_uncertainty_weighting = use_uncertainty_weighting
        E, F, H = intermediate_dim, fusion_dim, num_attention_heads
        av = _Holder()
        av.audio_projection, av.video_projection = nn.Linear(audio_dim, E), nn.Linear(video_dim, E)
        av.cross_attention = nn.MultiheadAttention(E, H, dropout=dropout, batch_first=True)
        av.fusion_layers = _lin_relu_drop_ln(2 * E, E, dropout)
        self.audio_visual_fusion = av
        tri = _Holder()
        tri.audiovisual_projection, tri.text_projection = nn.Linear(E, F), nn.Linear(text_dim, F)
        tri.modality_attention = nn.MultiheadAttention(F, H, dropout=dropout, batch_first=True)
        tri.final_fusion = _lin_relu_drop_ln(F, F, dropout)
        self.trimodal_fusion = tri
        if use_uncertainty_weighting:        # parameters only (fusion.py:346-382): no working caller in the reference
            g = _Holder()
            g.modality_encoders = nn.ModuleList([nn.Sequential(nn.Linear(d, 128), nn.ReLU(), nn.Linear(128, 64))
                                                 for d in (audio_dim, video_dim, text_dim)])
            g.uncertainty_encoder = nn.Sequential(nn.Linear(3, 64), nn.ReLU(), nn.Linear(64, 32))
            g.gating_network = nn.Sequential(nn.Linear(64 * 3 + 32, 128), nn.ReLU(), nn.Linear(128, 3), nn.Softmax(dim=-1))
            self.uncertainty_gate = g
        self.output_projection = _lin_relu_drop_ln(F, F, dropout)
        gen = torch.Generator().manual_seed(seed)
        for m in self.modules():             # the reference's _initialize_weights (fusion.py:108-117)
            if isinstance(m, nn.Linear):
                with torch.no_grad():
                    bound = (6.0 / (m.in_features + m.out_features)) ** 0.5
                    m.weight.uniform_(-bound, bound, generator=gen)
                    m.bias.zero_()
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        self._drop = _Drop(seed)

    def forward(self, audio_features, video_features, text_features, uncertainties=None) -> Dict[str, torch.Tensor]:
        if self.use_uncertainty_weighting and uncertainties is not None:
            raise TypeError("forward() missing 1 required keyword-only argument: 'uncertainties' "
                            "[UncertaintyAwareGating is unreachable in the reference (fusion.py:148-150 vs :384)]")
        cd, p = self.compute_dtype, self.dropout
        av, tri = self.audio_visual_fusion, self.trimodal_fusion
        B = audio_features.shape[0]
        drop = self._drop.next(self, p)
        E, H = self.intermediate_dim, self.heads
        ap = linear(audio_features, av.audio_projection, cd)
        vp = linear(video_features, av.video_projection, cd)
        mha = av.cross_attention
        att = _AVAttnFn.apply(torch.cat([vp, ap], dim=0), mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                              cd, H, drop, _SITE_AV)
        cat = torch.cat([att[:B], att[B:]], dim=1)        # [audio_attended | video_attended]  (fusion.py:262)
        fl = av.fusion_layers
        avf = _LinActLnFn.apply(cat, fl[0].weight, fl[0].bias, fl[3].weight, fl[3].bias, cd, drop, _SITE_AVF)
        x0 = linear(avf, tri.audiovisual_projection, cd)
        x1 = linear(text_features, tri.text_projection, cd)
        xtok = torch.stack([x0, x1], dim=1).reshape(2 * B, self.fusion_dim)       # row 2b + t
        tm = tri.modality_attention
        qkv = _LinearRaw.apply(xtok, tm.in_proj_weight, tm.in_proj_bias, cd)
        tdrop = (drop[0], drop[1] + 7919, drop[2]) if drop else None
        obar, attn_w = _TriAttnFn.apply(qkv, cd, tdrop, True)
        pooled = linear(obar, tm.out_proj, cd)
        ff = tri.final_fusion
        trif = _LinActLnFn.apply(pooled, ff[0].weight, ff[0].bias, ff[3].weight, ff[3].bias, cd, drop, _SITE_TRIF)
        op = self.output_projection
        fused = _LinActLnFn.apply(trif, op[0].weight, op[0].bias, op[3].weight, op[3].bias, cd, drop, _SITE_OUT)
        # attention weights of the 1-key calls: head-mean of the post-dropout weights = mean over heads of keep / (1 - p)
        if drop:
            lib = _lib.load()
            keep = torch.empty(2 * B, H, dtype=torch.uint8, device=fused.device)
            _lib.check(lib.mmdeer_dropout_mask(_SITE_AV, 2 * B, H, drop[0], drop[1], drop[2], keep.data_ptr(), _lib.current_stream()))
            w = keep.float().mean(dim=1, keepdim=True) / (1.0 - drop[0])
            a2v, v2a = w[:B], w[B:]
        else:
            a2v = v2a = torch.ones(B, 1, device=fused.device)
        return {"fused_features": fused, "audiovisual_features": avf, "trimodal_features": trif,
                "av_attention_weights": {"audio_to_video": a2v, "video_to_audio": v2a},
                "trimodal_attention_weights": attn_w, "uncertainty_weights": None}


class _LinearRaw(torch.autograd.Function):
    """y = x W^T + b for a (weight, bias) pair that is not an ``nn.Linear`` (the packed in_proj of ``nn.MultiheadAttention``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, compute_dtype):
        dt = _dt(compute_dtype)
        xa, w, b = _act(x, dt), weight.detach().to(dt).contiguous(), bias.detach().float().contiguous()
        ex = Exec(compute_dtype)
        M, N, K = xa.shape[0], w.shape[0], w.shape[1]
        y = torch.empty(M, N, dtype=dt, device=xa.device)
        ex.gemm(xa, w, y, M, N, K, K, K, N, bias=b)
        ctx.save_for_backward(xa, w)
        ctx.meta = (compute_dtype, x.dtype, weight.dtype)
        return y.float()

    @staticmethod
    def backward(ctx, g):
        xa, w = ctx.saved_tensors
        compute_dtype, xdt, wdt = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype)
        M, N, K = xa.shape[0], w.shape[0], w.shape[1]
        ga = _convert(ex, g, dt)
        dx = torch.empty(M, K, dtype=dt, device=xa.device)
        ex.dx(ga, N, w, dx, K, M)
        gw, gb = torch.zeros(N, K, device=xa.device), torch.zeros(N, device=xa.device)
        ex.dw(ga, N, xa, K, gw, gb, M, N, K)
        return dx.to(xdt), gw.to(wdt), gb.to(wdt), None
