"""The alternative fusion modules of the reference's src/models/fusion.py: ``AttentionFusion`` (:504-528),
``BilinearFusion`` (:531-554), ``AdaptiveFusionGating`` (:421-501) and the concatenation fallback of ``create_fusion_module``
(:584-592) -- same constructor arguments, ``state_dict`` keys, forward signatures, return values and error behaviour, forward
and backward.

Host logic only: every layer is a C-ABI operator call on the current stream, wrapped in a ``torch.autograd.Function`` so that
the modules compose with each other and with a caller's own layers.  ``nn.Linear`` = ``mmdeer_gemm`` (forward with bias / ReLU /
counter-hash dropout in the epilogue; ``dX = dY W`` with the ReLU / dropout mask; ``dW = dY^T X`` + bias gradient),
``nn.LayerNorm`` = ``mmdeer_layernorm_fwd / _bwd``, ``nn.Bilinear`` = ``mmdeer_outer_fwd`` + one GEMM against the weight read as
``[out][in1 in2]``, the softmax-weighted sums = ``mmdeer_softmax_mix_fwd / _bwd`` (csrc/fusions.hip).  No CPU path: CPU tensors
raise.  ``compute_dtype``: 'fp32' (exact-fp32 MFMA, the parity configuration) or 'bf16' (bf16 storage, fp32 accumulate); inputs
and outputs are fp32 tensors either way (bf16 inputs are accepted)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _lib, ops
from .opseq import Exec

_SITE_ENC, _SITE_CONCAT = 96, 97


def _dt(compute_dtype: str):
    return ops._act_dtype(compute_dtype)


def _act(x: torch.Tensor, dt) -> torch.Tensor:
    ops._check_dev(x)
    return x.detach().to(dt).contiguous()


def _convert(ex: Exec, src: torch.Tensor, dt) -> torch.Tensor:
    """dtype conversion on the device through the library (fp32 <-> bf16)."""
    src = src.contiguous()
    if src.dtype == dt:
        return src
    dst = torch.empty_like(src, dtype=dt)
    n = src.numel()
    if n % 4:
        return src.to(dt)
    _lib.check(ex.lib.mmdeer_convert(src.data_ptr(), int(src.dtype == torch.float32), dst.data_ptr(), int(dt == torch.float32), n, ex.s))
    return dst


class _Drop:
    """Dropout state of a module: seed + a counter that advances once per training forward (the masks are the library's counter
    hash of (seed, step, site, row, column); the backward regenerates nothing -- ReLU outputs carry their own zeros)."""

    def __init__(self, seed: int = 0):
        self.seed, self.step = int(seed), 0

    def next(self, module: nn.Module, p: float):
        if not module.training or p <= 0:
            return None
        d = (p, self.seed, self.step)
        self.step += 1
        return d


# ------------------------------------------------------------------------------------------------ autograd building blocks
class _LinearFn(torch.autograd.Function):
    """y = drop?(relu?(x W^T + b)) -> fp32.  N is rounded up to a multiple of 8 with zero weight rows: 8 columns = one 16-byte bf16
    row, the narrowest gradient matrix the dX / dW GEMMs' vector loads take."""

    @staticmethod
    def forward(ctx, x, weight, bias, compute_dtype, relu, drop, site):
        dt = _dt(compute_dtype)
        xa, N, K = _act(x, dt), weight.shape[0], weight.shape[1]      # (rejects CPU tensors before anything touches the device)
        ex = Exec(compute_dtype, drop)
        Np = (N + 7) // 8 * 8
        w = weight.detach().to(dt)
        b = bias.detach().float()
        if Np != N:
            w, b = torch.nn.functional.pad(w, (0, 0, 0, Np - N)), torch.nn.functional.pad(b, (0, Np - N))
        w, b = w.contiguous(), b.contiguous()
        B = xa.shape[0]
        y = torch.empty(B, Np, dtype=dt, device=xa.device)
        p = ex.p_of(drop[0]) if drop else 0.0
        ex.gemm(xa, w, y, B, Np, K, K, K, Np, bias=b, relu=int(relu), drop_site=site if p > 0 else -1, p=p)
        ctx.save_for_backward(xa, w, y)
        ctx.meta = (compute_dtype, relu, ex.scale_of(drop[0]) if drop else 1.0, N, x.dtype, weight.dtype)
        return y[:, :N].float()

    @staticmethod
    def backward(ctx, g):
        xa, w, y = ctx.saved_tensors
        compute_dtype, relu, scale, N, xdt, wdt = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype)
        B, Np, K = xa.shape[0], w.shape[0], w.shape[1]
        gy = torch.zeros(B, Np, dtype=dt, device=xa.device) if Np != N else None
        ga = _convert(ex, g, dt)
        if gy is not None:
            gy[:, :N].copy_(ga)
            ga = gy
        if relu:       # (y > 0) * 1 / (1 - p): dropped and clipped elements are both zeros of y
            ga = ex.add(torch.empty_like(ga), ga, mask=y, scale=scale)
        dx = torch.empty(B, K, dtype=dt, device=xa.device)
        ex.dx(ga, Np, w, dx, K, B)
        gw, gb = torch.zeros(Np, K, device=xa.device), torch.zeros(Np, device=xa.device)
        ex.dw(ga, Np, xa, K, gw, gb, B, Np, K)
        return dx.to(xdt), gw[:N].to(wdt), gb[:N].to(wdt), None, None, None, None


def linear(x, layer: nn.Linear, compute_dtype: str, relu: bool = False, drop=None, site: int = -1) -> torch.Tensor:
    if x.dim() != 2 or x.shape[1] != layer.in_features:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} and {layer.in_features}x{layer.out_features})")
    if layer.in_features % 4:
        raise NotImplementedError("input widths must be multiples of 4")
    return _LinearFn.apply(x, layer.weight, layer.bias, compute_dtype, relu, drop, site)


class _LinActLnFn(torch.autograd.Function):
    """LayerNorm(drop(relu(x W^T + b))) -- the Linear-ReLU-Dropout-LayerNorm block (fusion.py:586-591)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, compute_dtype, drop, site):
        dt = _dt(compute_dtype)
        xa, w = _act(x, dt), weight.detach().to(dt).contiguous()
        ex = Exec(compute_dtype, drop)
        B, N, K = xa.shape[0], w.shape[0], w.shape[1]
        y = torch.empty(B, N, dtype=dt, device=xa.device)
        p = ex.p_of(drop[0]) if drop else 0.0
        ex.gemm(xa, w, y, B, N, K, K, K, N, bias=bias.detach().float().contiguous(), relu=1, drop_site=site if p > 0 else -1, p=p)
        g32 = gamma.detach().float().contiguous()
        out, mean, rstd = ex.ln_fwd(y, g32, beta.detach().float().contiguous())
        ctx.save_for_backward(xa, w, y, mean, rstd, g32)
        ctx.meta = (compute_dtype, ex.scale_of(drop[0]) if drop else 1.0, x.dtype, weight.dtype)
        return out.float()

    @staticmethod
    def backward(ctx, g):
        xa, w, y, mean, rstd, g32 = ctx.saved_tensors
        compute_dtype, scale, xdt, wdt = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype)
        B, N, K = xa.shape[0], w.shape[0], w.shape[1]
        gg, gbt = torch.zeros(N, device=xa.device), torch.zeros(N, device=xa.device)
        dz = ex.ln_bwd(_convert(ex, g, dt), y, mean, rstd, g32, gg, gbt, scale)
        dx = ex.dx(dz, N, w, torch.empty(B, K, dtype=dt, device=xa.device), K, B)
        gw, gb = torch.zeros(N, K, device=xa.device), torch.zeros(N, device=xa.device)
        ex.dw(dz, N, xa, K, gw, gb, B, N, K)
        return dx.to(xdt), gw.to(wdt), gb.to(wdt), gg.to(wdt), gbt.to(wdt), None, None, None


class _MixFn(torch.autograd.Function):
    """softmax over the S rows of stacked (B, S, D) and their weighted sum.  Logits = stacked . w_att + b_att (AttentionFusion) or
    given (AdaptiveFusionGating).  Returns (mixed (B, D) fp32, weights (B, S) fp32)."""

    @staticmethod
    def forward(ctx, stacked, w_att, b_att, logits, compute_dtype):
        dt = _dt(compute_dtype)
        P = _act(stacked, dt)
        ex = Exec(compute_dtype)
        B, S, D = P.shape
        if S > 8 or D % 4:
            raise NotImplementedError("softmax mix: at most 8 rows, width a multiple of 4")
        a = _lib.SoftmaxMixArgs()
        keep = []
        a.P, a.ldp, a.sp, a.S, a.D, a.B, a.act_f32 = P.data_ptr(), S * D, D, S, D, B, ex.f32
        if w_att is not None:
            keep = [w_att.detach().float().reshape(-1).contiguous(), b_att.detach().float().reshape(-1).contiguous()]
            a.w_att, a.b_att = keep[0].data_ptr(), keep[1].data_ptr()
        else:
            keep = [logits.detach().float().contiguous()]
            a.logits, a.ld_logits = keep[0].data_ptr(), keep[0].stride(0)
        w8 = torch.empty(B, 8, device=P.device)
        out = torch.empty(B, D, dtype=dt, device=P.device)
        a.weights8, a.out, a.ld_out, a.stream = w8.data_ptr(), out.data_ptr(), D, ex.s
        _lib.check(ex.lib.mmdeer_softmax_mix_fwd(C.byref(a)))
        ctx.save_for_backward(P, w8, *keep)
        ctx.meta = (compute_dtype, w_att is not None, stacked.dtype, None if w_att is None else w_att.dtype, None if logits is None else (logits.dtype, logits.shape[1]))
        wts = w8[:, :S].clone()
        ctx.mark_non_differentiable(wts)
        return out.float(), wts

    @staticmethod
    def backward(ctx, g, _gw):
        P, w8, *keep = ctx.saved_tensors
        compute_dtype, has_att, sdt, adt, ldt = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype)
        B, S, D = P.shape
        dev = P.device
        go = _convert(ex, g, dt)
        dP, dl8 = torch.empty_like(P), torch.empty(B, 8, dtype=dt, device=dev)
        a = _lib.SoftmaxMixArgs()
        a.P, a.ldp, a.sp, a.S, a.D, a.B, a.act_f32 = P.data_ptr(), S * D, D, S, D, B, ex.f32
        if has_att:
            a.w_att, a.b_att = keep[0].data_ptr(), keep[1].data_ptr()
        else:
            a.logits, a.ld_logits = keep[0].data_ptr(), keep[0].stride(0)
        a.weights8, a.dout, a.ld_dout, a.dP, a.dlogits8, a.stream = w8.data_ptr(), go.data_ptr(), D, dP.data_ptr(), dl8.data_ptr(), ex.s
        _lib.check(ex.lib.mmdeer_softmax_mix_bwd(C.byref(a)))
        g_att = g_b = g_logits = None
        if has_att:
            # d w_att = sum_{b,s} ds[b,s] P[b,s,:]: dlogits8^T (8 x B) times P read as (B, S D) -> row s, column block s
            full, rows = torch.zeros(8, S * D, device=dev), torch.zeros(8, device=dev)
            ex.dw(dl8, 8, P.view(B, S * D), S * D, full, rows, B, 8, S * D)
            acc = full[0:1, 0:D]
            for s in range(1, S):
                acc = _add32(ex, acc, full[s:s + 1, s * D:(s + 1) * D])
            g_att = acc.reshape(1, D).to(adt)
            # d b_att = sum of all ds (zero up to rounding: the softmax is shift invariant), by a (1 x 8) . ones GEMM
            tot = torch.zeros(1, 4, device=dev)
            _f32_gemm(ex, rows.view(1, 8), torch.ones(4, 8, device=dev), tot, 1, 4, 8)
            g_b = tot[0, :1].to(adt)
        else:
            g_logits = torch.nn.functional.pad(dl8[:, :S].float(), (0, ldt[1] - S)).to(ldt[0]) if ldt[1] != S else dl8[:, :S].to(ldt[0])
        return dP.to(sdt), g_att, g_b, g_logits, None


def _add32(ex: Exec, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x + y on fp32 (1, N) views through mmdeer_add_masked."""
    out = torch.empty(x.shape[0], x.shape[1], device=x.device)
    _lib.check(ex.lib.mmdeer_add_masked(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), None, 0, 1.0,
                                        x.shape[0], x.shape[1], 1, ex.s))
    return out


def _f32_gemm(ex: Exec, A, W, Cm, M, N, K):
    """Cm = A W^T with fp32 operands whatever the module's compute dtype (tiny bookkeeping products)."""
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, A.stride(0), W.stride(0), Cm.stride(0)
    a.a_f32 = a.w_f32 = a.c_f32 = a.compute_f32 = 1
    a.tile, a.drop_site, a.regen_site, a.mask_scale, a.stream = -1, -1, -1, 1.0, ex.s
    _lib.check(ex.lib.mmdeer_gemm(C.byref(a)))


class _BilinearFn(torch.autograd.Function):
    """nn.Bilinear: y[b, o] = sum_ij x1[b, i] W[o, i, j] x2[b, j] + bias[o]  (fp32 result)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, compute_dtype):
        dt = _dt(compute_dtype)
        a1, a2 = _act(x1, dt), _act(x2, dt)
        ex = Exec(compute_dtype)
        O, I, J = weight.shape
        if J % 4 or J > 1024:
            raise NotImplementedError("bilinear: in2_features must be a multiple of 4 and at most 1024")
        B = a1.shape[0]
        z = torch.empty(B, I * J, dtype=dt, device=a1.device)
        _lib.check(ex.lib.mmdeer_outer_fwd(a1.data_ptr(), I, a2.data_ptr(), J, z.data_ptr(), B, I, J, ex.f32, ex.s))
        w = weight.detach().to(dt).reshape(O, I * J).contiguous()
        y = torch.empty(B, O, device=a1.device)
        ex.gemm(z, w, y, B, O, I * J, I * J, I * J, O, bias=bias.detach().float().contiguous())
        ctx.save_for_backward(a1, a2, z, w)
        ctx.meta = (compute_dtype, (O, I, J), x1.dtype, x2.dtype, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        a1, a2, z, w = ctx.saved_tensors
        compute_dtype, (O, I, J), d1, d2, wdt = ctx.meta
        dt = _dt(compute_dtype)
        ex = Exec(compute_dtype)
        B, dev = a1.shape[0], a1.device
        ga = _convert(ex, g, dt)
        dz = ex.dx(ga, O, w, torch.empty(B, I * J, dtype=dt, device=dev), I * J, B)
        dx1, dx2 = torch.empty(B, I, dtype=dt, device=dev), torch.empty(B, J, dtype=dt, device=dev)
        _lib.check(ex.lib.mmdeer_outer_bwd(dz.data_ptr(), a1.data_ptr(), I, a2.data_ptr(), J, dx1.data_ptr(), I, dx2.data_ptr(), J, B, I, J, ex.f32, ex.s))
        gw, gb = torch.zeros(O, I * J, device=dev), torch.zeros(O, device=dev)
        ex.dw(ga, O, z, I * J, gw, gb, B, O, I * J)
        return dx1.to(d1), dx2.to(d2), gw.view(O, I, J).to(wdt), gb.to(wdt), None


# ------------------------------------------------------------------------------------------------ the modules
class AttentionFusion(nn.Module):
    """``fusion.AttentionFusion`` (src/models/fusion.py:504-528): per-modality projections, one ``Linear(D, 1)`` score per
    projected row, softmax over the modalities, weighted sum."""

    def __init__(self, input_dims: List[int], output_dim: int, compute_dtype: str = "fp32"):
        super().__init__()
        _dt(compute_dtype)
        self.input_dims, self.compute_dtype = input_dims, compute_dtype
        self.projections = nn.ModuleList([nn.Linear(dim, output_dim) for dim in input_dims])
        self.attention = nn.Linear(output_dim, 1)

    def forward(self, modality_features: Sequence[torch.Tensor]) -> torch.Tensor:
        projected = [linear(feat, proj, self.compute_dtype) for proj, feat in zip(self.projections, modality_features)]   # zip: as the reference
        stacked = torch.stack(projected, dim=1)
        out, self.last_attention_weights = _MixFn.apply(stacked, self.attention.weight, self.attention.bias, None, self.compute_dtype)
        return out


class BilinearFusion(nn.Module):
    """``fusion.BilinearFusion`` (src/models/fusion.py:531-554)."""

    def __init__(self, input_dims: List[int], output_dim: int, compute_dtype: str = "fp32"):
        super().__init__()
        _dt(compute_dtype)
        self.input_dims, self.compute_dtype = input_dims, compute_dtype
        if len(input_dims) >= 2:
            self.bilinear = nn.Bilinear(input_dims[0], input_dims[1], output_dim)
            if len(input_dims) > 2:
                self.additional_linear = nn.Linear(sum(input_dims[2:]), output_dim)
        else:
            self.linear = nn.Linear(input_dims[0], output_dim)

    def forward(self, modality_features: Sequence[torch.Tensor]) -> torch.Tensor:
        if len(modality_features) >= 2:
            x1, x2 = modality_features[0], modality_features[1]
            if x1.dim() != 2 or x1.shape[1] != self.bilinear.in1_features or x2.shape[1] != self.bilinear.in2_features or x1.shape[0] != x2.shape[0]:
                raise RuntimeError("bilinear(): input shapes do not match the layer")
            out = _BilinearFn.apply(x1, x2, self.bilinear.weight, self.bilinear.bias, self.compute_dtype)
            if len(modality_features) > 2:
                out = out + linear(torch.cat(list(modality_features[2:]), dim=-1), self.additional_linear, self.compute_dtype)
            return out
        return linear(modality_features[0], self.linear, self.compute_dtype)


class AdaptiveFusionGating(nn.Module):
    """``fusion.AdaptiveFusionGating`` (src/models/fusion.py:421-501).  As in the reference, a strategy name missing from
    ``fusion_modules`` is skipped, an empty strategy list falls back to the concatenation Linear, and listing 'concatenation'
    raises ``TypeError`` (the reference hands that ``nn.Linear`` the tuple of modality tensors, :479)."""

    def __init__(self, input_dims: List[int], fusion_strategies: List[str], hidden_dim: int = 256, compute_dtype: str = "fp32",
                 dropout_seed: int = 0):
        super().__init__()
        _dt(compute_dtype)
        self.input_dims, self.fusion_strategies, self.num_strategies = input_dims, fusion_strategies, len(fusion_strategies)
        self.compute_dtype = compute_dtype
        total = sum(input_dims)
        self.feature_encoder = nn.Sequential(nn.Linear(total, hidden_dim), nn.ReLU(), nn.Dropout(0.3), nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU())
        self.strategy_selector = nn.Sequential(nn.Linear(hidden_dim // 2, self.num_strategies), nn.Softmax(dim=-1))
        self.fusion_modules = nn.ModuleDict({
            "concatenation": nn.Linear(total, hidden_dim),
            "attention": AttentionFusion(input_dims, hidden_dim, compute_dtype),
            "bilinear": BilinearFusion(input_dims, hidden_dim, compute_dtype),
        })
        self._drop = _Drop(dropout_seed)

    def forward(self, *modality_features) -> Dict[str, torch.Tensor]:
        cd = self.compute_dtype
        concatenated = torch.cat(modality_features, dim=-1)
        if self.num_strategies == 0:
            raise NotImplementedError("AdaptiveFusionGating with an empty strategy list")
        enc = self.feature_encoder
        h = linear(concatenated, enc[0], cd, relu=True, drop=self._drop.next(self, enc[2].p), site=_SITE_ENC)
        h = linear(h, enc[3], cd, relu=True)
        logits = linear(h, self.strategy_selector[0], cd)
        fused_outputs = []
        for name in self.fusion_strategies:
            if name in self.fusion_modules:
                if name == "concatenation":
                    raise TypeError("linear(): argument 'input' (position 1) must be Tensor, not tuple")
                fused_outputs.append(self.fusion_modules[name](modality_features))
        if fused_outputs:
            if len(fused_outputs) != self.num_strategies:      # torch.bmm of (B, 1, S) with (B, S', H)
                raise RuntimeError(f"batch2 tensor has {len(fused_outputs)} rows, strategy weights have {self.num_strategies}")
            weighted, weights = _MixFn.apply(torch.stack(fused_outputs, dim=1), None, None, logits, cd)
        else:
            weighted = linear(concatenated, self.fusion_modules["concatenation"], cd)
            weights = _MixFn.apply(logits.detach().unsqueeze(-1).expand(-1, -1, 4), None, None, logits, cd)[1]
        return {"fused_features": weighted, "strategy_weights": weights}


class ConcatFusion(nn.Sequential):
    """The concatenation fallback of ``create_fusion_module`` (src/models/fusion.py:584-592): an ``nn.Sequential`` of
    Linear, ReLU, Dropout, LayerNorm (same child indices, hence the same ``state_dict`` keys) whose forward is one GEMM + one
    LayerNorm launch."""

    def __init__(self, total_dim: int, fusion_dim: int = 512, dropout: float = 0.3, compute_dtype: str = "fp32", dropout_seed: int = 0):
        super().__init__(nn.Linear(total_dim, fusion_dim), nn.ReLU(), nn.Dropout(dropout), nn.LayerNorm(fusion_dim))
        _dt(compute_dtype)
        if fusion_dim not in (256, 512):
            raise NotImplementedError("the LayerNorm kernels are specialised for widths 256 and 512")
        self.compute_dtype = compute_dtype
        self._drop = _Drop(dropout_seed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lin, ln = self[0], self[3]
        if x.dim() != 2 or x.shape[1] != lin.in_features:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} and {lin.in_features}x{lin.out_features})")
        return _LinActLnFn.apply(x, lin.weight, lin.bias, ln.weight, ln.bias, self.compute_dtype, self._drop.next(self, self[2].p), _SITE_CONCAT)
