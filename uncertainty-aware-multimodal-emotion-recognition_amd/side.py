"""Side rows of the hot-path table (a8, a9, a14).

Each class keeps its parameters in the same ``nn.Module`` containers as the reference class it mirrors, so
``state_dict()`` keys and shapes interchange with reference checkpoints; the arithmetic runs on the HIP library.
a8 (``CrossModalAttention``) and a14 (``EnhancedAudioEncoder``'s feature branch) are differentiable: their Linear layers are
the autograd nodes of ``fusions.py`` over ``mmdeer_gemm`` and the attention core / LSTM cell / LayerNorm have backward
operators (``mmdeer_cross_modal_attn_bwd``, ``mmdeer_lstm_cell_t1_bwd``, ``mmdeer_layernorm_bwd``), so they train through
``loss.backward()`` with dropout live in ``.train()`` mode.  a9 is its three encoders only: the rest of that class cannot
execute in the reference (shape bug, parity unpinned).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import ctypes as C

import torch
from torch import nn

from . import _lib, fusions, ops
from .opseq import Exec


def _xavier_(module: nn.Module) -> None:
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)


class CrossModalAttention(nn.Module):
    """Mirror of ``deer.CrossModalAttention`` (reference src/models/deer.py:353-425).

    ``forward(audio, video, text) -> (weighted_audio, weighted_video)``, each (B, feature_dim // num_heads): the
    reference's softmax runs over the head axis and its weighted sum collapses the heads.  ``output_proj`` exists
    (and is checkpointed) but is never applied, as in the reference."""

    def __init__(self, feature_dim: int = 256, num_heads: int = 8, compute_dtype: str = "fp32"):
        super().__init__()
        if feature_dim != 256 or num_heads != 8:
            raise NotImplementedError("the HIP kernel is specialised for feature_dim=256, num_heads=8")
        self.feature_dim, self.num_heads, self.head_dim = feature_dim, num_heads, feature_dim // num_heads
        self.compute_dtype = compute_dtype
        self.query_proj = nn.Linear(feature_dim, feature_dim)
        self.key_proj = nn.Linear(feature_dim, feature_dim)
        self.value_proj = nn.Linear(feature_dim, feature_dim)
        self.output_proj = nn.Linear(feature_dim, feature_dim)
        self.uncertainty_gate = nn.Sequential(nn.Linear(feature_dim * 3, feature_dim), nn.ReLU(),
                                              nn.Linear(feature_dim, 2), nn.Softmax(dim=1))

    def forward(self, audio: torch.Tensor, video: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        for t in (audio, video, text):
            if t.dim() != 2 or t.shape[1] != self.feature_dim:
                raise ValueError(f"expected (B, {self.feature_dim}) features, got {tuple(t.shape)}")
        ops._check_dev(audio, video, text)
        c = self.compute_dtype
        q = fusions.linear(text, self.query_proj, c)
        # key / value projections of both modalities in one GEMM each: rows [0,B) audio, [B,2B) video
        av = torch.cat([audio, video], dim=0)
        k = fusions.linear(av, self.key_proj, c)
        v = fusions.linear(av, self.value_proj, c)
        ctx = torch.cat([audio, video, text], dim=1)
        g = fusions.linear(ctx, self.uncertainty_gate[0], c, relu=True)
        logits = fusions.linear(g, self.uncertainty_gate[2], c)          # 256 -> 2 (padded to the GEMM's column granule inside)
        return _CmaCoreFn.apply(q, k, v, logits, c)


class _CmaCoreFn(torch.autograd.Function):
    """deer.py:399-423 after the projections: per-head scores, softmax over the HEAD axis, head-collapsing weighted sum, gate."""

    @staticmethod
    def forward(ctx, q, k, v, logits, compute_dtype):
        dt = ops._act_dtype(compute_dtype)
        B = q.shape[0]
        qa, ka, va = (t.detach().to(dt).contiguous() for t in (q, k, v))
        gl = logits.detach().float().contiguous()
        oa = torch.empty(B, 32, device=q.device)
        ov = torch.empty_like(oa)
        if B:
            _lib.check(_lib.load().mmdeer_cross_modal_attn_fwd(qa.data_ptr(), ka[:B].data_ptr(), va[:B].data_ptr(), ka[B:].data_ptr(), va[B:].data_ptr(),
                                                               256, gl.data_ptr(), oa.data_ptr(), ov.data_ptr(), B, int(dt == torch.float32),
                                                               _lib.current_stream()))
        ctx.save_for_backward(qa, ka, va, gl)
        ctx.dts = (q.dtype, k.dtype, v.dtype, logits.dtype)
        return oa, ov

    @staticmethod
    def backward(ctx, ga, gv):
        qa, ka, va, gl = ctx.saved_tensors
        B, dt = qa.shape[0], qa.dtype
        dq, dk, dv = torch.empty_like(qa), torch.empty_like(ka), torch.empty_like(va)
        dgl = torch.empty(B, 2, device=qa.device)
        zero = lambda g: torch.zeros(B, 32, device=qa.device) if g is None else g.float().contiguous()   # noqa: E731
        ga, gv = zero(ga), zero(gv)
        if B:
            _lib.check(_lib.load().mmdeer_cross_modal_attn_bwd(qa.data_ptr(), ka[:B].data_ptr(), va[:B].data_ptr(), ka[B:].data_ptr(), va[B:].data_ptr(), 256,
                                                               gl.data_ptr(), ga.data_ptr(), gv.data_ptr(), dq.data_ptr(), dk[:B].data_ptr(), dv[:B].data_ptr(),
                                                               dk[B:].data_ptr(), dv[B:].data_ptr(), dgl.data_ptr(), B, int(dt == torch.float32),
                                                               _lib.current_stream()))
        d = ctx.dts
        return dq.to(d[0]), dk.to(d[1]), dv.to(d[2]), dgl.to(d[3]), None


class _LstmCellFn(torch.autograd.Function):
    """nn.LSTM cell at T = 1, zero initial state: gates (B, ndir * 4H) -> h (B, ndir * H).  The recurrent weights and the forget
    gate multiply zeros: `dead` parameters ride along and receive exact-zero gradients, as autograd gives them."""

    @staticmethod
    def forward(ctx, gates, hidden, ndir, compute_dtype, *dead):
        dt = ops._act_dtype(compute_dtype)
        g = gates.detach().to(dt).contiguous()
        h = ops.lstm_cell_t1(g, hidden, ndir)
        ctx.save_for_backward(g)
        ctx.meta = (hidden, ndir, gates.dtype, [(p.shape, p.dtype) for p in dead])
        return h.float()

    @staticmethod
    def backward(ctx, gh):
        (g,) = ctx.saved_tensors
        hidden, ndir, gdt, dead = ctx.meta
        B = g.shape[0]
        dh = gh.to(g.dtype).contiguous()
        dg = torch.empty_like(g)
        if B:
            _lib.check(_lib.load().mmdeer_lstm_cell_t1_bwd(g.data_ptr(), g.shape[1], dh.data_ptr(), dh.shape[1], dg.data_ptr(), B, hidden, ndir,
                                                           int(g.dtype == torch.float32), _lib.current_stream()))
        return (dg.to(gdt), None, None, None) + tuple(torch.zeros(s, dtype=d, device=g.device) for s, d in dead)


class _LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm behind a plain Linear (no ReLU / dropout below it): mmdeer_layernorm_fwd / _bwd with the mask off."""

    @staticmethod
    def forward(ctx, y, gamma, beta, compute_dtype):
        dt = ops._act_dtype(compute_dtype)
        ya = y.detach().to(dt).contiguous()
        ex = Exec(compute_dtype)
        g32 = gamma.detach().float().contiguous()
        out, mean, rstd = ex.ln_fwd(ya, g32, beta.detach().float().contiguous())
        ctx.save_for_backward(ya, mean, rstd, g32)
        ctx.meta = (compute_dtype, y.dtype, gamma.dtype)
        return out.float()

    @staticmethod
    def backward(ctx, g):
        ya, mean, rstd, g32 = ctx.saved_tensors
        compute_dtype, ydt, pdt = ctx.meta
        ex = Exec(compute_dtype)
        N = ya.shape[1]
        gg, gb = torch.zeros(N, device=ya.device), torch.zeros(N, device=ya.device)
        dz = ex.ln_bwd(g.to(ya.dtype).contiguous(), ya, mean, rstd, g32, gg, gb, 0.0)
        return dz.to(ydt), gg.to(pdt), gb.to(pdt), None


class _DropoutFn(torch.autograd.Function):
    """nn.Dropout on an activation that is not a GEMM output (the LSTM's inter-layer dropout): the library's keep-mask of the
    site (mmdeer_dropout_mask) applied by mmdeer_add_masked; the backward applies the same mask to the gradient."""

    @staticmethod
    def forward(ctx, x, compute_dtype, drop, site):
        dt = ops._act_dtype(compute_dtype)
        xa = x.detach().to(dt).contiguous()
        ex = Exec(compute_dtype, drop)
        B, N = xa.shape
        keep = torch.empty(B, N, dtype=torch.uint8, device=xa.device)
        _lib.check(ex.lib.mmdeer_dropout_mask(site, B, N, float(drop[0]), drop[1], drop[2], keep.data_ptr(), ex.s))
        mask = keep.to(dt)
        out = ex.add(torch.empty_like(xa), xa, mask=mask, scale=ex.scale_of(drop[0]))
        ctx.save_for_backward(mask)
        ctx.meta = (compute_dtype, ex.scale_of(drop[0]), x.dtype)
        return out.float()

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        compute_dtype, scale, xdt = ctx.meta
        ex = Exec(compute_dtype)
        return ex.add(torch.empty_like(mask), g.to(mask.dtype).contiguous(), mask=mask, scale=scale).to(xdt), None, None, None


class ModalityEncoders(nn.Module):
    """The three ``ReLU(Linear)`` encoders of ``deer.HierarchicalDEERFusion`` (reference src/models/deer.py:287-289,
    330-332).  The rest of that class cannot execute in the reference (its ``av_fusion`` expects 512 features and
    receives 64), so only this part is built; parameter names match the reference's."""

    def __init__(self, audio_dim: int = 84, video_dim: int = 256, text_dim: int = 768, hidden_dim: int = 256,
                 compute_dtype: str = "fp32"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.audio_encoder = nn.Linear(audio_dim, hidden_dim)
        self.video_encoder = nn.Linear(video_dim, hidden_dim)
        self.text_encoder = nn.Linear(text_dim, hidden_dim)

    def forward(self, audio, video, text):
        """Differentiable when gradients are enabled (forward GEMM with the ReLU in its epilogue; backward = the (Y > 0) mask,
        dX = dY W and dW = dY^T X + bias gradient as three more ``mmdeer_gemm`` calls per encoder), else the plain operator."""
        c = self.compute_dtype
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                        any(torch.is_tensor(x) and x.requires_grad for x in (audio, video, text))):
            from .fusions import linear
            enc = lambda x, m: linear(x, m, c, relu=True)                              # noqa: E731
        else:
            enc = lambda x, m: ops.linear(x, m.weight, m.bias, relu=True, compute=c)   # noqa: E731
        return enc(audio, self.audio_encoder), enc(video, self.video_encoder), enc(text, self.text_encoder)


class EnhancedAudioEncoder(nn.Module):
    """Feature branch of ``encoders.EnhancedAudioEncoder`` (reference src/models/encoders.py:65-126, 356-389) in
    evaluation mode: (B, 84) or (B, 1, 84) pre-extracted features -> (B, 512).

    With one time step and zero initial state the recurrent weights multiply zeros and the attention pool's
    softmax over time is 1, so the path is: per layer one GEMM against the stacked ``[forward; reverse]`` input
    weights (bias ``b_ih + b_hh``) + the gate kernel, then ``output_projection``.  Raw-waveform input (librosa
    feature extraction on the host) and T > 1 are outside this path."""

    def __init__(self, config: Optional[Dict] = None, compute_dtype: str = "fp32"):
        super().__init__()
        config = config or {}
        self.hidden_dim = config.get("hidden_dim", 512)
        self.num_layers = config.get("num_layers", 2)
        self.dropout = config.get("dropout", 0.3)
        self.bidirectional = config.get("bidirectional", True)
        self.enhanced_features_dim = 84
        self.compute_dtype = compute_dtype
        self.dropout_seed, self._train_step = config.get("dropout_seed", 0), 0     # counter-hash dropout: seed + one tick per training forward
        if not self.bidirectional or self.hidden_dim % 8:
            raise NotImplementedError("built for the reference default: bidirectional, hidden_dim % 8 == 0")
        self.lstm = nn.LSTM(input_size=84, hidden_size=self.hidden_dim // 2, num_layers=self.num_layers, batch_first=True,
                            dropout=self.dropout if self.num_layers > 1 else 0, bidirectional=True)
        self.attention = nn.Sequential(nn.Linear(self.hidden_dim, self.hidden_dim // 2), nn.Tanh(),
                                       nn.Linear(self.hidden_dim // 2, 1), nn.Softmax(dim=1))
        self.output_projection = nn.Sequential(nn.Linear(self.hidden_dim, self.hidden_dim), nn.ReLU(), nn.Dropout(self.dropout),
                                               nn.Linear(self.hidden_dim, self.hidden_dim), nn.LayerNorm(self.hidden_dim))
        _xavier_(self)
        for name, p in self.lstm.named_parameters():   # encoders.py:120-126
            if "weight" in name:
                nn.init.xavier_uniform_(p.data)
            else:
                nn.init.zeros_(p.data)

    def forward(self, audio_input: torch.Tensor) -> torch.Tensor:
        x = audio_input
        if x.shape[-1] != self.enhanced_features_dim:
            raise NotImplementedError("raw-waveform input (host-side librosa feature extraction) is outside the hot path")
        if x.dim() == 3:
            if x.shape[1] != 1:
                raise NotImplementedError("the HIP path covers the single-time-step feature branch (T = 1)")
            x = x[:, 0]
        if x.dim() != 2:
            raise ValueError(f"expected (B, 84) or (B, 1, 84), got {tuple(audio_input.shape)}")
        ops._check_dev(x)
        c = self.compute_dtype
        H = self.hidden_dim // 2
        drop = None
        if self.training and self.dropout > 0:
            drop = (self.dropout, int(self.dropout_seed), self._train_step)
            self._train_step += 1
        h = x
        for layer in range(self.num_layers):
            P = lambda n: getattr(self.lstm, f"{n}_l{layer}")                      # noqa: E731
            w = torch.cat([P("weight_ih"), getattr(self.lstm, f"weight_ih_l{layer}_reverse")], dim=0)
            b = torch.cat([P("bias_ih") + P("bias_hh"), getattr(self.lstm, f"bias_ih_l{layer}_reverse") + getattr(self.lstm, f"bias_hh_l{layer}_reverse")])
            gates = fusions._LinearFn.apply(h, w, b, c, False, None, -1)          # (B, 2 * 4H)
            h = _LstmCellFn.apply(gates, H, 2, c, P("weight_hh"), getattr(self.lstm, f"weight_hh_l{layer}_reverse"))   # (B, 2H) = [forward | reverse]
            if drop is not None and layer + 1 < self.num_layers:                    # nn.LSTM's inter-layer dropout
                h = _DropoutFn.apply(h, c, drop, _SITE_LSTM + layer)
        # attention pool over ONE time step: softmax over time = 1, attended = lstm_out; its parameters get exact zeros (encoders.py:382-383)
        h = _ZeroGradFn.apply(h, *self.attention.parameters())
        op = self.output_projection
        y = fusions.linear(h, op[0], c, relu=True, drop=drop, site=_SITE_OP)
        y = fusions.linear(y, op[3], c)
        return _LayerNormFn.apply(y, op[4].weight, op[4].bias, c)


_SITE_LSTM, _SITE_OP = 112, 120


class _ZeroGradFn(torch.autograd.Function):
    """Identity on x; the extra parameters (unreachable at T = 1) receive exact-zero gradients, as they do in the reference."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.meta = [(p.shape, p.dtype, p.device) for p in params]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g,) + tuple(torch.zeros(s, dtype=d, device=dev) for s, d, dev in ctx.meta)
