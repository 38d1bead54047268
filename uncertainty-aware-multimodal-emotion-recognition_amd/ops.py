"""Thin tensor-level wrappers over single C-ABI operators (include/mmdeer.h).

Used by the side-row modules (``side.py``) and by tests.  Every function runs the HIP
library on the tensors' device and stream; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mmdeer ops need GPU tensors: there is no CPU fallback (libmmdeer_hip.so, gfx950)")


def _act_dtype(compute: str) -> torch.dtype:
    if compute == "fp32":
        return torch.float32
    if compute == "bf16":
        return torch.bfloat16
    raise ValueError(f"compute must be 'fp32' or 'bf16' (got {compute!r})")


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
           compute: str = "fp32", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``relu?(x @ weight.T + bias)`` on ``mmdeer_gemm`` (nn.Linear + optional nn.ReLU).

    x: (M, K); weight: (N, K) as nn.Linear stores it; bias: (N,) fp32.  ``compute='fp32'`` is the exact-fp32 MFMA
    path, ``'bf16'`` casts the operands to bf16 (fp32 accumulation) and returns bf16."""
    _check_dev(x, weight, bias)
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise ValueError(f"linear: shapes {tuple(x.shape)} x {tuple(weight.shape)} do not match")
    dt = _act_dtype(compute)
    M, K = x.shape
    N = weight.shape[0]
    if N % 4 or K % 4:
        raise NotImplementedError("linear: N and K must be multiples of 4")
    xs = x.detach().to(dt).contiguous()
    ws = weight.detach().to(dt).contiguous()
    bs = bias.detach().float().contiguous() if bias is not None else None
    y = out if out is not None else torch.empty(M, N, dtype=dt, device=x.device)
    if y.dtype != dt or y.stride(-1) != 1:
        raise ValueError("linear: out must have the compute dtype and unit inner stride")
    if M == 0:
        return y
    lib = _lib.load()
    a = _lib.GemmArgs()
    a.A, a.W, a.C = xs.data_ptr(), ws.data_ptr(), y.data_ptr()
    a.bias = _lib.ptr(bs)
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw, a.ldc = K, K, y.stride(0)
    f32 = int(dt == torch.float32)
    a.a_f32 = a.w_f32 = a.c_f32 = f32
    a.compute_f32 = f32
    a.relu = int(relu)
    a.tile = -1
    a.drop_site = a.regen_site = -1
    a.mask_scale = 1.0
    a.stream = _lib.current_stream()
    _lib.check(lib.mmdeer_gemm(C.byref(a)))
    return y


def linear_into(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor,
                relu: bool = False, compute: str = "fp32") -> torch.Tensor:
    """``out = relu?(x @ weight.T + bias)`` on ``mmdeer_gemm`` with no copies: ``x`` / ``weight`` / ``out`` are 2-D
    views with unit inner stride (column blocks of wider buffers are fine); ``x`` and ``weight`` already have the
    compute dtype, ``out`` is the compute dtype or fp32."""
    _check_dev(x, weight, bias, out)
    dt = _act_dtype(compute)
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K or out.shape != (M, N):
        raise ValueError(f"linear_into: shapes {tuple(x.shape)} x {tuple(weight.shape)} -> {tuple(out.shape)} do not match")
    if x.dtype != dt or weight.dtype != dt or out.dtype not in (dt, torch.float32):
        raise ValueError("linear_into: operands must have the compute dtype (out: compute dtype or fp32)")
    if x.stride(1) != 1 or weight.stride(1) != 1 or out.stride(1) != 1 or N % 4 or K % 4:
        raise ValueError("linear_into: unit inner strides and N, K multiples of 4 are required")
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise ValueError("linear_into: bias must be contiguous fp32")
    if M == 0:
        return out
    a = _lib.GemmArgs()
    a.A, a.W, a.C = x.data_ptr(), weight.data_ptr(), out.data_ptr()
    a.bias = _lib.ptr(bias)
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw, a.ldc = x.stride(0), weight.stride(0), out.stride(0)
    f32 = int(dt == torch.float32)
    a.a_f32 = a.w_f32 = a.compute_f32 = f32
    a.c_f32 = int(out.dtype == torch.float32)
    a.relu = int(relu)
    a.tile = -1
    a.drop_site = a.regen_site = -1
    a.mask_scale = 1.0
    a.stream = _lib.current_stream()
    _lib.check(_lib.load().mmdeer_gemm(C.byref(a)))
    return out


def residual_layer_norm(y: torch.Tensor, x: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                        out: torch.Tensor) -> torch.Tensor:
    """``out = x + LayerNorm(y)`` (``x`` None: plain LayerNorm) on ``mmdeer_stackb_residual_ln``; 2-D views of one
    activation dtype, N in (256, 512).  ``out`` may alias ``x``."""
    _check_dev(y, x, gamma, beta, out)
    M, N = y.shape
    for t in (x, out):
        if t is not None and (t.shape != (M, N) or t.dtype != y.dtype or t.stride(1) != 1):
            raise ValueError("residual_layer_norm: operands must share shape and dtype and have unit inner stride")
    if y.stride(1) != 1 or y.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("residual_layer_norm: y must be fp32 or bf16 with unit inner stride")
    _lib.check(_lib.load().mmdeer_stackb_residual_ln(
        y.data_ptr(), y.stride(0), _lib.ptr(x), x.stride(0) if x is not None else 0, gamma.data_ptr(), beta.data_ptr(),
        out.data_ptr(), out.stride(0), M, N, int(y.dtype == torch.float32), _lib.current_stream()))
    return out


def layer_norm(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm(N) (eps 1e-5, biased variance) on ``mmdeer_layernorm_fwd``; returns fp32 (M, N)."""
    _check_dev(y, gamma, beta)
    M, N = y.shape
    if y.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("layer_norm: input must be fp32 or bf16")
    ys = y.contiguous()
    out = torch.empty(M, N, dtype=ys.dtype, device=y.device)
    out32 = torch.empty(M, N, dtype=torch.float32, device=y.device)
    mean = torch.empty(M, dtype=torch.float32, device=y.device)
    rstd = torch.empty_like(mean)
    if M == 0:
        return out32
    lib = _lib.load()
    _lib.check(lib.mmdeer_layernorm_fwd(ys.data_ptr(), out.data_ptr(), out32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        gamma.detach().float().contiguous().data_ptr(),
                                        beta.detach().float().contiguous().data_ptr(), M, N,
                                        int(ys.dtype == torch.float32), _lib.current_stream()))
    return out32


def cross_modal_attention_core(q, k_audio, v_audio, k_video, v_video, gate_logits):
    """``mmdeer_cross_modal_attn_fwd``: (B,256) projections + (B,2) gate logits -> two (B,32) fp32 tensors."""
    _check_dev(q, k_audio, v_audio, k_video, v_video, gate_logits)
    B = q.shape[0]
    dt = q.dtype
    ts = [t.contiguous() for t in (q, k_audio, v_audio, k_video, v_video)]
    for t in ts:
        if t.shape != (B, 256) or t.dtype != dt:
            raise ValueError("cross_modal_attention_core: projections must all be (B, 256) of one dtype")
    gl = gate_logits.float().contiguous()
    oa = torch.empty(B, 32, dtype=torch.float32, device=q.device)
    ov = torch.empty_like(oa)
    lib = _lib.load()
    _lib.check(lib.mmdeer_cross_modal_attn_fwd(*[t.data_ptr() for t in ts], 256, gl.data_ptr(), oa.data_ptr(), ov.data_ptr(),
                                               B, int(dt == torch.float32), _lib.current_stream()))
    return oa, ov


def lstm_cell_t1(gates: torch.Tensor, hidden: int, ndir: int) -> torch.Tensor:
    """``mmdeer_lstm_cell_t1``: gates (B, ndir*4*hidden), bias included -> h (B, ndir*hidden), same dtype."""
    _check_dev(gates)
    B = gates.shape[0]
    if gates.shape[1] != ndir * 4 * hidden:
        raise ValueError("lstm_cell_t1: gates must be (B, ndir*4*hidden)")
    g = gates.contiguous()
    out = torch.empty(B, ndir * hidden, dtype=g.dtype, device=g.device)
    lib = _lib.load()
    _lib.check(lib.mmdeer_lstm_cell_t1(g.data_ptr(), g.shape[1], out.data_ptr(), out.shape[1], B, hidden, ndir,
                                       int(g.dtype == torch.float32), _lib.current_stream()))
    return out
