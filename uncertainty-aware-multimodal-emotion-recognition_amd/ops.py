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
