"""Operator sequencing helper: one object per pass that turns nn.Linear / nn.LayerNorm layers into ``mmdeer_gemm`` /
``mmdeer_layernorm_*`` calls on the current stream -- forward (bias / ReLU / counter-hash dropout in the epilogue),
``dX = dY W`` (with the ``(Y > 0) * scale`` mask of the layer below, or a regenerated dropout factor), ``dW = dY^T X`` (+ bias
gradient from the same launch).  Host logic only: every number comes out of ``libmmdeer_hip.so``.  Used by the modules that
are sequences of such layers around a few row kernels (``stackb_train.py``, ``fusions.py``)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Exec:
    """One training step's launches.  ``drop``: (p_config, seed, step) or None (no dropout anywhere)."""

    def __init__(self, compute_dtype: str, drop=None):
        """drop: None (no dropout anywhere) or (p_config, seed, step[, device int64 counter tensor added to step on the device])."""
        self.f32 = int(compute_dtype == "fp32")
        self.dt = torch.float32 if self.f32 else torch.bfloat16
        self.drop = drop
        self.lib = _lib.load()
        self.s = _lib.current_stream()
        self.deferred = None       # a list: dw() records its problems instead of launching them (flush_dw runs them grouped)
        self.folds = None          # a list: ln_bwd() leaves its gamma / beta partials unfolded, copy1d() records copies (flush_folds)

    # ---- operator wrappers ------------------------------------------------------------------------------------------
    def gemm(self, A, W, Cm, M, N, K, lda, ldw, ldc, *, bias=None, relu=0, ta=0, tw=0, Y=None, ldy=0, mask_scale=1.0, bias_grad=None,
             drop_site=-1, drop_shift=0, regen_site=-1, p=0.0, splitk=1):
        if M == 0 or N == 0:
            return Cm
        a = _lib.GemmArgs()
        a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
        a.bias, a.bias_grad, a.Y = _ptr(bias), _ptr(bias_grad), _ptr(Y)
        a.M, a.N, a.K, a.lda, a.ldw, a.ldc, a.ldy = M, N, K, lda, ldw, ldc, ldy
        a.a_f32, a.w_f32, a.c_f32 = int(A.dtype == torch.float32), int(W.dtype == torch.float32), int(Cm.dtype == torch.float32)
        a.y_f32 = int(Y is not None and Y.dtype == torch.float32)
        a.trans_a, a.trans_w, a.relu = ta, tw, relu
        a.compute_f32, a.tile = self.f32, -1
        a.drop_site, a.drop_shift, a.regen_site = drop_site, drop_shift, regen_site
        a.mask_scale = mask_scale
        if self.drop is not None:
            a.dropout_p, a.seed, a.offset = p, self.drop[1], self.drop[2]
            if len(self.drop) > 3 and self.drop[3] is not None:
                a.offset_dev = self.drop[3].data_ptr()          # device counter added to the offset (HIP-graph replays)
        a.stream = self.s
        slab = None
        if splitk > 1 and Cm.dtype == torch.float32 and Cm.is_contiguous():
            slab = torch.empty(splitk * ((M * N + M + 3) // 4 * 4), dtype=torch.float32, device=Cm.device)
            a.splitk, a.slab = splitk, slab.data_ptr()
        _lib.check(self.lib.mmdeer_gemm(C.byref(a)))
        return Cm

    def p_of(self, p):          # effective dropout probability of a site
        return float(p) if self.drop is not None and p > 0 else 0.0

    def scale_of(self, p):
        p = self.p_of(p)
        return 1.0 / (1.0 - p) if p > 0 else 1.0

    def linear(self, x, ldx, w, b, out, ldo, M, relu=0, site=-1, p=0.0, shift=0):
        """out = drop?(relu?(x w^T + b)); w: (N, K) parameter in the compute dtype (packed by the caller)."""
        N, K = w.shape
        p = self.p_of(p)
        return self.gemm(x, w, out, M, N, K, ldx, w.stride(0), ldo, bias=b, relu=relu, drop_site=site if p > 0 else -1, drop_shift=shift, p=p)

    def dx(self, dy, ldy_, w, out, ldo, M, mask=None, ldm=0, mask_scale=1.0, regen_site=-1, shift=0, p=0.0, wt=None):
        """out = (dy w) [* ((mask > 0) * mask_scale)] [* regenerated dropout factor]; w: (N, K) as stored.  ``wt``: the
        transposed copy (K, N) -- the product then runs as an NT GEMM (both operands reduction-contiguous) on the LDS-DMA
        kernel instead of the register-staged one (11 us -> 6 us per layer at B = 4096)."""
        N, K = w.shape
        p = self.p_of(p)
        if wt is not None:
            return self.gemm(dy, wt, out, M, K, N, ldy_, wt.stride(0), ldo, tw=0, Y=mask, ldy=ldm, mask_scale=mask_scale,
                             regen_site=regen_site if p > 0 else -1, drop_shift=shift, p=p)
        return self.gemm(dy, w, out, M, K, N, ldy_, w.stride(0), ldo, tw=1, Y=mask, ldy=ldm, mask_scale=mask_scale,
                         regen_site=regen_site if p > 0 else -1, drop_shift=shift, p=p)

    def dw(self, dy, ldy_, x, ldx, gw, gb, M, N, K):
        """gw (N, K) fp32 = dy^T x, gb (N,) = column sums of dy; dy: (M, N) view, x: (M, K) view."""
        if M == 0:                       # empty batch: the sums are the zeros the caller allocated
            return gw
        if M * min(ldy_, ldx) < 8:       # an operand below the GEMM's 8-element minimum (M = 1, a 4-wide matrix): append a zero row
            d2, x2 = torch.zeros(M + 1, N, dtype=dy.dtype, device=dy.device), torch.zeros(M + 1, K, dtype=x.dtype, device=x.device)
            d2[:M].copy_(dy[:M, :N]); x2[:M].copy_(x[:M, :K])
            dy, ldy_, x, ldx, M = d2, N, x2, K, M + 1
        if self.deferred is not None and gw.is_contiguous() and gw.stride(0) == K:
            # grouped form (mmdeer_gemm_batch): up to 16 weight-gradient problems per launch + one fold, as mmdeer_backward does
            a = _lib.GemmArgs()
            a.A, a.W, a.C, a.bias_grad = dy.data_ptr(), x.data_ptr(), gw.data_ptr(), _ptr(gb)
            a.M, a.N, a.K, a.lda, a.ldw, a.ldc = N, K, M, ldy_, ldx, K
            a.a_f32, a.w_f32, a.c_f32 = int(dy.dtype == torch.float32), int(x.dtype == torch.float32), 1
            a.trans_a, a.trans_w, a.compute_f32, a.tile = 1, 1, self.f32, -1
            a.drop_site = a.regen_site = -1
            a.mask_scale = 1.0
            self.deferred.append((a, (dy, x, gw, gb)))
            return gw
        # few output tiles, deep reduction over the batch: split it into K-slices (fp32 slabs, folded by the library in index order)
        tiles = ((N + 255) // 256) * ((K + 255) // 256)
        splitk = max(1, min(16, 256 // tiles, M // 256))
        return self.gemm(dy, x, gw, N, K, M, ldy_, ldx, gw.stride(0), ta=1, tw=1, bias_grad=gb, splitk=splitk)

    def flush_dw(self):
        """Run the recorded weight-gradient problems: grouped launches + one deterministic slab fold per group."""
        if not self.deferred:
            return
        n = len(self.deferred)
        arr = (_lib.GemmArgs * n)(*[a for a, _ in self.deferred])
        dev = self.deferred[0][1][2].device
        need = int(self.lib.mmdeer_gemm_batch_slab_elems(arr, n))
        slab = torch.empty(max(need, 4), dtype=torch.float32, device=dev)
        _lib.check(self.lib.mmdeer_gemm_batch(arr, n, slab.data_ptr(), slab.numel(), self.s))
        self.deferred = []

    def ln_fwd(self, y, gamma, beta):
        M, N = y.shape
        out = torch.empty_like(y)
        mean = torch.empty(M, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        if M:
            _lib.check(self.lib.mmdeer_layernorm_fwd(y.data_ptr(), out.data_ptr(), None, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                     beta.data_ptr(), M, N, self.f32, self.s))
        return out, mean, rstd

    def copy1d(self, dst, src):
        """dst <- src (fp32).  Recorded for the batched fold launch when both are dense runs of a multiple of 4 elements."""
        n = src.numel()
        if (self.folds is not None and n % 4 == 0 and n > 0 and dst.is_contiguous() and src.is_contiguous() and dst.numel() == n
                and dst.data_ptr() % 16 == 0 and src.data_ptr() % 16 == 0):
            self.folds.append((src, dst, 1, n, n))
        else:
            dst.copy_(src)

    def flush_folds(self):
        """ONE launch for every recorded fold / copy (mmdeer_reduce_batch)."""
        if not self.folds:
            return
        n = len(self.folds)
        vp = C.c_void_p
        src = (vp * n)(*[f[0].data_ptr() for f in self.folds]); dst = (vp * n)(*[f[1].data_ptr() for f in self.folds])
        nparts = (C.c_int32 * n)(*[f[2] for f in self.folds]); cnt = (C.c_int32 * n)(*[f[3] for f in self.folds])
        stride = (C.c_longlong * n)(*[f[4] for f in self.folds])
        _lib.check(self.lib.mmdeer_reduce_batch(n, src, dst, nparts, cnt, stride, self.s))
        self.folds = []

    def ln_bwd(self, dout, y, mean, rstd, gamma, ggamma, gbeta, mask_scale):
        """dz = (y > 0) * mask_scale * LayerNorm'(dout); ggamma / gbeta fp32 (N,)."""
        M, N = y.shape
        dz = torch.empty_like(y)
        if M and self.folds is not None and N % 4 == 0:
            np_ = self.lib.mmdeer_layernorm_bwd_nparts(M)
            part = torch.empty(np_ * 2 * N, dtype=torch.float32, device=y.device)
            _lib.check(self.lib.mmdeer_layernorm_bwd(dout.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                     dz.data_ptr(), None, None, part.data_ptr(), M, N, self.f32, mask_scale, self.s))
            self.folds.append((part, ggamma, np_, N, 2 * N))
            self.folds.append((part[N:], gbeta, np_, N, 2 * N))
            return dz
        if M:
            part = torch.empty(self.lib.mmdeer_layernorm_bwd_nparts(M) * 2 * N, dtype=torch.float32, device=y.device)
            _lib.check(self.lib.mmdeer_layernorm_bwd(dout.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                     dz.data_ptr(), ggamma.data_ptr(), gbeta.data_ptr(), part.data_ptr(), M, N, self.f32,
                                                     mask_scale, self.s))
        return dz

    def add(self, out, x, y=None, mask=None, scale=1.0):
        M, N = x.shape
        _lib.check(self.lib.mmdeer_add_masked(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), _ptr(y), y.stride(0) if y is not None else 0,
                                              _ptr(mask), mask.stride(0) if mask is not None else 0, scale, M, N, self.f32, self.s))
        return out
