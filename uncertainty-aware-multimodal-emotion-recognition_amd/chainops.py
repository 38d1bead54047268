"""Host side of ``mmdeer_chain`` / ``mmdeer_repack`` (include/mmdeer.h): a run of sample-local Linear (+ReLU +Dropout)
(+LayerNorm | LayerNorm backward) layers as ONE launch of the layer-chain kernel (csrc/chain.hip), and the fragment-major
weight images that kernel streams.  Host logic only -- tables and buffer ownership; every number comes out of
``libmmdeer_hip.so``.  Used by the Stack B training step (``stackb_train.py``; reference complete_project.py:60-118,
120-184, 307-418 are such runs)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

K_OK = (64, 128, 256, 384, 512, 768)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Chain:
    """One chain launch under construction.  ``seg`` appends a GEMM segment, ``end`` closes the layer the last segment
    belongs to; ``launch`` enqueues it on the current stream."""

    def __init__(self, ex, X: torch.Tensor, ldx: int, K0: int, rows: int, p: float = 0.0, ts: int = 0):
        self.ex = ex
        self.a = _lib.ChainArgs()
        a = self.a
        a.X, a.ldx, a.K0, a.rows, a.samples_per_workgroup, a.nseg = X.data_ptr(), ldx, K0, rows, ts, 0
        a.dropout_p = float(p)
        if ex.drop is not None:
            a.seed, a.offset = ex.drop[1], ex.drop[2]
            if len(ex.drop) > 3 and ex.drop[3] is not None:
                a.offset_dev = ex.drop[3].data_ptr()
        a.stream = ex.s
        self.keep = [X]

    def seg(self, W: torch.Tensor, N: int, K: int, bias=None, relu=0, site=-1, shift=0, dcol=0, kin=0, nout_off=0,
            mask=None, ldm=0, mcol=0, mscale=1.0, res_add=0, res_dup=0) -> "Chain":
        a = self.a
        if a.nseg >= _lib.CHAIN_MAX_SEGS:
            raise ValueError("chain: too many segments")
        s = a.seg[a.nseg]
        a.nseg += 1
        s.W, s.bias, s.N, s.K, s.kin_off, s.nout_off, s.relu = W.data_ptr(), _p(bias), N, K, kin, nout_off, relu
        s.drop_site, s.drop_shift, s.dcol_off = (site if a.dropout_p > 0 else -1), shift, dcol
        s.mask_y, s.ld_mask, s.mask_col0, s.mask_scale = _p(mask), ldm, mcol, mscale
        s.res_add, s.res_dup = res_add, res_dup
        self.keep += [W, bias, mask]
        return self

    def end(self, nout: int, stash=None, ld_stash=0, stash2=None, split=0, ln=None, residual=0, lnb=None) -> "Chain":
        """ln = (gamma, beta, xln, mean, rstd); lnb = (gamma, y, mean, rstd, dz, partial, mask_scale)."""
        s = self.a.seg[self.a.nseg - 1]
        s.end_layer, s.nout = 1, nout
        s.stash, s.ld_stash, s.stash2, s.stash_split = _p(stash), ld_stash, _p(stash2), split
        self.keep += [stash, stash2]
        if ln is not None:
            g, b, xln, mean, rstd = ln
            s.gamma, s.beta, s.xln, s.mean, s.rstd, s.residual = g.data_ptr(), b.data_ptr(), xln.data_ptr(), mean.data_ptr(), rstd.data_ptr(), residual
            self.keep += list(ln)
        if lnb is not None:
            g, y, mean, rstd, dz, part, ms = lnb
            s.lnb_gamma, s.lnb_y, s.lnb_mean, s.lnb_rstd = g.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            s.lnb_dz, s.lnb_partial, s.lnb_mask_scale = dz.data_ptr(), part.data_ptr(), ms
            self.keep += [g, y, mean, rstd, dz, part]
        return self

    def workgroups(self) -> int:
        return int(self.ex.lib.mmdeer_chain_workgroups(self.a.rows, self.a.samples_per_workgroup))

    def launch(self) -> None:
        if self.a.rows:
            _lib.check(self.ex.lib.mmdeer_chain(C.byref(self.a)))


class FragImages:
    """Derived bf16 images of a set of bf16 matrices in one buffer -- fragment-major (what the chain streams) or row-major
    restatements (a column slice, zero-padded rows, several matrices side by side); ``refresh`` rewrites all of them from
    their sources in as few launches as possible (``mmdeer_repack``)."""

    def __init__(self, device):
        self.dev = device
        self.specs: List[tuple] = []   # key, src view, ld_src, rows, cols, cols_valid, transpose, layout, ld_dst, dst_col, extra element offset
        self.off: Dict[str, int] = {}
        self.shape: Dict[str, Tuple[int, int]] = {}
        self.n = 0
        self.buf: Optional[torch.Tensor] = None
        self.jobs = None

    def add(self, key: str, src: torch.Tensor, rows: int, cols: int, *, ld_src: Optional[int] = None, cols_valid: Optional[int] = None,
            transpose: int = 0) -> None:
        """The image of S (rows x cols at ``src``, row stride ld_src, columns >= cols_valid zero) or of S^T."""
        R, Cn = (cols, rows) if transpose else (rows, cols)
        if R % 16 or Cn % 64:
            raise ValueError(f"fragment-major image {key}: {R} x {Cn}")
        self.specs.append((key, src, ld_src if ld_src is not None else src.stride(0), rows, cols, cols if cols_valid is None else cols_valid, transpose,
                           1, 0, 0, 0))
        self.off[key] = self.n
        self.shape[key] = (R, Cn)
        self.n += (R * Cn + 63) // 64 * 64

    def area(self, key: str, rows: int, cols: int) -> None:
        """A row-major [rows][cols] area (zero-filled once) that ``place`` jobs write parts of."""
        self.off[key] = self.n
        self.shape[key] = (rows, cols)
        self.n += (rows * cols + 63) // 64 * 64

    def place(self, key: str, src: torch.Tensor, rows: int, cols: int, *, row0: int = 0, col0: int = 0, ld_src: Optional[int] = None) -> None:
        """S (rows x cols at ``src``) row-major into the area ``key`` at (row0, col0)."""
        R, Cn = self.shape[key]
        if cols % 8 or col0 % 8 or Cn % 8 or row0 + rows > R or col0 + cols > Cn:
            raise ValueError(f"row-major image {key}: {rows} x {cols} at ({row0}, {col0}) of {R} x {Cn}")
        self.specs.append((key, src, ld_src if ld_src is not None else src.stride(0), rows, cols, cols, 0, 0, Cn, col0, row0 * Cn))

    def finish(self) -> None:
        self.buf = torch.zeros(max(self.n, 64), dtype=torch.bfloat16, device=self.dev)
        n = len(self.specs)
        self.jobs = (_lib.RepackJob * max(n, 1))()
        for j, (key, src, ld, rows, cols, cv, tr, layout, ld_dst, dst_col, extra) in enumerate(self.specs):
            J = self.jobs[j]
            J.src, J.dst = src.data_ptr(), self.buf.data_ptr() + 2 * (self.off[key] + extra)
            J.ld_src, J.rows, J.cols, J.cols_valid, J.transpose, J.layout, J.ld_dst, J.dst_col = ld, rows, cols, cv, tr, layout, ld_dst, dst_col

    def refresh(self) -> None:
        if self.specs:
            _lib.check(_lib.load().mmdeer_repack(self.jobs, len(self.specs), _lib.current_stream()))

    def __call__(self, key: str, row0: int = 0) -> torch.Tensor:
        """The image (a flat view); ``row0``: the sub-image of rows [row0, ...) of the restated matrix (row0 % 16 == 0): a
        16-row block of the matrix is contiguous in the image."""
        R, Cn = self.shape[key]
        return self.buf[self.off[key] + row0 * Cn: self.off[key] + R * Cn]

    def mat(self, key: str) -> torch.Tensor:
        """A row-major area as a matrix."""
        R, Cn = self.shape[key]
        return self.buf[self.off[key]: self.off[key] + R * Cn].view(R, Cn)
