"""Optimiser step on the device: gradient clipping + AdamW fused with the weight pack.

``FusedAdamW`` is a ``torch.optim.Optimizer`` (so the reference's LR schedulers drive it unchanged) whose
``step()`` is one call of ``mmdeer_adamw_step``: ``clip_grad_norm_`` + ``torch.optim.AdamW`` arithmetic on the flat
gradient buffer of ``MultimodalDEER.train_step``, in-place update of the fp32 parameters, and refresh of the packed
bf16 / transposed copies the GEMM kernels read -- the next ``train_step`` starts without a pack pass and nothing
synchronises with the host (reference semantics: src/training/training.py:121-150 optimiser and groups, :219-224
clip + step).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, params: Optional[Iterable] = None, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: float = 0.0):
        """``params``: parameters or torch-style groups (``{'params': [...], 'lr': ...}``); default all parameters of
        ``model``.  Parameters the fused backward never gives a gradient (the unreachable gating MLP) are accepted and
        left untouched, as torch's AdamW does for ``grad is None``."""
        if lr <= 0 or eps <= 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdamW: bad hyper-parameters")
        groups = list(params) if params is not None else list(model.parameters())
        super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.model = model
        self.max_grad_norm = float(max_grad_norm)
        self._live = model.live_parameters()
        hyper = {(g["betas"], g["eps"], g["weight_decay"]) for g in self.param_groups}
        if len(hyper) != 1:
            raise NotImplementedError("FusedAdamW: betas / eps / weight_decay must be the same in every group (lr may differ)")
        owner = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                owner[id(p)] = gi
        missing = [i for i, p in enumerate(self._live) if id(p) not in owner]
        if missing:
            raise ValueError(f"FusedAdamW: {len(missing)} trainable parameter(s) of the model are in no parameter group")
        self._group_of = [owner[id(p)] for p in self._live]
        self._t = 0
        self._exp_avg: Optional[torch.Tensor] = None
        self._exp_avg_sq: Optional[torch.Tensor] = None
        self.last_grad_norm: Optional[torch.Tensor] = None

    def _moments(self, dev):
        if self._exp_avg is None or self._exp_avg.device != dev:
            n = self.model._flat_elems
            self._exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
            self._exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        return self._exp_avg, self._exp_avg_sq

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """One update from the gradients of the last ``train_step``.  Returns the pre-clip global gradient norm
        (device scalar, no sync)."""
        if closure is not None:
            raise NotImplementedError("FusedAdamW: closures are not supported")
        m = self.model
        flat = m._step_flat
        if flat is None:
            raise RuntimeError("FusedAdamW.step() needs the gradients of MultimodalDEER.train_step()")
        dev = flat.device
        wbuf = m._weights(dev)
        exp_avg, exp_avg_sq = self._moments(dev)
        self._t += 1
        g0 = self.param_groups[0]
        lib = _lib.load()
        a = _lib.AdamWArgs()
        a.compute_f32, a.pack_transposed, a.step = m.compute_f32, 1, self._t
        a.beta1, a.beta2 = float(g0["betas"][0]), float(g0["betas"][1])
        a.eps, a.weight_decay = float(g0["eps"]), float(g0["weight_decay"])
        a.max_grad_norm, a.grad_scale = self.max_grad_norm, float(grad_scale)
        lrs = (C.c_float * len(self._live))(*[float(self.param_groups[gi]["lr"]) for gi in self._group_of])
        a.lr = lrs
        a.params = (C.c_void_p * len(self._live))(*[p.data_ptr() for p in self._live])
        a.grads, a.exp_avg, a.exp_avg_sq = flat.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()
        norm = torch.empty((), dtype=torch.float32, device=dev)
        a.grad_norm = norm.data_ptr()
        a.weights, a.weights_bytes = wbuf.data_ptr(), wbuf.numel()
        a.stream = _lib.current_stream()
        _lib.check(lib.mmdeer_adamw_step(C.byref(a)))
        # the parameters changed behind torch's version counters, and the model's packed copies are already current
        m._st.param_gen += 1
        m._st.packed_key = m._param_key(wbuf)
        self.last_grad_norm = norm
        return norm

    # ---- checkpointing (flat moments instead of torch's per-parameter state)
    def state_dict(self) -> Dict:
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"step": self._t, "param_groups": groups,
                "exp_avg": None if self._exp_avg is None else self._exp_avg.detach().cpu(),
                "exp_avg_sq": None if self._exp_avg_sq is None else self._exp_avg_sq.detach().cpu()}

    def load_state_dict(self, state: Dict) -> None:
        self._t = int(state["step"])
        for g, sg in zip(self.param_groups, state["param_groups"]):
            g.update(sg)
        if state.get("exp_avg") is not None:
            dev = self._live[0].device
            self._exp_avg = state["exp_avg"].to(dev).float().contiguous()
            self._exp_avg_sq = state["exp_avg_sq"].to(dev).float().contiguous()


class FlatAdamW(torch.optim.Optimizer):
    """clip_grad_norm_ + AdamW (training.py:121-150, 219-224) for a model whose parameters and gradients live in flat buffers
    (``stackb.CompleteDEERModel._flat``): ONE call of ``mmdeer_adamw_flat`` (two launches, plus one for the transposed copies)
    updates the fp32 parameters, the moments and the compute-dtype copies the GEMMs read.  A ``torch.optim.Optimizer``, so the
    reference's LR schedulers drive it unchanged; default parameter groups follow the reference trainer: parameters with
    'encoder' in their name at ``encoder_lr_scale`` x lr (training.py:125-140).  Parameters the backward never reaches (the
    calibration layer: ``grad is None`` under autograd, which torch's AdamW skips) are left untouched.

    The clipping norm is taken over the WHOLE flat gradient buffer (one pass over contiguous memory): with ``params=`` naming a
    subset (frozen encoders) it includes the excluded parameters' gradients, where ``clip_grad_norm_`` over the optimiser's own
    parameters would not -- zero the frozen ranges of the gradient buffer if that matters.  betas / eps / weight_decay must be the
    same in every group; ``step()`` re-checks it (a scheduler or a caller editing ``param_groups`` later is not ignored silently)."""

    def __init__(self, model, params: Optional[Iterable] = None, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: float = 0.0, encoder_lr_scale: float = 0.5, skip=("calibration_layer.",)):
        if lr <= 0 or eps <= 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FlatAdamW: bad hyper-parameters")
        self.skip = tuple(skip)
        if params is None:
            named = [(n, p) for n, p in model.named_parameters() if not any(n.startswith(s) for s in self.skip)]
            groups = [g for g in ({"params": [p for n, p in named if "encoder" in n], "lr": lr * encoder_lr_scale},
                                  {"params": [p for n, p in named if "encoder" not in n], "lr": lr}) if g["params"]]
        else:
            groups = list(params)
        super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        hyper = {(g["betas"], g["eps"], g["weight_decay"]) for g in self.param_groups}
        if len(hyper) != 1:
            raise NotImplementedError("FlatAdamW: betas / eps / weight_decay must be the same in every group (lr may differ)")
        self.model, self.max_grad_norm = model, float(max_grad_norm)
        self._t = 0
        self._m = self._v = self._scratch = None
        self.last_grad_norm: Optional[torch.Tensor] = None

    def _segments(self, st):
        """Maximal runs of consecutive parameters (flat order) with one learning rate; the alignment gaps inside a run hold
        zero parameters and zero gradients, which the update leaves at zero.  Parameters in no group are not touched."""
        lr_of = {id(p): float(g["lr"]) for g in self.param_groups for p in g["params"]}
        segs = []
        for name, p in self.model.named_parameters():
            if id(p) not in lr_of:
                continue
            lr = lr_of[id(p)]
            lo = st["offs"][name]
            hi = lo + (p.numel() + 63) // 64 * 64
            if segs and segs[-1][2] == lr and segs[-1][1] == lo:
                segs[-1][1] = hi
            else:
                segs.append([lo, hi, lr])
        return segs

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        if closure is not None:
            raise NotImplementedError("FlatAdamW: closures are not supported")
        m = self.model
        dev = next(m.parameters()).device
        st = m._flat(dev)
        if self._m is None or self._m.device != dev or self._m.numel() != st["n"]:
            self._m = torch.zeros(st["n"], dtype=torch.float32, device=dev)
            self._v = torch.zeros_like(self._m)
            self._scratch = torch.empty(256, dtype=torch.float32, device=dev)
        segs = self._segments(st)
        if len({(tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in self.param_groups}) != 1:
            raise NotImplementedError("FlatAdamW: betas / eps / weight_decay must be the same in every group (lr may differ)")
        self._t += 1
        g0 = self.param_groups[0]
        a = _lib.AdamWFlatArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = st["p"].data_ptr(), st["g"].data_ptr(), self._m.data_ptr(), self._v.data_ptr()
        f32 = st["packed"] is st["p"]
        a.packed, a.packed_f32 = (None if f32 else st["packed"].data_ptr()), int(f32)
        a.flat_elems, a.nseg = st["n"], len(segs)
        self._keep = ((C.c_longlong * len(segs))(*[s[0] for s in segs]), (C.c_longlong * len(segs))(*[s[1] - s[0] for s in segs]),
                      (C.c_float * len(segs))(*[s[2] for s in segs]))
        a.seg_begin, a.seg_elems, a.seg_lr = self._keep
        norm = torch.empty((), dtype=torch.float32, device=dev)
        a.scratch, a.grad_norm, a.step = self._scratch.data_ptr(), norm.data_ptr(), self._t
        a.beta1, a.beta2, a.eps, a.weight_decay = float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), float(g0["weight_decay"])
        a.max_grad_norm, a.grad_scale = self.max_grad_norm, float(grad_scale)
        a.stream = _lib.current_stream()
        _lib.check(_lib.load().mmdeer_adamw_flat(C.byref(a)))
        # the kernel wrote the parameters (and their compute-dtype copy) behind torch's version counters: the copy IS current
        m._flat_pack_t(st)                     # ... and the transposed copies of the matrices the dX GEMMs read (one launch)
        st["versions"] = tuple(p._version for p in m.parameters())
        m._packed = None                       # the inference operand image is rebuilt from the new parameters on demand
        self.last_grad_norm = norm
        return norm

    def zero_grad(self, set_to_none: bool = False):   # the fused backward overwrites every gradient it produces
        return None

    def state_dict(self) -> Dict:
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"step": self._t, "param_groups": groups, "exp_avg": None if self._m is None else self._m.detach().cpu(),
                "exp_avg_sq": None if self._v is None else self._v.detach().cpu()}

    def load_state_dict(self, state: Dict) -> None:
        self._t = int(state["step"])
        for g, sg in zip(self.param_groups, state["param_groups"]):
            g.update(sg)
        if state.get("exp_avg") is not None:
            dev = next(self.model.parameters()).device
            self._m = state["exp_avg"].to(dev).float().contiguous()
            self._v = state["exp_avg_sq"].to(dev).float().contiguous()
            self._scratch = torch.empty(256, dtype=torch.float32, device=dev)
