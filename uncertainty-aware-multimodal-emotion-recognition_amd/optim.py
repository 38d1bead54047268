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
