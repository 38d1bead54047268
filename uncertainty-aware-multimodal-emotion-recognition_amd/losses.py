"""The loss classes of the path with the reference's constructor / call protocol (a10-a13), on the HIP
library.  Every class is an ``nn.Module`` whose ``forward(predictions, targets)`` takes the prediction dictionaries the
reference classes take and returns the same keys; values are differentiable (one kernel pass produces the loss and the
gradient of its total).  GPU tensors only -- there is no CPU path.

  reference class                                     here
  deer.DEERLoss            (src/models/deer.py:111-195)     DEERLossV1            mmdeer_deer_loss_v1
  losses.DEERLoss          (src/utils/losses.py:40-226)     DEERLoss              mmdeer_nig_loss (one dimension)
  losses.MultiTaskDEERLoss (:229-348)                       MultiTaskDEERLoss     mmdeer_nig_loss
  losses.UncertaintyRegularizationLoss (:351-416)           UncertaintyRegularizationLoss   mmdeer_uncertainty_reg_loss
  losses.CalibrationLoss   (:419-497)                       CalibrationLoss       mmdeer_calibration_loss
  losses.CombinedDEERLoss  (:500-577)                       CombinedDEERLoss
  losses.create_deer_loss  (:580-601)                       create_deer_loss
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib
from .model import _LossFn, make_loss_cfg, multitask_deer_loss
from .spec import DIM_NAMES


def _gpu_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError(f"mmdeer.losses: {what} must be a GPU tensor (there is no CPU fallback)")
    return t.contiguous().float()


def _match(t: torch.Tensor, ref: torch.Tensor, what: str) -> torch.Tensor:
    """Targets / parameters of the reference broadcast against each other ((B,) vs (B,1)); make that explicit."""
    if t.shape == ref.shape:
        return t
    if t.dim() == ref.dim() - 1:
        t = t.unsqueeze(-1)
    try:
        return t.expand_as(ref)
    except RuntimeError as e:
        raise ValueError(f"{what} of shape {tuple(t.shape)} does not broadcast to {tuple(ref.shape)}") from e


# --------------------------------------------------------------------------- deer.DEERLoss (variant 1)
class _V1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, nu, alpha, beta, targets, ew: float, kw: float):
        lib = _lib.load()
        n = mu.numel()
        dev = mu.device
        grads = torch.empty(4, n, dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        scratch = torch.empty(max(1, lib.mmdeer_deer_loss_v1_scratch(n)), dtype=torch.float32, device=dev)
        _lib.check(lib.mmdeer_deer_loss_v1(mu.data_ptr(), nu.data_ptr(), alpha.data_ptr(), beta.data_ptr(), targets.data_ptr(), n,
                                           ew, kw, out.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(),
                                           grads[3].data_ptr(), scratch.data_ptr(), _lib.current_stream()))
        ctx.save_for_backward(grads)
        ctx.shape = mu.shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        (grads,) = ctx.saved_tensors
        s = g_out[0]      # only the total (element 0) is a training objective; the components are reported values
        return tuple((grads[i] * s).view(ctx.shape) for i in range(4)) + (None, None, None)


class DEERLossV1(nn.Module):
    """``deer.DEERLoss`` (src/models/deer.py:111-195): NIG negative log-likelihood + evidence regulariser + clamped KL."""

    def __init__(self, evidence_weight: float = 1.0, kl_weight: float = 1.0):
        super().__init__()
        self.evidence_weight, self.kl_weight = evidence_weight, kl_weight

    def forward(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        mu = _gpu_f32(predictions["mu"], "predictions['mu']")
        nu, alpha, beta = (_gpu_f32(predictions[k], f"predictions['{k}']") for k in ("nu", "alpha", "beta"))
        y = _gpu_f32(targets, "targets")
        if y.dim() == 1:
            y = y.unsqueeze(-1)                                     # deer.py:142-143
        y = _match(y, mu, "targets").contiguous()
        if mu.numel() == 0:
            raise ValueError("DEERLossV1: empty batch")
        out = _V1Fn.apply(mu, nu, alpha, beta, y, float(self.evidence_weight), float(self.kl_weight))
        return {"total_loss": out[0], "nll_loss": out[1], "evidence_reg": out[2], "kl_reg": out[3], "mse": out[4]}


# --------------------------------------------------------------------------- losses.DEERLoss / MultiTaskDEERLoss
class DEERLoss(nn.Module):
    """``losses.DEERLoss`` (src/utils/losses.py:40-226) on (B, D) or (B,) NIG parameters: means over ALL elements, ECE
    over the flattened confidences.  Runs as one dimension of the multi-task kernel on the flattened elements."""

    def __init__(self, reg_weight: float = 0.1, kl_weight: float = 0.01, ece_weight: float = 0.05, epsilon: float = 1e-8):
        super().__init__()
        if epsilon != 1e-8:
            raise NotImplementedError("the kernels embed the reference default epsilon = 1e-8")
        self.reg_weight, self.kl_weight, self.ece_weight, self.epsilon = reg_weight, kl_weight, ece_weight, epsilon

    def forward(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        gamma = predictions.get("gamma", predictions.get("mu"))
        nu = predictions.get("nu", predictions.get("lambda"))
        alpha, beta = predictions.get("alpha"), predictions.get("beta")
        if gamma is None or nu is None or alpha is None or beta is None:
            raise ValueError("Missing required NIG parameters in predictions")      # losses.py:94-95
        gamma, nu, alpha, beta = (_gpu_f32(t, "NIG parameters") for t in (gamma, nu, alpha, beta))
        y = _gpu_f32(targets, "targets")
        if y.dim() == 1 and gamma.dim() == 2:                                       # losses.py:98-104
            y = y.unsqueeze(-1)
        elif y.dim() == 2 and gamma.dim() == 1:
            gamma, nu, alpha, beta = (t.unsqueeze(-1) for t in (gamma, nu, alpha, beta))
        batch = gamma.shape[0]
        shape = torch.broadcast_shapes(gamma.shape, y.shape)
        cols = [t.expand(shape).reshape(-1, 1) for t in (gamma, nu, alpha, beta, y)]
        # dimension 0 carries the data, task weights (3, 0, 0) and no cross term make the kernel's total its total
        three = [c.expand(-1, 3).contiguous() for c in cols]
        cfg = make_loss_cfg(self.reg_weight, self.kl_weight, self.ece_weight, 0.0, (3.0, 0.0, 0.0))
        out, _ = _LossFn.apply(*three, cfg)
        return {"total_loss": out[16], "nll_loss": out[1], "reg_loss": out[2], "kl_loss": out[3],
                "ece_loss": out[4] if self.ece_weight > 0 else torch.zeros((), device=out.device), "batch_size": batch}


class MultiTaskDEERLoss(nn.Module):
    """``losses.MultiTaskDEERLoss`` (src/utils/losses.py:229-348)."""

    def __init__(self, emotion_dims: Optional[List[str]] = None, task_weights: Optional[Dict[str, float]] = None,
                 cross_dim_weight: float = 0.05, **deer_kwargs):
        super().__init__()
        dims = list(emotion_dims) if emotion_dims is not None else list(DIM_NAMES)
        if dims != list(DIM_NAMES):
            raise NotImplementedError("the kernel is specialised for ['valence', 'arousal', 'dominance']")
        self.emotion_dims = dims
        self.task_weights = dict(task_weights) if task_weights else {d: 1.0 for d in dims}
        self.cross_dim_weight = cross_dim_weight
        self.deer_loss = DEERLoss(**deer_kwargs)

    def forward(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        d = self.deer_loss
        cfg = make_loss_cfg(d.reg_weight, d.kl_weight, d.ece_weight, self.cross_dim_weight,
                            tuple(float(self.task_weights.get(n, 1.0)) for n in self.emotion_dims))
        return multitask_deer_loss(predictions, _gpu_f32(targets, "targets"), cfg)


# --------------------------------------------------------------------------- the two extra terms (flat keys)
class _UncRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, beta, dw: float, sw: float):
        lib = _lib.load()
        B, D = alpha.shape
        grads = torch.empty(2, B, D, dtype=torch.float32, device=alpha.device)
        out = torch.empty(3, dtype=torch.float32, device=alpha.device)
        _lib.check(lib.mmdeer_uncertainty_reg_loss(alpha.data_ptr(), beta.data_ptr(), B, D, dw, sw, out.data_ptr(),
                                                   grads[0].data_ptr(), grads[1].data_ptr(), _lib.current_stream()))
        ctx.save_for_backward(grads)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (grads,) = ctx.saved_tensors
        return grads[0] * g_out[0], grads[1] * g_out[0], None, None


class UncertaintyRegularizationLoss(nn.Module):
    """``losses.UncertaintyRegularizationLoss`` (src/utils/losses.py:351-416).  Like the reference it looks up the FLAT
    keys 'alpha' / 'beta' and returns ``{'reg_loss': 0}`` when they are absent (a per-dimension dictionary)."""

    def __init__(self, diversity_weight: float = 0.1, sparsity_weight: float = 0.01):
        super().__init__()
        self.diversity_weight, self.sparsity_weight = diversity_weight, sparsity_weight

    def forward(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        alpha, beta = predictions.get("alpha"), predictions.get("beta")
        if alpha is None or beta is None:
            return {"reg_loss": torch.tensor(0.0)}                                  # losses.py:379-380
        alpha, beta = _gpu_f32(alpha, "alpha"), _gpu_f32(beta, "beta")
        if alpha.dim() == 1:
            alpha, beta = alpha.unsqueeze(-1), beta.unsqueeze(-1)
        if alpha.dim() != 2 or alpha.shape != beta.shape or not 1 <= alpha.shape[1] <= 8 or alpha.shape[0] == 0:
            raise ValueError("alpha / beta must be (B, D) with B >= 1 and 1 <= D <= 8")
        out = _UncRegFn.apply(alpha, beta, float(self.diversity_weight), float(self.sparsity_weight))
        return {"reg_loss": out[0], "diversity_loss": out[1], "sparsity_loss": out[2]}


class _CalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gamma, alpha, beta, targets, edges):
        import ctypes as C
        lib = _lib.load()
        n = gamma.numel()
        nb = len(edges) - 1
        grads = torch.empty(3, n, dtype=torch.float32, device=gamma.device)
        out = torch.empty(1, dtype=torch.float32, device=gamma.device)
        bins = torch.empty(nb, dtype=torch.int32, device=gamma.device)
        _lib.check(lib.mmdeer_calibration_loss_bins(gamma.data_ptr(), alpha.data_ptr(), beta.data_ptr(), targets.data_ptr(), n,
                                                    (C.c_float * (nb + 1))(*edges), nb, out.data_ptr(), bins.data_ptr(),
                                                    grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(),
                                                    _lib.current_stream()))
        ctx.save_for_backward(grads)
        ctx.shape = gamma.shape
        ctx.mark_non_differentiable(bins)
        return out, bins

    @staticmethod
    def backward(ctx, g_out, _g_bins):
        (grads,) = ctx.saved_tensors
        return tuple((grads[i] * g_out[0]).view(ctx.shape) for i in range(3)) + (None, None)


class CalibrationLoss(nn.Module):
    """``losses.CalibrationLoss`` (src/utils/losses.py:419-497) with ``n_bins`` uniform bins (default 15, at most 32): the bin
    boundaries are ``torch.linspace(0, 1, n_bins + 1)`` in fp32, the reference's own rule (:459), evaluated once on the host
    and handed to the kernel.  Flat keys as in the reference (0 when they are absent).  ``bin_strategy='quantile'``
    (data-dependent boundaries through torch.quantile, :461-462) is not built.  ``last_bin_counts`` holds the exact bin
    populations of the last call."""

    def __init__(self, n_bins: int = 15, bin_strategy: str = "uniform"):
        super().__init__()
        if bin_strategy != "uniform":
            raise NotImplementedError("CalibrationLoss: only bin_strategy='uniform' is built (the quantile variant is not)")
        if not 1 <= int(n_bins) <= 32:
            raise NotImplementedError("CalibrationLoss: 1 <= n_bins <= 32")
        self.n_bins, self.bin_strategy = int(n_bins), bin_strategy
        self._edges = [float(x) for x in torch.linspace(0, 1, self.n_bins + 1, dtype=torch.float32)]
        self.last_bin_counts: Optional[torch.Tensor] = None

    def forward(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> torch.Tensor:
        gamma = predictions.get("gamma", predictions.get("mu"))
        alpha, beta = predictions.get("alpha"), predictions.get("beta")
        if gamma is None or alpha is None or beta is None:
            return torch.tensor(0.0)                                                # losses.py:447-448
        gamma, alpha, beta = (_gpu_f32(t, "NIG parameters") for t in (gamma, alpha, beta))
        y = _match(_gpu_f32(targets, "targets"), gamma, "targets").contiguous()
        if gamma.numel() == 0:
            raise ValueError("CalibrationLoss: empty batch")
        out, bins = _CalFn.apply(gamma, alpha, beta, y, self._edges)
        self.last_bin_counts = bins
        return out[0]


class CombinedDEERLoss(nn.Module):
    """``losses.CombinedDEERLoss`` (src/utils/losses.py:500-577): multi-task loss + uncertainty regulariser + 0.1 x
    calibration loss.  As in the reference the two extra terms read flat keys, so on a per-dimension dictionary they are 0
    and ``combined_total_loss == total_loss``."""

    def __init__(self, emotion_dims: Optional[List[str]] = None, deer_config: Optional[Dict] = None,
                 uncertainty_reg_config: Optional[Dict] = None, calibration_config: Optional[Dict] = None,
                 use_uncertainty_reg: bool = True, use_calibration_loss: bool = True):
        super().__init__()
        deer_config = deer_config if deer_config is not None else {"reg_weight": 0.1, "kl_weight": 0.01, "ece_weight": 0.05}
        uncertainty_reg_config = uncertainty_reg_config if uncertainty_reg_config is not None else {"diversity_weight": 0.1, "sparsity_weight": 0.01}
        calibration_config = calibration_config if calibration_config is not None else {"n_bins": 15, "bin_strategy": "uniform"}
        self.deer_loss = MultiTaskDEERLoss(emotion_dims=emotion_dims, **deer_config)
        self.use_uncertainty_reg, self.use_calibration_loss = use_uncertainty_reg, use_calibration_loss
        if use_uncertainty_reg:
            self.uncertainty_reg_loss = UncertaintyRegularizationLoss(**uncertainty_reg_config)
        if use_calibration_loss:
            self.calibration_loss = CalibrationLoss(**calibration_config)

    def forward(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        all_losses = dict(self.deer_loss(predictions, targets))
        total = all_losses["total_loss"]
        if self.use_uncertainty_reg:
            extra = self.uncertainty_reg_loss(predictions, targets)
            total = total + extra["reg_loss"].to(total.device)
            all_losses.update(extra)
        if self.use_calibration_loss:
            cal = self.calibration_loss(predictions, targets)
            total = total + 0.1 * cal.to(total.device)                             # losses.py:568-570
            all_losses["calibration_loss"] = cal
        all_losses["combined_total_loss"] = total
        return all_losses


def create_deer_loss(loss_type: str = "combined", config: Optional[Dict] = None) -> nn.Module:
    """``losses.create_deer_loss`` (src/utils/losses.py:580-601)."""
    config = config or {}
    kind = loss_type.lower()
    if kind == "basic":
        return DEERLoss(**config)
    if kind == "multitask":
        return MultiTaskDEERLoss(**config)
    if kind == "combined":
        return CombinedDEERLoss(**config)
    raise ValueError(f"Unknown loss type: {loss_type}")
