"""Host-side agreement / evaluation metrics (numpy).  Out of the kernel scope:
small statistics on (N, 3) arrays; formulas follow the reference's src/utils/metrics.py."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

DIMENSION_NAMES = ("valence", "arousal", "dominance")


class DEERMetrics:
    """Mirror of metrics.DEERMetrics (metrics.py:55-125)."""

    def __init__(self):
        self.dimension_names = list(DIMENSION_NAMES)

    @staticmethod
    def _clean(y_true, y_pred):
        y_true = np.asarray(y_true, dtype=np.float64)
        y_pred = np.asarray(y_pred, dtype=np.float64)
        mask = ~(np.isnan(y_true) | np.isnan(y_pred))
        return y_true[mask], y_pred[mask]

    def concordance_correlation_coefficient(self, y_true, y_pred) -> float:
        """CCC = 2 rho sx sy / (sx^2 + sy^2 + (mx - my)^2), population variance (metrics.py:59-103)."""
        if len(y_true) == 0 or len(y_pred) == 0:
            return 0.0
        t, p = self._clean(y_true, y_pred)
        if t.size == 0:
            return 0.0
        vt, vp = np.var(t), np.var(p)
        if vt == 0 or vp == 0:
            return 0.0          # np.corrcoef would give NaN -> the reference returns 0.0
        rho = np.corrcoef(t, p)[0, 1]
        if np.isnan(rho):
            return 0.0
        den = vt + vp + (np.mean(t) - np.mean(p)) ** 2
        return float(2 * rho * np.sqrt(vt) * np.sqrt(vp) / den) if den != 0 else 0.0

    def mean_absolute_error(self, y_true, y_pred) -> float:
        t, p = self._clean(y_true, y_pred)
        return float(np.mean(np.abs(t - p))) if t.size else float("inf")

    def root_mean_squared_error(self, y_true, y_pred) -> float:
        t, p = self._clean(y_true, y_pred)
        return float(np.sqrt(np.mean((t - p) ** 2))) if t.size else float("inf")


def uncertainty_calibration_error(predictions, targets, uncertainties, n_bins: int = 10) -> float:
    """Quantile-binned ECE of metrics.py:214-279."""
    predictions, targets, uncertainties = (np.asarray(x, dtype=np.float64) for x in (predictions, targets, uncertainties))
    if len(predictions) == 0:
        return 1.0
    errors = np.abs(predictions - targets)
    if errors.ndim > 1:
        errors = errors.mean(axis=1)
        uncertainties = uncertainties.mean(axis=1)
    mask = ~(np.isnan(errors) | np.isnan(uncertainties) | np.isinf(uncertainties))
    if mask.sum() < n_bins:
        return 1.0
    errors, uncertainties = errors[mask], uncertainties[mask]
    edges = np.quantile(uncertainties, np.linspace(0, 1, n_bins + 1))
    edges[0] = 0
    edges[-1] = uncertainties.max() + 1e-6
    ece, n = 0.0, len(errors)
    for i in range(n_bins):
        in_bin = (uncertainties >= edges[i]) & (uncertainties < edges[i + 1])
        if in_bin.sum() > 0:
            ece += in_bin.sum() / n * abs(np.mean(1 - uncertainties[in_bin]) - np.mean(1 - errors[in_bin]))
    return float(ece)


def validation_metrics(predictions, targets, uncertainties: Optional[np.ndarray] = None) -> Dict[str, float]:
    """The dictionary DEERTrainer._compute_validation_metrics builds (training.py:316-353)."""
    m = DEERMetrics()
    out: Dict[str, float] = {}
    predictions, targets = np.asarray(predictions), np.asarray(targets)
    for i, name in enumerate(DIMENSION_NAMES[: predictions.shape[1]]):
        out[f"ccc_{name}"] = m.concordance_correlation_coefficient(targets[:, i], predictions[:, i])
        out[f"mae_{name}"] = m.mean_absolute_error(targets[:, i], predictions[:, i])
        out[f"rmse_{name}"] = m.root_mean_squared_error(targets[:, i], predictions[:, i])
    out["ece"] = uncertainty_calibration_error(predictions, targets, uncertainties) if uncertainties is not None else 0.0
    out["ccc_overall"] = float(np.mean([out[f"ccc_{n}"] for n in DIMENSION_NAMES[: predictions.shape[1]]]))
    return out


def _np_lerp(a: float, b: float, t: float) -> float:
    """numpy's _lerp (lib/_function_base_impl.py), the interpolation of np.quantile's 'linear' method, in float64."""
    d = b - a
    r = a + d * t
    if t >= 0.5:
        r = b - d * (1 - t)
    return a if d == 0 else r


def device_calibration_error(err, unc, n_bins: int = 10) -> float:
    """``uncertainty_calibration_error`` (metrics.py:214-279) on per-sample DEVICE tensors: the per-sample arrays never
    leave the GPU -- an exact order-statistic selection (``mmdeer_eval_quantile_select``) returns the 2 (n_bins + 1)
    values np.quantile interpolates between, the host forms the bin edges from them, and ``mmdeer_eval_ece_bins`` sums
    {count, 1 - uncertainty, 1 - error} per bin: 22 floats down, 11 doubles up, 30 doubles down."""
    import torch
    from . import _lib
    n = int(err.numel())
    if n == 0:
        return 1.0
    lib = _lib.load()
    dev = err.device
    nq = n_bins + 1
    vals = torch.empty(nq, 2, dtype=torch.float32, device=dev)
    frac = torch.empty(nq, dtype=torch.float64, device=dev)
    nv = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(lib.mmdeer_eval_quantile_select(err.data_ptr(), unc.data_ptr(), n, nq, vals.data_ptr(), frac.data_ptr(),
                                               nv.data_ptr(), _lib.current_stream()))
    nvalid = int(nv.item())
    if nvalid < n_bins:
        return 1.0
    v, f = vals.cpu().numpy().astype(np.float64), frac.cpu().numpy()
    edges = np.array([_np_lerp(v[r, 0], v[r, 1], f[r]) for r in range(nq)], dtype=np.float64)
    edges[0] = 0.0
    edges[-1] = v[nq - 1, 0] + 1e-6            # the largest valid uncertainty + 1e-6
    edges_dev = torch.from_numpy(edges).to(dev)
    bins = torch.empty(n_bins, 3, dtype=torch.float64, device=dev)
    _lib.check(lib.mmdeer_eval_ece_bins(err.data_ptr(), unc.data_ptr(), n, edges_dev.data_ptr(), n_bins, bins.data_ptr(),
                                        _lib.current_stream()))
    b = bins.cpu().numpy()
    ece = 0.0
    for cnt, s_conf, s_acc in b:
        if cnt > 0:
            ece += cnt / nvalid * abs(s_conf / cnt - s_acc / cnt)
    return float(ece)


class StreamingMetrics:
    """Validation metrics accumulated on the device: ``update(pred, target, unc)`` per batch is one launch
    of ``mmdeer_eval_accumulate``; ``compute()`` copies 24 doubles -- and, for the quantile-binned calibration error, a few
    dozen scalars (``device_calibration_error``: the per-sample arrays stay in HBM) -- to the host and returns the
    dictionary of ``validation_metrics``."""

    def __init__(self, device):
        import torch
        self._torch = torch
        self.device = torch.device(device)
        self.acc = torch.zeros(3, 8, dtype=torch.float64, device=self.device)
        self._err, self._unc = [], []

    def update(self, predictions, targets, uncertainties=None) -> None:
        import ctypes  # noqa: F401
        from . import _lib
        torch = self._torch
        p = predictions.detach().float().contiguous()
        t = targets.detach().float().contiguous()
        if p.shape != t.shape or p.dim() != 2 or p.shape[1] != 3 or not p.is_cuda:
            raise ValueError("StreamingMetrics.update expects (B, 3) GPU tensors")
        u = uncertainties.detach().float().contiguous() if uncertainties is not None else None
        B = p.shape[0]
        err = torch.empty(B, dtype=torch.float32, device=p.device) if u is not None else None
        unc = torch.empty(B, dtype=torch.float32, device=p.device) if u is not None else None
        lib = _lib.load()
        _lib.check(lib.mmdeer_eval_accumulate(p.data_ptr(), t.data_ptr(), _lib.ptr(u), self.acc.data_ptr(), _lib.ptr(err),
                                              _lib.ptr(unc), B, _lib.current_stream()))
        if u is not None:
            self._err.append(err)
            self._unc.append(unc)

    def compute(self) -> Dict[str, float]:
        acc = self.acc.cpu().numpy()
        out: Dict[str, float] = {}
        for i, name in enumerate(DIMENSION_NAMES):
            n, sp, st, spp, stt, spt, sabs, ssq = acc[i]
            if n == 0:
                out[f"ccc_{name}"], out[f"mae_{name}"], out[f"rmse_{name}"] = 0.0, float("inf"), float("inf")
                continue
            mp, mt = sp / n, st / n
            vp, vt = max(spp / n - mp * mp, 0.0), max(stt / n - mt * mt, 0.0)
            cov = spt / n - mp * mt
            den = vp + vt + (mp - mt) ** 2
            out[f"ccc_{name}"] = float(2.0 * cov / den) if (vp > 0 and vt > 0 and den != 0) else 0.0   # 2 rho sp st == 2 cov
            out[f"mae_{name}"] = float(sabs / n)
            out[f"rmse_{name}"] = float(np.sqrt(ssq / n))
        if self._err:
            torch = self._torch
            out["ece"] = device_calibration_error(torch.cat(self._err), torch.cat(self._unc))
        else:
            out["ece"] = 0.0
        out["ccc_overall"] = float(np.mean([out[f"ccc_{n}"] for n in DIMENSION_NAMES]))
        return out
