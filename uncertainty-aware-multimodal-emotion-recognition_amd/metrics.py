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
