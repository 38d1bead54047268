"""The loss classes of mmdeer.losses (a10 deer.DEERLoss, a11 losses.DEERLoss, a12 MultiTaskDEERLoss, a13
UncertaintyRegularizationLoss / CalibrationLoss / CombinedDEERLoss) on the GPU through the C-ABI: values against the
vectors captured from the reference classes (tests/golden/loss_cases.npz), values and gradients against the oracle."""
import os

import numpy as np
import pytest
import torch

from mmdeer import losses
from mmdeer.spec import DIM_NAMES

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "loss_cases.npz")
CASES = [("reg", slice(5, 64)), ("all", slice(0, 64)), ("one", slice(7, 8)), ("two", slice(9, 11))]


def _oracle():
    from oracle import deer_oracle as O   # test infrastructure only
    return O


def _nig():
    g = dict(np.load(GOLDEN))
    O = _oracle()
    mu, nu, alpha, beta, *_ = O.nig_activations(torch.from_numpy(g["evidence"]))
    return g, mu, nu, alpha, beta, torch.from_numpy(g["targets"])


def close(val, ref, what, rel=5e-5):
    val, ref = float(val), float(ref)
    if np.isnan(ref) or np.isinf(ref):
        assert (np.isnan(val) and np.isnan(ref)) or val == ref, f"{what}: {val} vs {ref}"
    else:
        assert val == pytest.approx(ref, rel=rel, abs=5e-6), what


def cu(t):
    return t.to(DEV)


@pytest.mark.parametrize("tag,sl", CASES)
def test_loss_classes_match_reference_vectors(tag, sl):
    g, mu, nu, alpha, beta, y = _nig()
    # a10: deer.DEERLoss on one dimension, 1-D targets
    for kw in (1.0, 0.1):
        out = losses.DEERLossV1(kl_weight=kw)({"mu": cu(mu[sl, 0:1]), "nu": cu(nu[sl, 0:1]), "alpha": cu(alpha[sl, 0:1]),
                                               "beta": cu(beta[sl, 0:1])}, cu(y[sl, 0]))
        assert set(out) == {"total_loss", "nll_loss", "evidence_reg", "kl_reg", "mse"}
        for k, v in out.items():
            close(v, g[f"v1.kw{kw}.{tag}.{k}"], f"v1.{kw}.{k}")
    # a11: losses.DEERLoss on (B, 3) tensors, and the 1-D parameter / 2-D target path
    flat = {"gamma": cu(mu[sl]), "nu": cu(nu[sl]), "alpha": cu(alpha[sl]), "beta": cu(beta[sl])}
    lb = losses.DEERLoss()(flat, cu(y[sl]))
    for k in ("total_loss", "nll_loss", "reg_loss", "kl_loss", "ece_loss"):
        close(lb[k], g[f"basic.{tag}.{k}"], "basic." + k)
    assert lb["batch_size"] == int(g[f"basic.{tag}.batch_size"])
    l1 = losses.create_deer_loss("basic")({"mu": cu(mu[sl, 0]), "lambda": cu(nu[sl, 0]), "alpha": cu(alpha[sl, 0]), "beta": cu(beta[sl, 0])},
                                          cu(y[sl, 0:1]))
    close(l1["total_loss"], g[f"basic1d.{tag}.total_loss"], "basic1d")
    # a12 / a13 on a per-dimension dictionary: the extras see no flat keys and return 0, combined == multitask
    pred = {}
    for i, d in enumerate(DIM_NAMES):
        pred[f"{d}_mu"], pred[f"{d}_nu"] = cu(mu[sl, i:i + 1]), cu(nu[sl, i:i + 1])
        pred[f"{d}_alpha"], pred[f"{d}_beta"] = cu(alpha[sl, i:i + 1]), cu(beta[sl, i:i + 1])
    mt = losses.MultiTaskDEERLoss()(pred, cu(y[sl]))
    for k in ("valence_total_loss", "arousal_ece_loss", "dominance_kl_loss", "cross_dim_loss", "total_loss"):
        close(mt[k], g[f"multitask.{tag}.{k}"], k)
    cb = losses.create_deer_loss("combined")(pred, cu(y[sl]))
    close(cb["combined_total_loss"], g[f"combined.{tag}.combined_total_loss"], "combined")
    assert float(cb["reg_loss"]) == 0.0 and float(cb["calibration_loss"]) == 0.0
    # a13 with flat keys
    if sl.stop - sl.start > 1:
        ur = losses.UncertaintyRegularizationLoss()(flat, cu(y[sl]))
        for k in ("reg_loss", "diversity_loss", "sparsity_loss"):
            close(ur[k], g[f"unc_reg.{tag}.{k}"], k, rel=2e-4)
    cal = losses.CalibrationLoss()
    close(cal(flat, cu(y[sl])), g[f"calibration.{tag}"], "calibration")
    assert int(cal.last_bin_counts.sum()) <= 3 * (sl.stop - sl.start)
    for nb in (10, 7):                                    # any n_bins: boundaries = fp32 torch.linspace, the reference's rule
        caln = losses.CalibrationLoss(n_bins=nb)
        close(caln(flat, cu(y[sl])), g[f"calibration{nb}.{tag}"], f"calibration n_bins={nb}")
        assert caln.last_bin_counts.numel() == nb


def test_loss_gradients_match_the_oracle():
    O = _oracle()
    g, mu, nu, alpha, beta, y = _nig()
    sl = slice(5, 64)                                       # the regular rows (finite everywhere)
    leaves = [t[sl].clone().double().requires_grad_(True) for t in (mu, nu, alpha, beta)]
    dev = [t[sl].clone().to(DEV).requires_grad_(True) for t in (mu, nu, alpha, beta)]
    yd, yc = y[sl].double(), y[sl].to(DEV)

    def check(name, tol=2e-3):
        for a, b, nm in zip(dev, leaves, ("mu", "nu", "alpha", "beta")):
            if b.grad is None:
                assert a.grad is None or float(a.grad.abs().max()) == 0.0
                continue
            scale = max(b.grad.abs().max().item(), 1e-12)
            err = (a.grad.cpu().double() - b.grad).abs().max().item() / scale
            assert err < tol, f"{name}: d/d{nm} off by {err:.2e}"
        for t in dev + leaves:
            t.grad = None

    # a10 (variant 1), all three columns at once
    O.deer_loss_v1(*leaves, yd, evidence_weight=0.7, kl_weight=0.3)["total_loss"].backward()
    losses.DEERLossV1(0.7, 0.3)({"mu": dev[0], "nu": dev[1], "alpha": dev[2], "beta": dev[3]}, yc)["total_loss"].backward()
    check("deer.DEERLoss")
    # a11 on (B, 3)
    O.deer_loss_v2(*leaves, yd)["total_loss"].backward()
    losses.DEERLoss()({"gamma": dev[0], "nu": dev[1], "alpha": dev[2], "beta": dev[3]}, yc)["total_loss"].backward()
    check("losses.DEERLoss")
    # a13 extras
    O.uncertainty_reg_loss(leaves[2], leaves[3])["reg_loss"].backward()
    losses.UncertaintyRegularizationLoss()({"alpha": dev[2], "beta": dev[3]}, yc)["reg_loss"].backward()
    check("UncertaintyRegularizationLoss")
    O.calibration_loss(leaves[0], leaves[2], leaves[3], yd).backward()
    losses.CalibrationLoss()({"mu": dev[0], "alpha": dev[2], "beta": dev[3]}, yc).backward()
    check("CalibrationLoss")
    # CombinedDEERLoss with BOTH key styles present: every term contributes
    pred_o, pred_d = {}, {"gamma": dev[0], "nu": dev[1], "alpha": dev[2], "beta": dev[3]}
    for i, d in enumerate(DIM_NAMES):
        for k, j in (("mu", 0), ("nu", 1), ("alpha", 2), ("beta", 3)):
            pred_o[f"{d}_{k}"] = leaves[j][:, i:i + 1]
            pred_d[f"{d}_{k}"] = dev[j][:, i:i + 1]
    ref = (O.multitask_loss(pred_o, yd)["total_loss"] + O.uncertainty_reg_loss(leaves[2], leaves[3])["reg_loss"]
           + 0.1 * O.calibration_loss(leaves[0], leaves[2], leaves[3], yd))
    ref.backward()
    got = losses.CombinedDEERLoss()(pred_d, yc)
    close(got["combined_total_loss"], ref, "combined with flat keys", rel=2e-4)
    got["combined_total_loss"].backward()
    check("CombinedDEERLoss")


def test_large_batch_and_interface():
    O = _oracle()
    B = 5000                                                 # > one workgroup's stride, ragged against 256
    gen = torch.Generator().manual_seed(3)
    e = torch.randn(B, 3, 4, generator=gen)
    mu, nu, alpha, beta, *_ = O.nig_activations(e)
    y = torch.tanh(torch.randn(B, 3, generator=gen))
    flat = {"mu": cu(mu), "nu": cu(nu), "alpha": cu(alpha), "beta": cu(beta)}
    close(losses.DEERLossV1()(flat, cu(y))["total_loss"], O.deer_loss_v1(mu, nu, alpha, beta, y)["total_loss"], "v1 large", rel=1e-4)
    close(losses.CalibrationLoss()(flat, cu(y)), O.calibration_loss(mu, alpha, beta, y), "calibration large", rel=1e-4)
    close(losses.UncertaintyRegularizationLoss()(flat, cu(y))["reg_loss"], O.uncertainty_reg_loss(alpha, beta)["reg_loss"], "unc large", rel=2e-4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.DEERLossV1()({"mu": mu, "nu": nu, "alpha": alpha, "beta": beta}, y)
    with pytest.raises(ValueError, match="Missing required NIG parameters"):
        losses.DEERLoss()({"gamma": cu(mu)}, cu(y))
    with pytest.raises(ValueError, match="Unknown loss type"):
        losses.create_deer_loss("other")
    with pytest.raises(NotImplementedError):
        losses.CalibrationLoss(bin_strategy="quantile")
    with pytest.raises(NotImplementedError):
        losses.CalibrationLoss(n_bins=64)
