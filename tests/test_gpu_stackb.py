"""Stack B (complete_project.CompleteDEERModel, eval forward) on the GPU through the C-ABI, against the
golden vectors captured from the reference and against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from mmdeer import stackb, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TENSOR_KEYS = ("mu_all", "uncertainty_all", "calibrated_uncertainty", "attention_weights", "modality_uncertainties", "fused_features")


def _shapes():
    with open(os.path.join(GOLDEN, "stackb_state_dict_names.json")) as fh:
        return json.load(fh)


def _model(compute="fp32", tag="stackb"):
    m = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype=compute)
    P = {k: torch.from_numpy(v) for k, v in synth.module_fill(tag, _shapes()).items()}
    m.load_state_dict(P)
    return m.to("cuda:0").eval(), P


def _inputs(B, seed):
    b = synth.make_batch(B, seed=seed)
    return [torch.from_numpy(b[k]) for k in ("audio", "video", "text")]


def _oracle():
    from oracle import deer_oracle as O   # test infrastructure only
    return O


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_stackb_matches_reference_golden(compute):
    g = np.load(os.path.join(GOLDEN, "stackb_B9.npz"))
    m, _ = _model(compute)
    out = m(*(x.to("cuda:0") for x in _inputs(9, 78)))
    keys = [k[4:] for k in g.files if k.startswith("out.")]
    assert sorted(out) == sorted(keys)
    # fp32: the exact-fp32 MFMA path; bf16: operands rounded to bf16 at every layer (12 layers deep)
    tol = dict(rtol=2e-4, atol=2e-5) if compute == "fp32" else dict(rtol=0.08, atol=0.06)
    for k in keys:
        assert out[k].dtype == torch.float32 and tuple(out[k].shape) == g["out." + k].shape, k
        np.testing.assert_allclose(out[k].cpu().numpy(), g["out." + k], err_msg=k, **tol)


@pytest.mark.parametrize("B", [1, 1027])
def test_stackb_matches_oracle_other_parameters_and_ragged_batch(B):
    O = _oracle()
    m, P = _model("fp32", tag="stackb2")
    xs = _inputs(B, 5)
    out = m(*(x.to("cuda:0") for x in xs))
    with torch.no_grad():
        ref = O.stackb_forward(P, *xs)
    for k in TENSOR_KEYS + ("valence_nu", "arousal_alpha", "dominance_beta", "valence_aleatoric_uncertainty", "arousal_epistemic_uncertainty"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=3e-4, atol=3e-5, err_msg=k)
    p, u = m.get_predictions_and_uncertainties(out)
    assert p is out["mu_all"] and u is out["calibrated_uncertainty"]
    # softmax rows sum to one; sigmoid outputs stay in (0, 1)
    np.testing.assert_allclose(out["attention_weights"].sum(1).cpu().numpy(), np.ones(B, np.float32), rtol=1e-5)
    for k in ("modality_uncertainties", "calibrated_uncertainty"):
        v = out[k].cpu().numpy()
        assert (v > 0).all() and (v < 1).all()


def test_stackb_missing_modality_and_empty_batch():
    O = _oracle()
    m, P = _model("fp32")
    a, v, t = _inputs(6, 11)
    v = torch.zeros_like(v)                     # a missing modality arrives as zeros (SURVEY 8c edge cases)
    out = m(a.cuda(), v.cuda(), t.cuda())
    with torch.no_grad():
        ref = O.stackb_forward(P, a, v, t)
    for k in TENSOR_KEYS:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=3e-4, atol=3e-5, err_msg=k)
    out = m(a[:0].cuda(), v[:0].cuda(), t[:0].cuda())
    assert out["mu_all"].shape == (0, 3) and out["fused_features"].shape == (0, 512) and out["valence_mu"].shape == (0,)


def test_stackb_other_encoder_depths_and_input_widths():
    """encoder_layers and the input widths are ModelConfig fields (complete_project.py:33-56), not constants."""
    O = _oracle()
    for layers, dims, compute, tol in ((1, (40, 64, 128), "fp32", dict(rtol=3e-4, atol=3e-5)), (0, (84, 256, 768), "fp32", dict(rtol=3e-4, atol=3e-5)),
                                       (4, (200, 256, 768), "bf16", dict(rtol=0.08, atol=0.06))):
        cfg = stackb.ModelConfig(audio_dim=dims[0], video_dim=dims[1], text_dim=dims[2], encoder_layers=layers)
        m = stackb.CompleteDEERModel(cfg, compute_dtype=compute)
        P = {k: torch.from_numpy(v) for k, v in synth.module_fill(f"stackb.l{layers}", {k: tuple(v.shape) for k, v in m.state_dict().items()}).items()}
        m.load_state_dict(P)
        m = m.to("cuda:0").eval()
        xs = [torch.from_numpy(synth.normal(700 + i, 37 * d).reshape(37, d).astype(np.float32)) for i, d in enumerate(dims)]
        out = m(*(x.cuda() for x in xs))
        with torch.no_grad():
            ref = O.stackb_forward(P, *xs, layers=layers)
        for k in TENSOR_KEYS:
            np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), err_msg=f"layers={layers} {k}", **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [256, 512])
def test_residual_layer_norm_operator(dtype, N):
    from mmdeer import ops
    M = 133
    y = torch.from_numpy(synth.normal(810, M * N).reshape(M, N).astype(np.float32)).cuda().to(dtype)
    x = torch.from_numpy(synth.normal(811, M * 2 * N).reshape(M, 2 * N).astype(np.float32)).cuda().to(dtype)[:, N:]   # a column block
    g = torch.from_numpy((1 + 0.1 * synth.normal(812, N)).astype(np.float32)).cuda()
    b = torch.from_numpy((0.1 * synth.normal(813, N)).astype(np.float32)).cuda()
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    for res in (None, x):
        out = ops.residual_layer_norm(y, res, g, b, torch.empty(M, N, dtype=dtype, device="cuda:0"))
        ref = torch.nn.functional.layer_norm(y.float(), (N,), g, b, 1e-5) + (res.float() if res is not None else 0)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), **tol)
    with pytest.raises(RuntimeError, match="256 or 512"):
        ops.residual_layer_norm(y[:, :128], None, g, b, torch.empty(M, 128, dtype=dtype, device="cuda:0"))


def test_stackb_interface():
    m, _ = _model("fp32")
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == _shapes()
    xs = [x.to("cuda:0") for x in _inputs(4, 3)]
    o = m.train()(*xs)                                       # training mode: differentiable (mu, nu, alpha, beta)
    assert o["valence_mu"].requires_grad and o["mu_all"].requires_grad and not o["calibrated_uncertainty"].requires_grad
    m.eval()
    assert not m(*xs)["mu_all"].requires_grad
    with pytest.raises(ValueError, match="expected features"):
        m(xs[0], xs[1][:, :128], xs[2])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(*(x.cpu() for x in xs))
    # parameters edited in place are picked up (the packed operand images are keyed on parameter versions)
    before = m(*xs)["mu_all"].clone()
    with torch.no_grad():
        m.prediction_heads["valence"].evidence_network[6].bias.add_(1.0)
    after = m(*xs)["mu_all"]
    np.testing.assert_allclose((after - before).cpu().numpy(), np.tile(np.float32([1, 0, 0]), (4, 1)), atol=1e-5)
    with pytest.raises(NotImplementedError):
        stackb.CompleteDEERModel(stackb.ModelConfig(encoder_dim=128))


def test_stackb_graph_replay_equals_eager():
    m, _ = _model("bf16")
    xs = [x.to("cuda:0") for x in _inputs(64, 21)]
    replay = m.capture(*xs)
    ys = [x.to("cuda:0") for x in _inputs(64, 22)]
    got = {k: v.clone() for k, v in replay(*ys).items()}
    want = m(*ys)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    again = replay(*xs)
    assert torch.equal(again["mu_all"], m(*xs)["mu_all"])


def test_stackb_bf16_agrees_with_fp32_by_ccc():
    """bf16 operands through ~14 layers: per-dimension concordance (metrics.py:85-101 formula) of the bf16 outputs with the
    fp32 outputs at B = 1024, reference-style (Xavier) weights."""
    torch.manual_seed(11)
    m32 = stackb.CompleteDEERModel(compute_dtype="fp32")
    m16 = stackb.CompleteDEERModel(compute_dtype="bf16")
    m16.load_state_dict(m32.state_dict())
    m32, m16 = m32.to("cuda:0").eval(), m16.to("cuda:0").eval()
    xs = [x.to("cuda:0") for x in _inputs(1024, 17)]
    o32, o16 = m32(*xs), m16(*xs)

    def ccc(x, y):
        x, y = x.double(), y.double()
        mx, my = x.mean(0), y.mean(0)
        vx, vy = x.var(0, unbiased=False), y.var(0, unbiased=False)
        cov = ((x - mx) * (y - my)).mean(0)
        return 2 * cov / (vx + vy + (mx - my) ** 2)

    for k, floor in (("mu_all", 0.999), ("uncertainty_all", 0.99), ("calibrated_uncertainty", 0.99), ("fused_features", 0.995)):
        c = ccc(o16[k], o32[k])
        assert float(c.min()) > floor, (k, c.cpu().numpy())
    assert float((o16["attention_weights"] - o32["attention_weights"]).abs().max()) < 0.05


# ------------------------------------------------------------------------------------------- training
def _train_model(compute="fp32", tag="stackb", **cfg):
    m = stackb.CompleteDEERModel(stackb.ModelConfig(**cfg), compute_dtype=compute)
    P = {k: torch.from_numpy(v) for k, v in synth.module_fill(tag, _shapes()).items()}
    m.load_state_dict(P)
    return m.to("cuda:0"), P


def _batch_dev(b):
    return [torch.from_numpy(b[k]).cuda() for k in ("audio", "video", "text")], torch.from_numpy(b["targets"]).cuda()


def test_stackb_gradients_match_reference_golden_and_oracle():
    """fp32 backward through every layer: golden digests captured from the reference (eval mode = no dropout) and the
    oracle's autograd on the full tensors, <= 2e-3 of each gradient's largest element."""
    from tests.test_oracle_golden import check_gradient_digests, stackb_oracle_gradients
    g = np.load(os.path.join(GOLDEN, "stackb_B9.npz"))
    m, P = _train_model("fp32")
    b = synth.make_batch(9, seed=78)
    xs, y = _batch_dev(b)
    out = m.forward_train(*xs, dropout=False)
    loss = m.compute_loss(out, y)
    assert float(loss["total_loss"]) == pytest.approx(float(g["grad.total_loss"]), rel=2e-4)
    loss["total_loss"].backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    check_gradient_digests(g, grads, rtol=2e-3, atol_frac=2e-3)
    _, og = stackb_oracle_gradients(_oracle(), P, b)
    for n, gr in og.items():
        if gr is None:
            assert grads[n] is None, n
            continue
        ref = gr.numpy()
        np.testing.assert_allclose(grads[n].cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * max(float(np.abs(ref).max()), 1e-12), err_msg=n)


@pytest.mark.parametrize("B", [1, 4, 515])
def test_stackb_gradients_other_parameters_and_ragged_batches(B):
    from tests.test_oracle_golden import stackb_oracle_gradients
    m, P = _train_model("fp32", tag="stackb2")
    b = synth.make_batch(B, seed=31 + B)
    xs, y = _batch_dev(b)
    loss = m.compute_loss(m.forward_train(*xs, dropout=False), y)
    loss["total_loss"].backward()
    ol, og = stackb_oracle_gradients(_oracle(), P, b)
    assert float(loss["total_loss"]) == pytest.approx(float(ol), rel=3e-4)
    for n, p in m.named_parameters():
        if og[n] is None:
            assert p.grad is None, n
            continue
        ref = og[n].numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=3e-3, atol=3e-3 * max(float(np.abs(ref).max()), 1e-12), err_msg=f"{n} B={B}")


def test_stackb_bf16_gradients_track_fp32():
    """bf16 storage / fp32 accumulation: every gradient within a few bf16 roundings of the fp32 one (cosine + norm)."""
    b = synth.make_batch(256, seed=3)
    res = {}
    for compute in ("fp32", "bf16"):
        m, _ = _train_model(compute)
        xs, y = _batch_dev(b)
        m.compute_loss(m.forward_train(*xs, dropout=False), y)["total_loss"].backward()
        res[compute] = {n: p.grad.double().flatten() for n, p in m.named_parameters() if p.grad is not None}
    for n, g32 in res["fp32"].items():
        gb = res["bf16"][n]
        if float(g32.norm()) == 0.0:
            assert float(gb.norm()) == 0.0, n
            continue
        cos = float((g32 @ gb) / (g32.norm() * gb.norm()))
        ratio = float(gb.norm() / g32.norm())
        if g32.numel() < 16:        # 3-element batch sums of signed terms that nearly cancel: direction only
            assert cos > 0.9, (n, cos)
            continue
        assert cos > 0.95 and 0.9 < ratio < 1.1, (n, cos, ratio)


def test_stackb_training_mode_dropout_and_optimizer_steps():
    """.train(): dropout is live (two forwards differ, same step counter reproduces), p = 0 reduces to the eval forward,
    gradients are finite and AdamW + clip (complete_project.py:640-650, training.py:219-224) lowers the loss."""
    m, _ = _train_model("fp32")
    b = synth.make_batch(64, seed=9)
    xs, y = _batch_dev(b)
    m.train()
    o1 = m(*xs)["mu_all"].detach().clone()
    o2 = m(*xs)["mu_all"].detach().clone()
    assert not torch.equal(o1, o2)
    m._train_step = 0
    assert torch.equal(m(*xs)["mu_all"].detach(), o1)          # the mask is a function of (seed, step, site, row, column)
    ev = m.eval()(*xs)["mu_all"]
    assert float((o1 - ev).abs().max()) > 1e-3
    # dropout rate of one site, from the zeros of a ReLU-free dropout output is not observable here; check p = 0 instead
    m0, _ = _train_model("fp32", dropout=0.0)
    m0.train()
    # estimator dropout stays at 0.2 (complete_project.py:186), so modality_uncertainties differ from eval but only through it
    o = m0(*xs)
    assert not torch.equal(o["modality_uncertainties"], m0.eval()(*xs)["modality_uncertainties"])
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    first = last = None
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss = m.compute_loss(m(*xs), y)
        loss["total_loss"].backward()
        for p in m.parameters():
            assert p.grad is None or bool(torch.isfinite(p.grad).all())
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        last = float(loss["total_loss"])
        first = last if first is None else first
    assert last < first


def test_stackb_dropout_backward_matches_finite_differences():
    """With dropout live the backward must use the SAME masks as the forward: directional derivative of the loss along a
    random parameter direction (masks frozen by resetting the step counter) vs the analytic gradient, fp32."""
    m, _ = _train_model("fp32", tag="stackb2")
    b = synth.make_batch(33, seed=12)
    xs, y = _batch_dev(b)
    m.train()

    def loss_at():
        m._train_step = 5
        m.mark_parameters_changed()
        return m.compute_loss(m(*xs), y)["total_loss"]

    L = loss_at()
    L.backward()
    gen = torch.Generator(device="cuda").manual_seed(0)
    names = ["audio_encoder.encoder_layers.1.layers.0.weight", "attention_module.self_attention.value_proj.weight",
             "attention_module.uncertainty_estimator.estimator.0.weight", "attention_module.weight_network.0.weight",
             "fusion_module.av_fusion.0.weight", "fusion_module.fusion_gate.0.weight", "prediction_heads.arousal.evidence_network.3.weight",
             "text_encoder.input_projection.0.weight", "attention_module.cross_attention.output_proj.bias"]
    params = dict(m.named_parameters())
    for n in names:
        p = params[n]
        d = torch.randn(p.shape, generator=gen, device="cuda")
        d = d / d.norm()
        ana = float((p.grad * d).sum())
        eps = 2e-3
        with torch.no_grad():
            p.add_(eps * d); lp = float(loss_at()); p.sub_(2 * eps * d); lm = float(loss_at()); p.add_(eps * d)
        num = (lp - lm) / (2 * eps)
        assert num == pytest.approx(ana, rel=0.08, abs=2e-3 * max(1.0, float(p.grad.norm()))), (n, num, ana)


def test_stackb_captured_train_step_equals_eager_and_draws_fresh_masks():
    """capture_train_step: forward + loss + backward (+ clip + AdamW) as one HIP graph.  Replays are bit-identical to eager steps
    taken at the same dropout step, successive replays use different masks, and a captured optimiser trains."""
    import copy
    m1, _ = _train_model("bf16")
    m2 = copy.deepcopy(m1)
    b = synth.make_batch(128, seed=21)
    xs, y = _batch_dev(b)
    replay = m1.capture_train_step(*xs, y)
    losses = []
    for r in range(1, 4):
        ld = replay()
        torch.cuda.synchronize()
        losses.append(float(ld["total_loss"]))
        g1 = {n: p.grad.clone() for n, p in m1.named_parameters() if p.grad is not None}
        # the eager twin at the same dropout step: the replay used the step the device counter now holds
        m2.train()
        assert int(replay.counter.item()) == m1._train_step - 1
        m2._train_step = int(replay.counter.item())
        for p in m2.parameters():
            p.grad = None
        l2 = m2.compute_loss(m2(*xs), y)
        l2["total_loss"].backward()
        assert float(l2["total_loss"]) == losses[-1], r
        for n, p in m2.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, g1[n]), (r, n)
    assert len(set(losses)) == 3                          # fresh masks on every replay
    # new inputs through the static buffers
    b2 = synth.make_batch(128, seed=22)
    xs2, y2 = _batch_dev(b2)
    assert float(replay(*xs2, y2)["total_loss"]) != losses[-1]
    # replay -> eager step (e.g. a ragged tail batch) -> replay must draw three DIFFERENT masks: the steps they hash
    steps = []
    replay(*xs, y); steps.append(int(replay.counter.item()))
    steps.append(m1._train_step)                             # what the eager forward below hashes
    m1.train(); l_e = m1.compute_loss(m1(*xs), y); l_e["total_loss"].backward()
    replay(*xs, y); steps.append(int(replay.counter.item()))
    torch.cuda.synchronize()
    assert steps[1] == steps[0] + 1 and steps[2] == steps[1] + 1, steps
    # optimiser inside the graph
    m3, _ = _train_model("bf16")
    opt = torch.optim.AdamW(m3.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
    rep = m3.capture_train_step(*xs, y, optimizer=opt, max_grad_norm=1.0)
    w0 = m3.fusion_module.fusion_gate[0].weight.detach().clone()
    ls = [float(rep()["total_loss"]) for _ in range(15)]
    assert ls[-1] < ls[0] and not torch.equal(w0, m3.fusion_module.fusion_gate[0].weight)


@pytest.mark.parametrize("compute,B", [("fp32", 96), ("bf16", 1024)])
def test_fused_train_step_equals_the_autograd_path(compute, B):
    """train_step_fused (flat buffers, grouped weight-gradient launches, no autograd) runs the same
    operator sequence as compute_loss(model(...)).backward(): same loss bit for bit (the forward is the same launches), the
    same gradients up to the summation order of the split-K weight-gradient slices; FlatAdamW = clip_grad_norm_ +
    torch.optim.AdamW with the trainer's parameter groups (training.py:121-150, 219-224)."""
    import copy

    from mmdeer.optim import FlatAdamW
    m1, _ = _train_model(compute)
    m2 = copy.deepcopy(m1)
    m2.train_plan = "ops"          # the launch-by-launch plan: the sequence autograd runs (the layer-chain plan has its own test below)
    b = synth.make_batch(B, seed=33)
    xs, y = _batch_dev(b)
    m1.train(); m2.train()
    m1._train_step = m2._train_step = 17
    l1 = m1.compute_loss(m1(*xs), y)
    l1["total_loss"].backward()
    l2 = m2.train_step_fused(*xs, y)
    torch.cuda.synchronize()
    assert float(l1["total_loss"]) == float(l2["total_loss"])
    assert torch.equal(l1["ece_bin_counts"], l2["ece_bin_counts"])
    g1 = {n: p.grad for n, p in m1.named_parameters()}
    seen = 0
    for n, p in m2.named_parameters():
        if g1[n] is None:                                   # the calibration layer: off the path
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        scale = max(float(g1[n].abs().max()), 1e-12)
        tol = 2e-5 if compute == "fp32" else 2e-4           # fp32 sums of the same products in another order
        assert float((p.grad - g1[n]).abs().max()) <= tol * scale + 1e-9, (n, float((p.grad - g1[n]).abs().max()), scale)
        seen += 1
    assert seen >= 100
    # optimiser: one step against torch's clip + AdamW on the autograd twin (encoder-named parameters at half the rate)
    enc = [p for n, p in m1.named_parameters() if "encoder" in n and p.grad is not None]
    rest = [p for n, p in m1.named_parameters() if "encoder" not in n and p.grad is not None]
    ref = torch.optim.AdamW([{"params": enc, "lr": 5e-4}, {"params": rest, "lr": 1e-3}], weight_decay=1e-2, eps=1e-8)
    for n, p in m1.named_parameters():                      # same gradients on both sides: isolate the update rule
        if p.grad is not None:
            p.grad.copy_(dict(m2.named_parameters())[n].grad)
    torch.nn.utils.clip_grad_norm_([p for p in m1.parameters() if p.grad is not None], 1.0)
    ref.step()
    opt = FlatAdamW(m2, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    opt.step()
    torch.cuda.synchronize()
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=2e-6, atol=2e-7), (n, float((p1 - p2).abs().max()))
    # the compute-dtype copy followed the update: the next fused step sees the new parameters without any cast
    m1._train_step = m2._train_step = 40
    for p in m1.parameters():
        p.grad = None
    l1b = m1.compute_loss(m1(*xs), y)
    l2b = m2.train_step_fused(*xs, y)
    assert float(l1b["total_loss"]) == pytest.approx(float(l2b["total_loss"]), rel=1e-5 if compute == "fp32" else 2e-3)
    assert float(l2b["total_loss"]) != float(l2["total_loss"])


@pytest.mark.parametrize("B", [100, 515, 1024, 4096])
def test_fused_train_step_layer_chains_match_the_launch_by_launch_plan(B):
    """the bf16 fused step runs its sample-local layer runs (residual encoders, attention value / output
    projections, estimator, fusion stages, evidence heads -- complete_project.py:60-118, 120-184, 307-418 -- and their dX runs) as
    launches of the layer-chain kernel (mmdeer_chain).  It is the launch-by-launch plan BIT FOR BIT, forward and backward: every tensor
    of the tape (activations, pre-LayerNorm rows, LayerNorm statistics, evidence), the loss, the ECE bin counts and every gradient.
    GEMM accumulation order, epilogues and dropout masks the chain kernel shares with the GEMM kernels; the LayerNorm forward /
    backward row arithmetic and the fold of the gamma / beta partials over 16-row groups are one statement (csrc/ln_rows.h) executed by
    the chain on its LDS panel and by the stand-alone bf16 kernels on rows in memory."""
    import copy
    m1, _ = _train_model("bf16")
    m2 = copy.deepcopy(m1)
    m1.train_plan, m2.train_plan = "ops", "auto"
    b = synth.make_batch(B, seed=35)
    xs, y = _batch_dev(b)
    m1.train(); m2.train()
    m1._train_step = m2._train_step = 23
    l1 = m1.train_step_fused(*xs, y)
    l2 = m2.train_step_fused(*xs, y)
    torch.cuda.synchronize()
    T1, T2 = l1["_keep"][0], l2["_keep"][0]
    assert not T1["chain"] and T2["chain"]
    assert float(l2["total_loss"]) == float(l1["total_loss"])
    assert torch.equal(l1["ece_bin_counts"], l2["ece_bin_counts"])
    for key in ("E", "VV", "S", "X", "H1", "H2", "pre", "AV", "T", "r", "w4", "u4", "R2", "G", "fused", "fused32", "H0", "H3", "ev", "planes"):
        assert torch.equal(T1[key], T2[key]), (key, int((T1[key] != T2[key]).sum()))
    for name in ("av", "tri"):
        for a, c in zip(T1[name], T2[name]):
            assert torch.equal(a, c), name
    for t1, t2 in zip(T1["enc"], T2["enc"]):
        for a, c in zip(t1["h"] + t1["y"] + [t1["y0"], t1["m0"], t1["r0"]], t2["h"] + t2["y"] + [t2["y0"], t2["m0"], t2["r0"]]):
            assert torch.equal(a, c)
        for (ma, ra), (mc, rc) in zip(t1["st"], t2["st"]):
            assert torch.equal(ma, mc) and torch.equal(ra, rc)
    seen = 0
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.isfinite(p2.grad).all(), n
        assert torch.equal(p1.grad, p2.grad), (n, float((p1.grad - p2.grad).abs().max()), float(p1.grad.abs().max()))
        seen += float(p1.grad.abs().max()) > 0
    assert seen >= 100


def test_fused_train_step_graph_replay_trains_and_matches_eager():
    import copy

    from mmdeer.optim import FlatAdamW
    m1, _ = _train_model("bf16")
    m2 = copy.deepcopy(m1)
    b = synth.make_batch(256, seed=34)
    xs, y = _batch_dev(b)
    rep = m1.capture_train_step_fused(*xs, y)
    m2.train()
    for r in range(3):                                      # replays are bit-identical to eager fused steps at the same dropout step
        lg = rep()
        torch.cuda.synchronize()
        m2._train_step = int(rep.counter.item())
        le = m2.train_step_fused(*xs, y)
        assert float(lg["total_loss"]) == float(le["total_loss"]), r
        for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(p1.grad, p2.grad), (r, n)
    opt = FlatAdamW(m1, lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
    losses = []
    for _ in range(15):
        losses.append(float(rep()["total_loss"]))
        opt.step()
    assert losses[-1] < losses[0] - 0.05 and all(np.isfinite(losses))
    # the trained parameters serve the inference path too (the operand image is rebuilt from the flat buffer)
    m1.eval()
    with torch.no_grad():
        out = m1(*xs)
    assert torch.isfinite(out["mu_all"]).all()
