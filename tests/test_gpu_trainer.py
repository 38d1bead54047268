"""BASELINE config 1 on the GPU: the launcher / trainer plumbing on synthetic data, B = 32."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from mmdeer.trainer import DEERTrainer, TrainingConfig, evaluate_deer_model, profile_training_speed  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def loaders(n, bs, seed, as_dict):
    b = synth.make_batch(n, seed=seed)
    if as_dict:   # multi_dataset_framework.py:88-98 batch format
        data = [{"audio_features": torch.from_numpy(b["audio"][i:i + bs]), "video_features": torch.from_numpy(b["video"][i:i + bs]),
                 "text_features": torch.from_numpy(b["text"][i:i + bs]), "targets": torch.from_numpy(b["targets"][i:i + bs])}
                for i in range(0, n, bs)]
        return {"iemocap": data}
    ds = torch.utils.data.TensorDataset(*(torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets")))
    return {"synthetic_train": torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False)}


@pytest.mark.parametrize("as_dict", [False, True])
def test_trainer_reduces_loss_and_reports_metrics(tmp_path, as_dict):
    model = MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.1, seed=3))
    cfg = TrainingConfig(learning_rate=3e-4, batch_size=32, num_epochs=4, output_dir=str(tmp_path / "out"), val_frequency=1,
                         save_frequency=2, log_dir=str(tmp_path / "log"), checkpoint_dir=str(tmp_path / "ckpt"))
    tr = DEERTrainer(model, cfg, "cuda:0")
    assert len(tr.optimizer.param_groups) >= 2                 # 'attention'-named group + default group
    hist = tr.train(loaders(256, 32, 1, as_dict), loaders(96, 32, 2, as_dict))
    json.dumps(hist)
    assert len(hist["train_loss"]) == 4 and hist["train_loss"][-1] < hist["train_loss"][0]
    assert all(np.isfinite(hist["grad_norm"]))
    ev = tr.evaluate_model(loaders(96, 32, 5, as_dict))
    for k in ("ccc_valence", "mae_arousal", "rmse_dominance", "ece", "ccc_overall", "test_loss"):
        assert np.isfinite(ev[k]), k
    ck = torch.load(tmp_path / "ckpt" / "best_model.pt", weights_only=False)
    assert {"model_state_dict", "training_config", "training_history", "training_time"} <= set(ck)     # run_multimodal_deer.py:512-517
    assert {"optimizer_state_dict", "scheduler_state_dict", "epoch", "loss"} <= set(ck)                  # what a resume needs
    assert sorted(f for f in os.listdir(tmp_path / "ckpt")) == ["best_model.pt", "checkpoint_epoch_0.pt", "checkpoint_epoch_2.pt", "final_model.pt"]
    m2 = MultimodalDEER(ModelConfig()).to("cuda:0")
    m2.load_state_dict(ck["model_state_dict"])
    assert np.isfinite(evaluate_deer_model(m2, loaders(64, 32, 5, as_dict), "cuda:0")["ccc_overall"])


def test_trainer_validation_frequency_patience_and_resume(tmp_path):
    """training.py:379-425 semantics: validation every val_frequency epochs, the best model by ccc_overall, patience counted
    in validations; no validation loaders -> no best model, no early stop, every epoch runs; load_checkpoint resumes the
    optimiser (moments, step) and the scheduler."""
    mk = lambda: MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.0, seed=3))
    cfg = dict(learning_rate=3e-4, batch_size=32, output_dir=str(tmp_path / "o"), log_dir=str(tmp_path / "l"))
    tr = DEERTrainer(mk(), TrainingConfig(num_epochs=6, val_frequency=2, save_frequency=100, patience=50,
                                          checkpoint_dir=str(tmp_path / "c1"), **cfg), "cuda:0")
    h = tr.train(loaders(64, 32, 1, False), loaders(64, 32, 2, False))
    assert len(h["val_loss"]) == 3 and len(h["learning_rate"]) == 6            # epochs 0, 2, 4 validated
    assert os.path.exists(tmp_path / "c1" / "best_model.pt")
    tr2 = DEERTrainer(mk(), TrainingConfig(num_epochs=5, patience=1, checkpoint_dir=str(tmp_path / "c2"), **cfg), "cuda:0")
    h2 = tr2.train(loaders(64, 32, 1, False), {})                               # no validation: all epochs, no best model
    assert len(h2["train_loss"]) == 5 and not os.path.exists(tmp_path / "c2" / "best_model.pt")
    assert os.path.exists(tmp_path / "c2" / "final_model.pt")
    # resume: optimiser step count, moments and the scheduler's epoch come back
    tr3 = DEERTrainer(mk(), TrainingConfig(num_epochs=5, checkpoint_dir=str(tmp_path / "c3"), **cfg), "cuda:0")
    ck = tr3.load_checkpoint(str(tmp_path / "c2" / "final_model.pt"))
    assert tr3.optimizer._t == tr2.optimizer._t and tr3.scheduler.last_epoch == tr2.scheduler.last_epoch
    assert torch.equal(tr3.optimizer._exp_avg.cpu(), tr2.optimizer._exp_avg.cpu()) and ck["epoch"] == 4
    for (n1, p1), (_, p2) in zip(tr3.model.named_parameters(), tr2.model.named_parameters()):
        assert torch.equal(p1, p2), n1
    # a finished run's checkpoint resumed with the same num_epochs has nothing left -- said aloud, not silently skipped;
    # loaded with resume=False (evaluation / fine-tuning from the weights) the next train() runs all its epochs
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        h3 = tr3.train(loaders(64, 32, 1, False), {})
    assert any("nothing left to train" in str(x.message) for x in w) and len(h3["train_loss"]) == len(ck["training_history"]["train_loss"])
    tr4 = DEERTrainer(mk(), TrainingConfig(num_epochs=2, checkpoint_dir=str(tmp_path / "c4"), **cfg), "cuda:0")
    tr4.load_checkpoint(str(tmp_path / "c2" / "final_model.pt"), resume=False)
    n0 = len(tr4.history["train_loss"])
    h4 = tr4.train(loaders(64, 32, 1, False), {})
    assert len(h4["train_loss"]) == n0 + 2


def test_trainer_resume_with_dropout_continues_the_mask_stream(tmp_path):
    """a checkpoint carries the dropout stream position (model._step) and train() continues after the saved
    epoch -- a run resumed from the epoch-1 checkpoint of a 4-epoch run ends bit for bit where the uninterrupted run ended,
    with dropout 0.3 live and without the test touching _step."""
    mk = lambda: MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.3, seed=3))
    cfg = dict(learning_rate=3e-4, batch_size=32, num_epochs=4, val_frequency=100, save_frequency=1, output_dir=str(tmp_path / "o"),
               log_dir=str(tmp_path / "l"))
    tr = DEERTrainer(mk(), TrainingConfig(checkpoint_dir=str(tmp_path / "c1"), **cfg), "cuda:0")
    tr.train(loaders(64, 32, 1, False), {})
    ck_path = str(tmp_path / "c1" / "checkpoint_epoch_1.pt")
    ck = torch.load(ck_path, weights_only=False)
    assert ck["dropout_state"]["_step"] == 4 and ck["epoch"] == 1          # 2 epochs x 2 batches
    tr2 = DEERTrainer(mk(), TrainingConfig(checkpoint_dir=str(tmp_path / "c2"), **cfg), "cuda:0")
    tr2.load_checkpoint(ck_path)
    assert tr2.model._step == 4
    h2 = tr2.train(loaders(64, 32, 1, False), {})
    assert tr2.current_epoch == 3 and tr2.model._step == tr.model._step == 8
    assert len(h2["learning_rate"]) == 4                                   # 2 from the checkpoint's history + 2 resumed epochs
    for (n1, p1), (_, p2) in zip(tr.model.named_parameters(), tr2.model.named_parameters()):
        assert torch.equal(p1, p2), n1


def test_clip_matches_torch_clip_grad_norm():
    model = MultimodalDEER(ModelConfig(dropout=0.0)).to("cuda:0").train()
    b = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.make_batch(16, seed=4).items()}
    tr = DEERTrainer(model, TrainingConfig(gradient_clip=0.05, output_dir="/tmp/mmdeer_t/o", log_dir="/tmp/mmdeer_t/l",
                                           checkpoint_dir="/tmp/mmdeer_t/c"), "cuda:0")
    model.train_step(b["audio"], b["video"], b["text"], b["targets"])
    ref = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    total_ref = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1e9)
    total = tr.clip_gradients()
    assert float(total) == pytest.approx(float(total_ref), rel=1e-5)
    scale = min(1.0, 0.05 / (float(total_ref) + 1e-6))
    for g0, p in zip(ref, [p for p in model.parameters() if p.grad is not None]):
        assert torch.allclose(p.grad, g0 * scale, rtol=1e-5, atol=1e-9)


def test_profile_training_speed_runs():
    model = MultimodalDEER(ModelConfig(compute_dtype="bf16")).to("cuda:0")
    r = profile_training_speed(model, batch_size=256, warmup=2, iters=5)
    assert r["forward"]["samples_per_sec"] > 0 and r["forward_backward"]["samples_per_sec"] > 0


def test_launcher_quick_run(tmp_path):
    """`run_multimodal_deer.py --mode full --quick --batch_size 32` (BASELINE configs[0])."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "run_multimodal_deer.py"), "--mode", "full", "--quick",
                        "--batch_size", "32", "--epochs", "2", "--output_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = [d for d in os.listdir(tmp_path) if d.startswith("experiment_")]
    rep = json.load(open(tmp_path / exp[0] / "report.json"))
    # the reference's defaults (training.py:64, 379): validation -- and a history row -- every val_frequency = 5 epochs
    assert len(rep["history"]["train_loss"]) == 1 and len(rep["history"]["learning_rate"]) == 2
    assert rep["sample_predictions"]["nig_keys"] == ["gamma", "nu", "alpha", "beta"]
    assert np.array(rep["sample_predictions"]["predictions"]).shape == (4, 3)


def test_launcher_stack_b(tmp_path):
    """`--stack b`: the script's CompleteDEERModel resolved to complete_project's class -- trainer, validation,
    evaluation, checkpoint and report over the operator-sequence training path."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "run_multimodal_deer.py"), "--mode", "full", "--quick", "--stack", "b",
                        "--batch_size", "32", "--epochs", "5", "--learning_rate", "1e-3", "--output_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = [d for d in os.listdir(tmp_path) if d.startswith("experiment_")]
    rep = json.load(open(tmp_path / exp[0] / "report.json"))
    assert len(rep["history"]["train_loss"]) == 1 and len(rep["history"]["learning_rate"]) == 5
    assert np.isfinite(rep["history"]["train_loss"][0]) and np.isfinite(rep["history"]["val_loss"][0])
    assert np.isfinite(rep["evaluation"]["test_loss"]) and "ccc_overall" in rep["evaluation"]
    assert rep["sample_predictions"]["nig_keys"] == ["valence_mu", "valence_nu"]
    ck = torch.load(tmp_path / exp[0] / "models" / "final_model.pt", map_location="cpu", weights_only=False)
    from mmdeer import stackb
    m = stackb.CompleteDEERModel()
    m.load_state_dict(ck["model_state_dict"])


def test_fused_adamw_matches_torch_adamw_with_clipping():
    """optim.FusedAdamW (mmdeer_adamw_step) against clip_grad_norm_ + torch.optim.AdamW on the same gradients, three
    steps, two learning-rate groups; and the packed weights it leaves behind drive the next forward."""
    import copy

    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.optim import FusedAdamW

    torch.manual_seed(0)
    m1 = MultimodalDEER(ModelConfig(compute_dtype="fp32", seed=3)).to("cuda:0").train()
    m2 = copy.deepcopy(m1)
    b = synth.make_batch(48, seed=5)
    a, v, t, y = (torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text", "targets"))

    def groups(m):
        enc = [p for n, p in m.named_parameters() if "projection" in n]
        rest = [p for n, p in m.named_parameters() if "projection" not in n]
        return [{"params": enc, "lr": 5e-4}, {"params": rest, "lr": 1e-3}]

    o1 = FusedAdamW(m1, groups(m1), weight_decay=0.01, eps=1e-8, max_grad_norm=0.5)
    o2 = torch.optim.AdamW(groups(m2), weight_decay=0.01, eps=1e-8)
    def worst_diff():
        return max(((p1 - p2).abs().max().item(), n) for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()))

    for step in range(3):
        m1._step = m2._step = 10 + step          # identical dropout masks
        l1 = m1.train_step(a, v, t, y)
        l2 = m2.train_step(a, v, t, y)
        assert float(l1["total_loss"]) == pytest.approx(float(l2["total_loss"]), rel=1e-5)
        n1 = o1.step()
        n2 = torch.nn.utils.clip_grad_norm_([p for p in m2.parameters() if p.grad is not None], 0.5)
        o2.step()
        assert float(n1) == pytest.approx(float(n2), rel=1e-5)
        if step == 0:
            # identical gradients in: the two updates may differ by rounding only
            w = worst_diff()
            assert w[0] < 2.5e-7, w          # one ulp at |p| ~ 1 (LayerNorm weights)
    # later steps: Adam normalises every element's update to ~lr, so elements whose gradient is rounding noise
    # (dead units) legitimately diverge by a fraction of lr; everything else stays together
    diffs = torch.cat([(p1 - p2).abs().flatten() for p1, p2 in zip(m1.parameters(), m2.parameters())])
    assert (diffs > 2e-6).float().mean().item() < 1e-3
    assert diffs.max().item() < 3e-3
    # the fused step refreshed the packed copies: an eval forward through each model agrees
    m1.eval(); m2.eval()
    out1, out2 = m1(a, v, t), m2(a, v, t)
    assert torch.allclose(out1["mu_all"], out2["mu_all"], rtol=1e-4, atol=1e-5)


def test_fused_adamw_writes_every_weight_image_itself():
    """in bf16 mode the optimiser step's update kernel writes every derived weight image (row-major packed copy,
    fragment-major W / W^T, row-major W^T, head-major in_proj, padded audio projection) tile by tile -- byte for byte what the
    element-wise update followed by the repack launch (option adam_fused = 0) leaves in the weights buffer, with the same
    parameters and moments."""
    import copy

    from mmdeer import _lib
    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.optim import FusedAdamW

    m1 = MultimodalDEER(ModelConfig(compute_dtype="bf16", seed=4)).to("cuda:0").train()
    m2 = copy.deepcopy(m1)
    b = synth.make_batch(640, seed=6)
    a, v, t, y = (torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text", "targets"))
    o1 = FusedAdamW(m1, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    o2 = FusedAdamW(m2, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    for m in (m1, m2):                    # alignment gaps of the buffer are never written: give them the same bytes
        m._weights(torch.device("cuda:0")).zero_()
        m._st.packed_key = None           # ... and pack again at the first step
    for step in range(2):
        m1._step = m2._step = 20 + step
        l1 = m1.train_step(a, v, t, y)
        l2 = m2.train_step(a, v, t, y)
        assert float(l1["total_loss"]) == float(l2["total_loss"])
        with _lib.options(adam_fused=1):
            o1.step()
        with _lib.options(adam_fused=0):
            o2.step()
        torch.cuda.synchronize()
        for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(p1, p2), (step, n)
        w1, w2 = m1._weights(torch.device("cuda:0")), m2._weights(torch.device("cuda:0"))
        lib = _lib.load()
        for name in (b"wpack", b"wtpack", b"vpack", b"wa_pad", b"wqkv_hm", b"wfpack", b"wtfpack"):
            lo = lib.mmdeer_weights_offset(0, name)
            assert lo >= 0
        assert torch.equal(w1, w2), (step, int((w1 != w2).sum()))


@pytest.mark.parametrize("n", [96, 100])
def test_trainer_graph_mode_matches_eager_mode(tmp_path, n):
    """DEERTrainer with use_graph=True (captured train_step + FusedAdamW) follows the same trajectory as the eager
    trainer: same data, same seeds, same dropout counters.  n = 100 leaves a ragged tail batch of 4 that runs eagerly
    on its own workspace between the replays -- the graph's weight copies must be refreshed after it."""
    import copy

    from torch.utils.data import DataLoader, TensorDataset

    b = synth.make_batch(n, seed=21)
    ds = TensorDataset(*(torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets")))
    m1 = MultimodalDEER(ModelConfig(compute_dtype="fp32", seed=4)).to("cuda:0")
    m2 = copy.deepcopy(m1)
    hist = []
    for m, graph in ((m1, False), (m2, True)):
        cfg = TrainingConfig(batch_size=32, num_epochs=1, output_dir=str(tmp_path / f"o{graph}"), log_dir=str(tmp_path / f"l{graph}"),
                             checkpoint_dir=str(tmp_path / f"c{graph}"), use_graph=graph)
        tr = DEERTrainer(m, cfg, device="cuda:0")
        loaders = {"iemocap": DataLoader(ds, batch_size=32, shuffle=False)}
        h = []
        for _ in range(3):                 # a validation pass on the training batch size (the graph's own workspace) in between
            h.append(tr.train_epoch(loaders)["total_loss"])
            h.append(tr.validate_epoch(loaders)["val_loss"])
        hist.append(h)
    assert hist[0] == pytest.approx(hist[1], rel=1e-5)
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=1e-4, atol=1e-6), n


@pytest.mark.parametrize("n", [96, 100])
def test_trainer_graph_mode_matches_eager_mode_stack_b(tmp_path, n):
    """DEERTrainer on Stack B (complete_project.CompleteDEERModel, the model the reference script trains) with use_graph=True: the fused
    step replayed as a captured HIP graph + FlatAdamW follows the eager trainer's trajectory -- same data, same dropout steps (the
    capture's one eager warm-up IS the first batch's step), same kernels.  n = 100 leaves a ragged tail batch that runs eagerly between
    the replays."""
    import copy

    from torch.utils.data import DataLoader, TensorDataset

    from mmdeer import stackb

    b = synth.make_batch(n, seed=22)
    ds = TensorDataset(*(torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets")))
    m1 = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype="bf16").to("cuda:0")
    m2 = copy.deepcopy(m1)
    hist = []
    for m, graph in ((m1, False), (m2, True)):
        cfg = TrainingConfig(batch_size=32, num_epochs=1, output_dir=str(tmp_path / f"o{graph}"), log_dir=str(tmp_path / f"l{graph}"),
                             checkpoint_dir=str(tmp_path / f"c{graph}"), use_graph=graph)
        tr = DEERTrainer(m, cfg, device="cuda:0")
        assert tr.fused_b
        loaders = {"iemocap": DataLoader(ds, batch_size=32, shuffle=False)}
        h = []
        for _ in range(3):
            h.append(tr.train_epoch(loaders)["total_loss"])
            h.append(tr.validate_epoch(loaders)["val_loss"])
        hist.append(h)
        if graph:
            assert len(tr._graphs_b) == 1
    assert hist[0] == pytest.approx(hist[1], rel=1e-6)
    for (name, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1, p2), name


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_graph_replay_with_fused_optimizer_trains(dtype):
    """capture_train_step + FusedAdamW as a training loop: the loss on a fixed batch goes down, every replay draws a
    new dropout mask, and an eval forward afterwards sees the updated (optimiser-packed) weights."""
    from mmdeer.optim import FusedAdamW

    m = MultimodalDEER(ModelConfig(compute_dtype=dtype, seed=2)).to("cuda:0").train()
    b = synth.make_batch(128, seed=33)
    a, v, t, y = (torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text", "targets"))
    if dtype == "bf16":
        a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
    opt = FusedAdamW(m, lr=2e-3, weight_decay=1e-5, max_grad_norm=1.0)
    replay = m.capture_train_step(a, v, t, y)
    opt.step()                                  # the capture's warm-up step counts as step 1
    losses = [float(replay.first["total_loss"])]
    for _ in range(30):
        d = replay()
        opt.step()
        losses.append(float(d["total_loss"]))
    assert all(l == l for l in losses)
    assert sum(losses[-5:]) / 5 < sum(losses[:5]) / 5 - 0.02, losses
    m.eval()
    with torch.no_grad():
        out = m(a, v, t)
    assert torch.isfinite(out["mu_all"]).all()
    # the eval forward used the packs the optimiser wrote: it matches a model rebuilt from the updated state_dict
    m2 = MultimodalDEER(ModelConfig(compute_dtype=dtype, seed=99)).to("cuda:0").eval()
    m2.load_state_dict(m.state_dict())
    with torch.no_grad():
        out2 = m2(a, v, t)
    assert torch.allclose(out["mu_all"], out2["mu_all"], rtol=1e-5, atol=1e-6)


def test_streaming_metrics_match_host_metrics():
    """metrics.StreamingMetrics (mmdeer_eval_accumulate, fp64 sums on the device) against the numpy formulas of the
    reference's metrics module, over ragged batches, with NaN rows and a constant column."""
    from mmdeer.metrics import StreamingMetrics, validation_metrics

    rng = np.random.default_rng(5)
    n = 3000
    tgt = np.tanh(rng.normal(size=(n, 3))).astype(np.float32)
    pred = (0.8 * tgt + 0.3 * rng.normal(size=(n, 3)) + np.array([0.05, -0.1, 0.0])).astype(np.float32)
    unc = np.abs(rng.normal(size=(n, 3))).astype(np.float32) * 0.5
    pred[17, 1] = np.nan                      # masked by the reference's _clean()
    sm = StreamingMetrics("cuda:0")
    i = 0
    for bs in (1, 255, 256, 1000, 7, n):      # ragged batches
        j = min(n, i + bs)
        if j > i:
            sm.update(*(torch.from_numpy(x[i:j]).to("cuda:0") for x in (pred, tgt, unc)))
        i = j
    got = sm.compute()
    ref = validation_metrics(pred, tgt, None)
    for k, v in ref.items():
        if k != "ece":
            assert got[k] == pytest.approx(v, rel=1e-6, abs=1e-9), k
    # calibration error: same per-sample reduction as the reference (NaN-free copy: np.quantile cannot take NaN)
    pred2 = pred.copy(); pred2[17, 1] = 0.0
    sm2 = StreamingMetrics("cuda:0")
    sm2.update(*(torch.from_numpy(x).to("cuda:0") for x in (pred2, tgt, unc)))
    assert sm2.compute()["ece"] == pytest.approx(validation_metrics(pred2, tgt, unc)["ece"], rel=1e-5)
    # a constant prediction column has no correlation: CCC = 0 as in the reference
    pred3 = pred2.copy(); pred3[:, 2] = 0.25
    sm3 = StreamingMetrics("cuda:0")
    sm3.update(*(torch.from_numpy(x).to("cuda:0") for x in (pred3, tgt, unc)))
    assert sm3.compute()["ccc_dominance"] == 0.0


def test_evaluate_deer_model_accepts_stack_b():
    """evaluation.evaluate_deer_model over loaders for the Stack B model: streaming metrics equal the host metrics of
    the concatenated predictions; the loss is stackb.CompleteDEERModel.compute_loss (MultiTaskDEERLoss on its keys), the mean of
    the per-batch values as training.py:285-299 accumulates them."""
    from mmdeer import stackb
    from mmdeer.metrics import validation_metrics
    m = stackb.CompleteDEERModel().to("cuda:0")
    ld = loaders(96, 32, 9, True)
    ev = evaluate_deer_model(m, ld, "cuda:0")
    assert not m.training
    preds, tgts, uncs, losses = [], [], [], []
    for b in ld["iemocap"]:
        out = m(b["audio_features"].cuda(), b["video_features"].cuda(), b["text_features"].cuda())
        losses.append(float(m.compute_loss(out, b["targets"].cuda())["total_loss"]))
        p, u = m.get_predictions_and_uncertainties(out)
        preds.append(p.cpu().numpy()); tgts.append(b["targets"].numpy()); uncs.append(u.cpu().numpy())
    ref = validation_metrics(np.concatenate(preds), np.concatenate(tgts), np.concatenate(uncs))
    for k in ("ccc_valence", "ccc_overall", "mae_arousal", "rmse_dominance"):
        np.testing.assert_allclose(ev[k], ref[k], rtol=1e-5, atol=1e-6, err_msg=k)
    assert ev["test_loss"] == pytest.approx(np.mean(losses), rel=1e-6) and np.isfinite(ev["test_loss"])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_alternating_batch_sizes_keep_the_packed_weights_current(dtype):
    """The packed weight copies live in the per-batch-size workspace and the fused optimiser refreshes only the one the
    last step used: a ragged tail batch, a validation pass in between and the return to the full batch size must all
    see current weights.  Reference run: the same sequence with a forced repack before every call."""
    import copy

    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.optim import FusedAdamW

    m1 = MultimodalDEER(ModelConfig(compute_dtype=dtype, dropout=0.3, seed=4)).to("cuda:0").train()
    m2 = copy.deepcopy(m1)
    o1, o2 = FusedAdamW(m1, lr=1e-3, max_grad_norm=1.0), FusedAdamW(m2, lr=1e-3, max_grad_norm=1.0)

    def data(B, seed):
        b = synth.make_batch(B, seed=seed)
        return [torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text", "targets")]

    plan = [("train", 32, 1), ("train", 8, 2), ("eval", 20, 3), ("train", 32, 4), ("eval", 32, 5), ("train", 8, 6), ("train", 32, 7)]
    for kind, B, seed in plan:
        a, v, t, y = data(B, seed)
        outs = []
        for m, o, force in ((m1, o1, False), (m2, o2, True)):
            if force:
                m._st.packed_key = None                      # reference: repack from the fp32 parameters every time
            if kind == "train":
                m.train()
                m._step = 100 + seed                         # identical dropout masks in both runs
                ld = m.train_step(a, v, t, y)
                o.step()
                outs.append(ld["total_loss"].clone())
            else:
                m.eval()
                with torch.no_grad():
                    outs.append(m(a, v, t)["mu_all"].clone())
        assert torch.equal(outs[0], outs[1]), (kind, B, seed)
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1, p2), n


def test_fused_adamw_checkpoint_round_trip_resumes_bit_for_bit():
    """FusedAdamW.state_dict() / load_state_dict() (flat moments, step count, group settings): a run resumed from a
    checkpoint in a fresh model + optimiser continues exactly like the run that was never interrupted."""
    import copy
    import io

    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.optim import FusedAdamW

    def data(seed):
        b = synth.make_batch(24, seed=seed)
        return [torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text", "targets")]

    def step(m, o, seed):
        m._step = 50 + seed
        ld = m.train_step(*data(seed))
        o.step()
        return float(ld["total_loss"])

    m = MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.2, seed=6)).to("cuda:0").train()
    o = FusedAdamW(m, lr=2e-3, weight_decay=0.01, max_grad_norm=1.0)
    for s in (1, 2, 3):
        step(m, o, s)
    buf = io.BytesIO()
    torch.save({"model": m.state_dict(), "optim": o.state_dict()}, buf)
    want = [step(m, o, s) for s in (4, 5)]

    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    m2 = MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.2, seed=99)).to("cuda:0").train()
    m2.load_state_dict(ck["model"])
    o2 = FusedAdamW(m2, lr=2e-3, weight_decay=0.01, max_grad_norm=1.0)
    o2.load_state_dict(ck["optim"])
    m2.config.seed = m.config.seed            # the dropout stream is keyed on (seed, step)
    got = [step(m2, o2, s) for s in (4, 5)]
    assert got == want
    for (n, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1, p2), n


@pytest.mark.parametrize("tag", ["main", "nan", "small", "ties"])
def test_streaming_metrics_match_the_reference_vectors(tag):
    """metrics.StreamingMetrics (mmdeer_eval_accumulate + the on-device quantile selection and bin sums of the calibration
    error: no per-sample array is copied to the host) against vectors captured from the imported reference's DEERMetrics
    and uncertainty_calibration_error, fed in ragged batches."""
    import os

    import numpy as np

    from mmdeer.metrics import StreamingMetrics, device_calibration_error

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_cases.npz"))
    p, t, u = (torch.from_numpy(g[f"{tag}.{k}"]).to("cuda:0") for k in ("predictions", "targets", "uncertainties"))
    sm = StreamingMetrics("cuda:0")
    n, lo = p.shape[0], 0
    for step in (5, 64, 1, 300, 10 ** 6):
        hi = min(n, lo + step)
        if hi > lo:
            sm.update(p[lo:hi], t[lo:hi], u[lo:hi])
        lo = hi
    out = sm.compute()
    for d in ("valence", "arousal", "dominance"):
        assert out[f"ccc_{d}"] == pytest.approx(float(g[f"{tag}.ccc_{d}"]), rel=1e-5, abs=1e-7), d
        assert out[f"mae_{d}"] == pytest.approx(float(g[f"{tag}.mae_{d}"]), rel=1e-5), d
        assert out[f"rmse_{d}"] == pytest.approx(float(g[f"{tag}.rmse_{d}"]), rel=1e-5), d
    assert out["ece"] == pytest.approx(float(g[f"{tag}.ece"]), rel=2e-5, abs=1e-7)
    err, unc = torch.cat(sm._err), torch.cat(sm._unc)
    assert err.is_cuda and unc.is_cuda                       # the per-sample arrays stayed on the device
    assert device_calibration_error(err, unc, n_bins=15) == pytest.approx(float(g[f"{tag}.ece_15"]), rel=2e-5, abs=1e-7)
