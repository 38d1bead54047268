"""BASELINE configs the round-1 suite did not run on hardware:

* configs[4]: B = 8192 per GPU, bf16, missing-modality batches (audio-only / text-only = zero-filled blocks, the only
  missing-modality mechanism of the reference: encoders.py:826-849, preprocessing.py:349,360) -- a full train step
  against the fp32 CPU oracle (CCC >= 0.999, |delta| <= 5e-2: SURVEY 7) plus the size-independent properties;
* configs[3] (data parallel) semantics on one device: two shards played as two ranks in the DEFAULT (DDP-mean) mode --
  the averaged shard gradients equal the oracle's mean of per-shard gradients; and the trainer's graph path in
  exact-global mode (the captured step must exchange the loss statistics like the eager one).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from mmdeer.spec import DIM_NAMES, param_table  # noqa: E402
from oracle import deer_oracle as O  # noqa: E402

DEV = "cuda:0"


def batch(B, seed=42, zero=()):
    b = synth.make_batch(B, seed=seed)
    for z in zero:
        b[z] = np.zeros_like(b[z])
    return {k: torch.from_numpy(v) for k, v in b.items()}


def oracle_params(model, dtype=torch.float32, requires_grad=False):
    return O.to_params({k: v.detach().cpu() for k, v in model.state_dict().items()}, dtype, requires_grad)


@pytest.mark.parametrize("tag,zero", [("audio_only", ("video", "text")), ("text_only", ("audio", "video"))])
def test_config5_B8192_bf16_missing_modality_train_step(tag, zero):
    B = 8192
    b = batch(B, seed=29, zero=zero)
    m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.0, seed=11)).to(DEV).train()
    a, v, t = (b[k].to(DEV).bfloat16() for k in ("audio", "video", "text"))     # configs[4]: bf16 feature blocks
    y = b["targets"].to(DEV)
    for z in zero:       # the mask is exact zeroing: the bf16 blocks of the missing modalities are all +0.0 bits
        assert int({"audio": a, "video": v, "text": t}[z].view(torch.int16).abs().max()) == 0
    ld = m.train_step(a, v, t, y)
    g1 = m.flat_grad().clone()
    nig = ld["_outputs"]["_nig"].cpu()
    # fp32 CPU oracle of the same step on the same (bf16-rounded) inputs
    P = oracle_params(m, torch.float32, requires_grad=True)
    fo, ho, ldo, grads = O.train_step(P, a.float().cpu(), v.float().cpu(), t.float().cpu(), b["targets"])
    ref = {k: torch.cat([ho[f"{d}_{k}"] for d in DIM_NAMES], dim=1).detach() for k in ("mu", "nu", "alpha", "beta")}
    for i, k in enumerate(("mu", "nu", "alpha", "beta")):
        assert (nig[i] - ref[k]).abs().max().item() < 5e-2, (tag, k)
        for d in range(3):
            if float(ref[k][:, d].std()) > 1e-3:
                assert O.ccc(nig[i][:, d], ref[k][:, d]) > 0.999, (tag, k, d)
    assert float(ld["total_loss"]) == pytest.approx(float(ldo["total_loss"]), rel=2e-2, abs=2e-3)
    # gradient direction on the matrices that receive signal from the present modality
    named = dict(m.named_parameters())
    checked = 0
    for name, shape, _ in param_table():
        if len(shape) != 2:
            continue
        rr = grads[name].double().flatten()
        if float(rr.norm()) < 1e-9:
            continue                       # e.g. the projections of the zeroed modalities: exact-zero gradients
        gg = named[name].grad.cpu().double().flatten()
        cos = float((gg @ rr) / (gg.norm() * rr.norm() + 1e-30))
        assert cos > 0.95, (tag, name, cos)
        checked += 1
    assert checked >= 10
    # a zeroed modality's input projection gets an exactly-zero weight gradient (dW = dY^T X with X == 0)
    zero_w = {"audio": "fusion.audio_visual_fusion.audio_projection.weight", "video": "fusion.audio_visual_fusion.video_projection.weight",
              "text": "fusion.trimodal_fusion.text_projection.weight"}
    for z in zero:
        assert float(named[zero_w[z]].grad.abs().max()) == 0.0, (tag, z)
    # size-independent properties at the full size: determinism (no atomics), bin populations, batch permutation
    m._step -= 1
    l2 = m.train_step(a, v, t, y)
    assert torch.equal(g1, m.flat_grad()) and float(l2["total_loss"]) == float(ld["total_loss"])
    assert int(ld["ece_bin_counts"].sum()) == 3 * B and torch.isfinite(g1).all()
    m.eval()
    with torch.no_grad():
        o1 = m(a, v, t)["mu_all"].clone()
        perm = torch.randperm(B, device=DEV)
        o2 = m(a[perm], v[perm], t[perm])["mu_all"]
    assert torch.equal(o1[perm], o2)


def test_config5_B8192_bf16_training_with_dropout_properties():
    """The dropout-on training step at the configs[4] size: finite, deterministic for a fixed step counter, a fresh mask
    per step, and the zero-filled modality still yields exact-zero gradients for its projection."""
    B = 8192
    b = batch(B, seed=31, zero=("video", "text"))
    m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=5)).to(DEV).train()
    a, v, t = (b[k].to(DEV).bfloat16() for k in ("audio", "video", "text"))
    y = b["targets"].to(DEV)
    s0 = m._step
    l1 = m.train_step(a, v, t, y)
    g1 = m.flat_grad().clone()
    m._step = s0
    l2 = m.train_step(a, v, t, y)
    assert torch.equal(g1, m.flat_grad()) and float(l1["total_loss"]) == float(l2["total_loss"])
    l3 = m.train_step(a, v, t, y)                       # next step: another mask
    assert not torch.equal(g1, m.flat_grad())
    assert all(np.isfinite(float(l["total_loss"])) for l in (l1, l3)) and torch.isfinite(m.flat_grad()).all()
    assert int(l3["ece_bin_counts"].sum()) == 3 * B
    named = dict(m.named_parameters())
    assert float(named["fusion.trimodal_fusion.text_projection.weight"].grad.abs().max()) == 0.0


def test_ddp_mean_of_two_shards_equals_the_oracle_mean_of_shard_gradients():
    """Default data-parallel semantics (DDP = mean over ranks of per-shard gradients): two ranks are played by
    two equal shards on the one GPU, their flat gradient buffers are averaged exactly as the AVG all-reduce of
    parallel.BucketedAllReduce would, and the result equals the oracle's mean of the two per-shard gradients -- and is
    NOT the gradient of the union batch (the ECE / cross-dimension terms are non-linear in batch statistics)."""
    n = 96
    b = batch(2 * n, seed=19)
    shards = [{k: v[i * n:(i + 1) * n] for k, v in b.items()} for i in range(2)]
    models = [MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.0, seed=3), init="closed_form").to(DEV).train() for _ in range(2)]
    losses = []
    for m, s in zip(models, shards):
        losses.append(float(m.train_step(*(s[k].to(DEV) for k in ("audio", "video", "text", "targets")))["total_loss"]))
    flat = torch.stack([m.flat_grad() for m in models]).mean(0)          # what every rank holds after all_reduce(AVG)
    P = oracle_params(models[0], requires_grad=True)
    refs, ref_losses = [], []
    for s in shards:
        for p in P.values():
            if p.grad is not None:
                p.grad = None
        _, _, ldo, grads = O.train_step(P, s["audio"], s["video"], s["text"], s["targets"])
        refs.append({k: g.clone() for k, g in grads.items()})
        ref_losses.append(float(ldo["total_loss"]))
    assert np.mean(losses) == pytest.approx(np.mean(ref_losses), rel=2e-5)
    _, _, ldu, gunion = O.train_step(P, b["audio"], b["video"], b["text"], b["targets"])
    off = dict(zip([nm for nm, _, _ in param_table()], models[0]._offsets))
    checked, differs = 0, 0
    for name, shape, _ in param_table():
        if name not in refs[0] or refs[0][name] is None:
            continue
        want = (refs[0][name].double() + refs[1][name].double()) / 2
        got = flat[off[name]:off[name] + want.numel()].view(want.shape).cpu().double()
        scale = max(want.abs().max().item(), 1e-12)
        assert (got - want).abs().max().item() / scale < 2e-3, name
        checked += 1
        if (gunion[name].double() - want).abs().max().item() / scale > 1e-3:
            differs += 1
    assert checked >= 30 and differs >= 1


def test_trainer_graph_step_in_exact_global_mode_equals_the_eager_step(tmp_path):
    """with use_graph=True and a communicator in exact-global mode the captured step must take the
    statistics-exchange path like the eager step (gradients = this rank's share of the global-batch gradient, exchanged
    with SUM).  A 1-rank stand-in communicator makes both paths comparable bit for bit."""
    import copy

    from mmdeer.trainer import DEERTrainer, TrainingConfig

    class OneRank:                       # the surface DEERTrainer / train_step use of parallel.BucketedAllReduce
        active, exact_global, events, world = True, True, None, 1

        def __init__(self):
            self.stats_calls = 0

        def sum_small(self, t):
            self.stats_calls += 1

        def launch(self, flat):
            pass

        def wait(self, flat=None):
            pass

    b = synth.make_batch(64, seed=23)
    data = [tuple(torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets"))]
    m1 = MultimodalDEER(ModelConfig(compute_dtype="fp32", dropout=0.0, seed=6)).to(DEV)
    m2 = copy.deepcopy(m1)
    grads, comms = [], []
    for m, graph in ((m1, False), (m2, True)):
        comm = OneRank()
        cfg = TrainingConfig(batch_size=64, num_epochs=1, output_dir=str(tmp_path / f"o{graph}"), log_dir=str(tmp_path / f"l{graph}"),
                             checkpoint_dir=str(tmp_path / f"c{graph}"), use_graph=graph)
        tr = DEERTrainer(m, cfg, device=DEV, comm=comm)
        tr.train_epoch({"iemocap": data})
        tr.train_epoch({"iemocap": data})          # second epoch: a replay in graph mode
        torch.cuda.synchronize()
        grads.append(m.flat_grad().clone())
        comms.append(comm)
    assert comms[0].stats_calls >= 2 and comms[1].stats_calls >= 1     # the capture recorded the exchange
    assert torch.allclose(grads[0], grads[1], rtol=1e-5, atol=1e-8)
    for (n_, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=1e-4, atol=1e-6), n_


def test_data_parallel_ranks_share_parameters_but_not_dropout_masks():
    """ranks build the model with the same seed (identical parameters) -- their dropout masks must still differ
    (ModelConfig.dropout_seed, set per rank by bench.py), while one rank's masks stay a pure function of (seed, step)."""
    from mmdeer import _lib
    lib = _lib.load()
    ranks = [MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=42, dropout_seed=42 + 1000003 * r)).to(DEV).train() for r in range(2)]
    for (n0, p0), (_, p1) in zip(ranks[0].named_parameters(), ranks[1].named_parameters()):
        assert torch.equal(p0, p1), n0
    b = batch(64, seed=2)
    a, v, t, y = (b[k].to(DEV) for k in ("audio", "video", "text", "targets"))
    masks = []
    for m in ranks:
        m.train_step(a.bfloat16(), v.bfloat16(), t.bfloat16(), y)
        mk = torch.empty(64, 512, dtype=torch.uint8, device=DEV)
        _lib.check(lib.mmdeer_dropout_mask(5, 64, 512, 0.3, m.dropout_seed, m._step, mk.data_ptr(), torch.cuda.current_stream().cuda_stream))
        masks.append(mk.clone())
    torch.cuda.synchronize()
    agree = float((masks[0] == masks[1]).float().mean())
    assert 0.5 < agree < 0.66            # independent Bernoulli(0.7) masks agree with probability 0.58
    assert ranks[0].dropout_seed != ranks[1].dropout_seed and not torch.equal(ranks[0].flat_grad(), ranks[1].flat_grad())
