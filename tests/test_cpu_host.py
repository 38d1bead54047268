"""CPU-only checks: the C-ABI library builds, loads and exports every symbol the header declares;
host-side logic (parameter table, state_dict names, synthetic generators, sharding, error paths).
No compute call is made: there is no GPU here."""
import json
import os
import re

import numpy as np
import pytest
import torch

from mmdeer import _lib, build, synth
from mmdeer.model import CompleteDEERModel, ModelConfig, MultimodalDEER, loss_dict_from
from mmdeer.parallel import shard_rows
from mmdeer.spec import gate_param_table, n_live_params, param_offsets, param_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mmdeer.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmdeer_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mmdeer.h but not exported"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert bound == set(declared), (bound ^ set(declared))
    assert lib.mmdeer_abi_version() == 15
    assert b"gfx950" in lib.mmdeer_version()


def test_ctypes_struct_layouts_match_the_library():
    """Every argument struct of include/mmdeer.h is mirrored in mmdeer/_lib.py: the sizes the compiler gave them (mmdeer_sizeof) are
    the sizes ctypes computes -- a field added on one side only fails here (and at load time) instead of shifting pointers."""
    import ctypes as C
    lib = _lib.load()
    assert len(_lib.STRUCTS) >= 13
    for name, cls in _lib.STRUCTS.items():
        assert lib.mmdeer_sizeof(name.encode()) == C.sizeof(cls), name
    assert lib.mmdeer_sizeof(b"no_such_struct") == -1


def test_layer_chain_operator_validates_its_table_on_the_host():
    """mmdeer_chain checks the whole segment table before it launches anything (no GPU is touched here: every case below is refused,
    or empty): unsupported widths, panels that do not fit, residual / stash options where they do not apply."""
    import ctypes as C
    lib = _lib.load()
    A = 1 << 20                                     # a plausible, aligned address: never dereferenced by the host checks
    a = _lib.ChainArgs()
    a.X, a.ldx, a.K0, a.rows, a.nseg = A, 256, 256, 0, 1
    assert lib.mmdeer_chain(C.byref(a)) == 0        # an empty batch is nothing to do
    a.rows, a.nseg = 64, 0
    assert lib.mmdeer_chain(C.byref(a)) != 0 and b"segments" in lib.mmdeer_last_error()
    a.nseg = 1
    s = a.seg[0]
    s.W, s.N, s.K, s.end_layer, s.nout, s.drop_site = A, 256, 320, 1, 256, -1
    assert lib.mmdeer_chain(C.byref(a)) != 0        # K = 320 reads past the 256-wide input panel
    s.K, s.N, s.nout = 256, 96, 96
    assert lib.mmdeer_chain(C.byref(a)) != 0        # N % 64
    s.N, s.nout, s.W = 256, 256, A + 2
    assert lib.mmdeer_chain(C.byref(a)) != 0        # misaligned weight image
    s.W, s.res_add = A, 1
    assert lib.mmdeer_chain(C.byref(a)) != 0 and b"bypass" in lib.mmdeer_last_error()
    s.res_add, s.stash, s.ld_stash, s.stash_split = 0, A, 256, 128
    assert lib.mmdeer_chain(C.byref(a)) != 0        # a split without the second tensor
    s.stash_split, a.samples_per_workgroup = 0, 24
    assert lib.mmdeer_chain(C.byref(a)) != 0 and b"samples_per_workgroup" in lib.mmdeer_last_error()
    a.samples_per_workgroup, a.K0 = 32, 768
    assert lib.mmdeer_chain(C.byref(a)) != 0        # a 768-wide input needs the 16-sample workgroups
    assert lib.mmdeer_chain_workgroups(4096, 0) == 256 and lib.mmdeer_chain_workgroups(4097, 0) == 129 and lib.mmdeer_chain_workgroups(100, 16) == 7
    j = (_lib.RepackJob * 1)()
    j[0].src, j[0].dst, j[0].ld_src, j[0].rows, j[0].cols, j[0].cols_valid, j[0].layout = A, A, 100, 100, 100, 100, 1
    assert lib.mmdeer_repack(j, 1, None) != 0       # a fragment-major image needs rows % 16 == 0 and columns % 64 == 0
    assert lib.mmdeer_repack(j, 0, None) == 0


def test_frag_images_job_table_on_the_host():
    """mmdeer/chainops.py: FragImages lays its images out in one buffer and describes them to mmdeer_repack -- offsets, shapes, sub-images
    and the refusals are host logic (no launch here)."""
    from mmdeer.chainops import FragImages
    cpu = torch.device("cpu")
    W = torch.zeros(256, 84, dtype=torch.bfloat16)
    V = torch.zeros(512, 256, dtype=torch.bfloat16)
    F = FragImages(cpu)
    F.add("w", W, 256, 128, ld_src=84, cols_valid=84)            # zero-padded to K = 128
    F.add("v", V, 512, 256)
    F.add("v.T", V, 512, 256, transpose=1)
    F.area("bd", 24, 384)
    F.place("bd", V[:4], 4, 128, row0=8, col0=128)
    with pytest.raises(ValueError):
        F.add("bad", W, 256, 84)                                 # columns of a fragment-major image: multiples of 64
    with pytest.raises(ValueError):
        F.place("bd", V[:4], 4, 128, row0=22)                    # past the area
    F.finish()
    assert F.off["w"] == 0 and F.off["v"] == 256 * 128 and all(o % 64 == 0 for o in F.off.values())
    assert F.shape["v.T"] == (256, 512) and F("v.T").numel() == 256 * 512
    assert F("v", 64).numel() == (512 - 64) * 256 and F("v", 64).data_ptr() == F("v").data_ptr() + 2 * 64 * 256
    assert F.mat("bd").shape == (24, 384)
    jobs = {name: F.jobs[i] for i, (name, *_) in enumerate(F.specs)}
    assert (jobs["w"].ld_src, jobs["w"].cols, jobs["w"].cols_valid, jobs["w"].layout, jobs["w"].transpose) == (84, 128, 84, 1, 0)
    assert (jobs["v.T"].transpose, jobs["v.T"].layout) == (1, 1)
    assert (jobs["bd"].layout, jobs["bd"].ld_dst, jobs["bd"].dst_col) == (0, 384, 128) and jobs["bd"].dst == F.buf.data_ptr() + 2 * (F.off["bd"] + 8 * 384)


def test_parameter_table_matches_python_spec():
    lib = _lib.load()
    offs, total = param_offsets()
    table = param_table()
    assert lib.mmdeer_num_params() == len(table) == 50
    assert lib.mmdeer_flat_elems() == total
    for i, ((name, shape, _), off) in enumerate(zip(table, offs)):
        assert lib.mmdeer_param_name(i).decode() == name
        assert lib.mmdeer_param_rows(i) == shape[0]
        assert lib.mmdeer_param_cols(i) == (shape[1] if len(shape) == 2 else 1)
        assert lib.mmdeer_param_offset(i) == off and off % 64 == 0
    assert n_live_params() == 2907212                      # live gradient payload
    # the three heads' layers are contiguous in the flat buffer (stacked / batched GEMMs rely on it)
    names = [n for n, _, _ in table]
    i0 = names.index("head.deer_heads.0.evidence_net.0.weight")
    assert offs[i0 + 1] - offs[i0] == 128 * 256 and offs[i0 + 2] - offs[i0 + 1] == 128 * 256
    # buckets tile the flat buffer in reverse execution order
    assert lib.mmdeer_bucket_begin(2) == 0 and lib.mmdeer_bucket_end(0) == total
    assert lib.mmdeer_bucket_end(2) == lib.mmdeer_bucket_begin(1) and lib.mmdeer_bucket_end(1) == lib.mmdeer_bucket_begin(0)


def test_workspace_bytes_grow_with_batch():
    lib = _lib.load()
    w0, w1, w2 = (lib.mmdeer_workspace_bytes(b, 0) for b in (0, 1024, 4096))
    assert 0 < w0 < w1 < w2
    assert lib.mmdeer_workspace_bytes(4096, 1) > w2
    assert w2 < 1 << 30


def test_state_dict_names_follow_the_reference(golden_dir):
    m = MultimodalDEER(ModelConfig())
    sd = m.state_dict()
    names = json.load(open(os.path.join(golden_dir, "state_dict_names.json")))
    # same key set (load_state_dict is keyed; the registration order of uncertainty_gate differs)
    assert sorted(k for k in sd if k.startswith("fusion.")) == sorted("fusion." + n for n in names["fusion"])
    assert sorted(k for k in sd if k.startswith("head.")) == sorted("head." + n for n in names["head"])
    for k, shape in names["shapes"].items():
        assert list(sd[k].shape) == shape, k
    assert sum(p.numel() for p in m.parameters()) == 3105711
    assert CompleteDEERModel is MultimodalDEER
    # reference init statistics: zero biases, unit LayerNorm, Xavier-bounded weights
    assert float(sd["fusion.output_projection.0.bias"].abs().max()) == 0.0
    assert torch.equal(sd["fusion.output_projection.3.weight"], torch.ones(512))
    w = sd["fusion.trimodal_fusion.text_projection.weight"]
    bound = (6.0 / (768 + 512)) ** 0.5
    assert float(w.abs().max()) <= bound and float(w.std()) == pytest.approx(bound / 3 ** 0.5, rel=0.02)
    fp = sd["head.feature_processor.0.weight"]                            # torch default init: U(+-1/sqrt(fan_in))
    assert float(fp.abs().max()) <= 1 / 512 ** 0.5 + 1e-7


def test_unsupported_geometry_and_cpu_inputs_fail_loudly():
    with pytest.raises(NotImplementedError):
        MultimodalDEER(ModelConfig(audio_dim=40))
    m = MultimodalDEER(ModelConfig()).eval()
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        m(torch.zeros(2, 84), torch.zeros(2, 256), torch.zeros(2, 768))
    with pytest.raises(ValueError):
        m({"audio": torch.zeros(2, 84)})


def test_loss_dict_layout():
    lo = torch.arange(20, dtype=torch.float32)
    d = loss_dict_from(lo, 5)
    assert float(d["total_loss"]) == 16 and float(d["cross_dim_loss"]) == 15
    assert float(d["arousal_kl_loss"]) == 8 and d["dominance_batch_size"] == 5
    for k in ("deer_loss", "nll_loss", "evidence_reg", "kl_reg"):       # keys training.py:187-190 accumulates
        assert k in d


def test_synth_is_counter_based_and_shardable():
    full = synth.make_batch(12, seed=3)
    a = synth.make_batch(5, seed=3, row_offset=0)
    b = synth.make_batch(7, seed=3, row_offset=5)
    for k in full:
        np.testing.assert_array_equal(full[k], np.concatenate([a[k], b[k]]))
    assert full["audio"].shape == (12, 84) and full["targets"].shape == (12, 3)
    assert np.abs(full["targets"]).max() < 1.0
    big = synth.make_batch(4096, seed=42)
    assert abs(float(big["text"].mean())) < 0.01 and abs(float(big["text"].std()) - 1.0) < 0.01
    s1, s2 = synth.closed_form_state(), synth.closed_form_state()
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k])
    assert len(synth.closed_form_state(include_gate=True)) == len(param_table()) + len(gate_param_table())


def test_shard_rows_partitions_the_batch():
    for total, world in ((4096, 8), (10, 3), (7, 8)):
        spans = [shard_rows(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
            assert e0 == b1 and e0 >= b0


def test_fused_adamw_argument_checks_without_a_gpu():
    """optim.FusedAdamW validates its configuration on the host and refuses to step without gradients of a
    train_step (no silent torch fallback)."""
    import torch

    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.optim import FusedAdamW

    m = MultimodalDEER(ModelConfig())
    with pytest.raises(ValueError):
        FusedAdamW(m, lr=-1.0)
    live = m.live_parameters()
    with pytest.raises(ValueError):
        FusedAdamW(m, [{"params": live[:3]}])                       # trainable parameters left out
    with pytest.raises(NotImplementedError):
        FusedAdamW(m, [{"params": live[:3], "weight_decay": 0.1}, {"params": live[3:]}])   # per-group decay
    opt = FusedAdamW(m, [{"params": live[:10], "lr": 5e-5}, {"params": live[10:]}], lr=1e-4)
    assert [g["lr"] for g in opt.param_groups] == [5e-5, 1e-4]
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)     # torch schedulers drive it unchanged
    with pytest.raises(RuntimeError):
        opt.step()
    sd = opt.state_dict()
    assert sd["step"] == 0 and sd["exp_avg"] is None
    del sched


def test_stackb_state_dict_matches_the_reference_names():
    """mmdeer.stackb.CompleteDEERModel exposes complete_project.CompleteDEERModel's state_dict keys and shapes
    (captured by tests/golden/make_golden.py), Xavier/zero/one initialisation, and has no CPU compute path."""
    import json
    import torch
    from mmdeer import stackb
    with open(os.path.join(os.path.dirname(__file__), "golden", "stackb_state_dict_names.json")) as fh:
        shapes = json.load(fh)
    m = stackb.create_complete_deer_model()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes
    sd = m.state_dict()
    assert float(sd["calibration_layer.temperature"].sum()) == 3.0
    assert float(sd["audio_encoder.input_projection.0.bias"].abs().sum()) == 0.0
    w = sd["fusion_module.av_fusion.0.weight"]
    assert float(w.abs().max()) <= (6.0 / (512 + 512)) ** 0.5 + 1e-6 and float(w.std()) > 0.02
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(torch.zeros(2, 84), torch.zeros(2, 256), torch.zeros(2, 768))


def test_loss_classes_keep_the_reference_protocol_and_refuse_cpu_tensors():
    """mmdeer.losses mirrors the constructor / call protocol of the reference's loss classes (deer.py:111-195,
    losses.py:40-601); without a GPU every numeric call fails loudly, the key-lookup fallbacks behave as in the reference."""
    import torch
    from mmdeer import losses
    assert isinstance(losses.create_deer_loss("basic", {"reg_weight": 0.2}), losses.DEERLoss)
    assert isinstance(losses.create_deer_loss("MultiTask"), losses.MultiTaskDEERLoss)
    comb = losses.create_deer_loss()
    assert isinstance(comb, losses.CombinedDEERLoss) and comb.use_calibration_loss and comb.calibration_loss.n_bins == 15
    with pytest.raises(ValueError, match="Unknown loss type"):
        losses.create_deer_loss("focal")
    with pytest.raises(NotImplementedError):
        losses.MultiTaskDEERLoss(emotion_dims=["valence"])
    pred = {k: torch.ones(4, 1) for k in ("mu", "nu", "alpha", "beta")}
    for fn in (losses.DEERLossV1(), losses.DEERLoss(), losses.UncertaintyRegularizationLoss(), losses.CalibrationLoss()):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn(pred, torch.zeros(4))
    # flat keys absent: the two extra terms return 0 exactly as the reference does (losses.py:379-380, 447-448)
    per_dim = {"valence_alpha": torch.ones(4, 1)}
    assert float(losses.UncertaintyRegularizationLoss()(per_dim, torch.zeros(4))["reg_loss"]) == 0.0
    assert float(losses.CalibrationLoss()(per_dim, torch.zeros(4))) == 0.0


def test_host_metrics_match_the_reference_vectors():
    """mmdeer/metrics.py (the numpy mirror used by the CPU-side evaluators) against vectors captured from the imported
    reference's DEERMetrics / uncertainty_calibration_error (tests/golden/make_golden.py metrics): CCC, MAE, RMSE per
    dimension and the quantile-binned ECE, incl. NaN / inf rows, tied uncertainties and a too-small set."""
    import os

    import numpy as np
    import pytest

    from mmdeer.metrics import DEERMetrics, uncertainty_calibration_error, validation_metrics

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_cases.npz"))
    m = DEERMetrics()
    for tag in ("main", "nan", "small", "ties"):
        p, t, u = g[f"{tag}.predictions"], g[f"{tag}.targets"], g[f"{tag}.uncertainties"]
        for i, d in enumerate(("valence", "arousal", "dominance")):
            assert m.concordance_correlation_coefficient(t[:, i], p[:, i]) == pytest.approx(float(g[f"{tag}.ccc_{d}"]), rel=1e-6, abs=1e-9), (tag, d)
            assert m.mean_absolute_error(t[:, i], p[:, i]) == pytest.approx(float(g[f"{tag}.mae_{d}"]), rel=1e-5), (tag, d)
            assert m.root_mean_squared_error(t[:, i], p[:, i]) == pytest.approx(float(g[f"{tag}.rmse_{d}"]), rel=1e-5), (tag, d)
        assert uncertainty_calibration_error(p, t, u) == pytest.approx(float(g[f"{tag}.ece"]), rel=1e-5, abs=1e-8), tag
        assert uncertainty_calibration_error(p, t, u, n_bins=15) == pytest.approx(float(g[f"{tag}.ece_15"]), rel=1e-5, abs=1e-8), tag
    p, t, u = g["main.predictions"], g["main.targets"], g["main.uncertainties"]
    vm = validation_metrics(p, t, u)
    assert vm["ece"] == pytest.approx(float(g["main.ece"]), rel=1e-5)
    assert vm["ccc_overall"] == pytest.approx(np.mean([float(g[f"main.ccc_{d}"]) for d in ("valence", "arousal", "dominance")]), rel=1e-6)


def test_create_fusion_module_factory():
    """fusion.create_fusion_module (fusion.py:557-592): the 'hierarchical' branch with the reference's defaults and key
    names; the 'attention' and concatenation branches build the mirrors of mmdeer.fusions (same state_dict keys), CPU tensors raise."""
    import pytest

    from mmdeer.model import HierarchicalMultimodalFusion, create_fusion_module

    gen = create_fusion_module("hierarchical", {})                     # the reference's defaults are 256/256/256 inputs: the operator path
    assert isinstance(gen, HierarchicalMultimodalFusion) and gen.audio_visual_fusion.audio_projection.weight.shape == (256, 256)
    assert gen.uncertainty_gate.modality_encoders[2][0].weight.shape == (128, 256)
    with pytest.raises(NotImplementedError, match="fusion_dim = 512"):
        create_fusion_module("hierarchical", {"fusion_dim": 256})      # the attention operators are built for 64-wide heads
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gen(torch.zeros(2, 256), torch.zeros(2, 256), torch.zeros(2, 256))
    m = create_fusion_module("Hierarchical", {"audio_dim": 84, "video_dim": 256, "text_dim": 768, "dropout": 0.1})
    assert isinstance(m, HierarchicalMultimodalFusion)
    assert "trimodal_fusion.modality_attention.in_proj_weight" in m.state_dict()
    from mmdeer import fusions
    att = create_fusion_module("attention", {})                        # fusion.py:579-583
    assert isinstance(att, fusions.AttentionFusion) and att.attention.weight.shape == (1, 512) and len(att.projections) == 3
    seq = create_fusion_module("concatenation", {"input_dims": [84, 256, 768], "dropout": 0.2})     # fusion.py:584-592
    assert isinstance(seq, torch.nn.Sequential) and seq[0].weight.shape == (512, 1108) and seq[2].p == 0.2
    assert sorted(seq.state_dict()) == ["0.bias", "0.weight", "3.bias", "3.weight"]
    ada = fusions.AdaptiveFusionGating([84, 256, 768], ["attention", "bilinear"], 256)
    assert ada.fusion_modules["bilinear"].bilinear.weight.shape == (256, 84, 256) and ada.strategy_selector[0].out_features == 2
    with pytest.raises(RuntimeError, match="no CPU fallback"):         # product path: HIP or nothing
        att([torch.zeros(2, 256)] * 3)


def test_compat_modules_resolve_by_bare_name():
    """SURVEY 8b row 1 / VERDICT r2 missing #5: with compat/ on sys.path the bare-name imports of the reference's script
    (run_multimodal_deer.py:70-82) resolve to the HIP-backed classes.  A fresh interpreter: names like `metrics`, `losses`,
    `training` must not leak into this process's sys.modules."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from multi_dataset_framework import MultiDatasetDEERFramework\n"
        "from complete_project import CompleteDEERModel, ModelConfig\n"
        "from training import DEERTrainer, TrainingConfig\n"
        "from preprocessing import create_enhanced_dataloaders\n"
        "from deer import test_deer_implementation\n"
        "from encoders import AudioEncoder, VideoEncoder, TextEncoder\n"
        "from fusion import HierarchicalMultimodalFusion\n"
        "from evaluation import evaluate_deer_model\n"
        "from metrics import DEERMetrics\n"
        "from losses import DEERLoss\n"
        "from visualization import create_comprehensive_report, test_visualization_components\n"
        "import mmdeer.model as M, mmdeer.trainer as T, mmdeer.losses as L, mmdeer.metrics as X\n"
        "assert CompleteDEERModel is M.MultimodalDEER and ModelConfig is M.ModelConfig\n"
        "assert DEERTrainer is T.DEERTrainer and TrainingConfig is T.TrainingConfig and evaluate_deer_model is T.evaluate_deer_model\n"
        "assert HierarchicalMultimodalFusion is M.HierarchicalMultimodalFusion and DEERLoss is L.DEERLoss and DEERMetrics is X.DEERMetrics\n"
        "cfg = ModelConfig(audio_dim=84, video_dim=256, text_dim=768, fusion_dim=512, emotion_dims=3, dropout=0.3, attention_heads=8)\n"
        "m = CompleteDEERModel(cfg)                      # run_multimodal_deer.py:237-247\n"
        "assert sum(p.numel() for p in m.parameters()) == 3105711\n"
        "for stub, args in ((MultiDatasetDEERFramework, ()), (AudioEncoder, ()), (create_enhanced_dataloaders, ())):\n"
        "    try:\n"
        "        stub(*args)\n"
        "    except NotImplementedError as e:\n"
        "        assert 'SURVEY' in str(e)\n"
        "    else:\n"
        "        raise SystemExit('out-of-scope stub did not raise')\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "compat"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    env = dict(os.environ, MMDEER_STACK="b")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                        "from complete_project import CompleteDEERModel\nimport mmdeer.stackb as S\nassert CompleteDEERModel is S.CompleteDEERModel"
                        % (ROOT, os.path.join(ROOT, "compat"))], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]


def test_option_table_matches_the_enum_order():
    """csrc/options.h (enum OptId) and the name table of csrc/api.hip are two lists that must stay in the same order: an option
    looked up by id would otherwise read its neighbour's value (it happened once: dw_tile read chain_max)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uncertainty-aware-multimodal-emotion-recognition_amd", "csrc")
    enum = [e.lower() for e in re.findall(r"^\s+OPT_([A-Z0-9_]+)\b", open(os.path.join(root, "options.h")).read(), re.M) if e != "COUNT"]
    table = re.findall(r'\{"([a-z0-9_]+)",\s*-?\d+,', open(os.path.join(root, "api.hip")).read())
    assert enum == table and len(enum) >= 15


def test_set_option_refuses_values_outside_the_range():
    """several options index tables (dw_tile -> tile sizes of the weight-gradient launch); a value outside an option's
    range must be refused with a message, and the option must keep its value."""
    from mmdeer import _lib
    lib = _lib.load()
    for name, bad in (("dw_tile", 7), ("dw_tile", -1), ("dw_tile", 1), ("splitk_max", 0), ("splitk_max", 99), ("chain", 2),
                      ("chain_min", 0), ("dw_kg", 3), ("tile", 5), ("chain_depth", 3 * 100)):
        before = _lib.get_option(name)
        assert lib.mmdeer_set_option(name.encode(), bad) == -1
        assert name in lib.mmdeer_last_error().decode() and "outside" in lib.mmdeer_last_error().decode()
        assert _lib.get_option(name) == before
    assert lib.mmdeer_set_option(b"no_such_option", 1) == -1 and "unknown" in lib.mmdeer_last_error().decode()
    with _lib.options(dw_tile=3, chain_depth=2):
        assert _lib.get_option("dw_tile") == 3 and _lib.get_option("chain_depth") == 2


def test_chain_weight_ring_registers_are_never_touched_by_the_compiler():
    """csrc/chain.hip keeps its weight ring in v[224:255] / v[240:255] behind amdgpu_num_vgpr; tools/check_chain_ring.py reads the
    device ISA and fails when the register allocator uses one of them (or an accumulation register) outside the kernel's asm."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_chain_ring", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   "tools", "check_chain_ring.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    problems, seen = mod.check(mod.device_asm())
    assert not problems, "\n".join(problems[:20])
    assert all(n > 0 for n in seen.values()) and len(seen) >= 2
