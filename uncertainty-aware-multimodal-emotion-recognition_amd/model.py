"""Host-side mirror of the reference's module protocol for the fusion + DEER path.

``MultimodalDEER`` is the composite SURVEY 8b defines (the reference has no class
of that name):

    HierarchicalMultimodalFusion(84, 256, 768, fusion 512, inter 256, 8 heads, p 0.3)   fusion.py:47-106
      -> MultiDimensionalDEER(512, 3 dims, hidden 256)                                  deer.py:201-231
      -> compute_loss == MultiTaskDEERLoss() defaults                                   losses.py:237-240

with the union of the calling conventions of the reference's script, trainer and
evaluator (``model(audio, video, text)``, ``model({'audio':..,'video':..,'text':..})``,
``compute_loss``, ``get_predictions_and_uncertainties``) and the reference's
``state_dict`` parameter names.  All arithmetic runs in libmmdeer_hip.so; this file
only owns parameters, buffers and the autograd glue.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, synth
from .spec import DEFAULT_DIMS, DIM_NAMES, Dims, gate_param_table, param_offsets, param_table

LOSS_KEYS = ("total_loss", "nll_loss", "reg_loss", "kl_loss", "ece_loss")


@dataclass
class ModelConfig:
    """Field names follow the reference's ModelConfig (complete_project.py:33-58)."""

    audio_dim: int = 84
    video_dim: int = 256
    text_dim: int = 768
    encoder_dim: int = 256
    fusion_dim: int = 512
    emotion_dims: int = 3
    attention_heads: int = 8
    encoder_layers: int = 3
    dropout: float = 0.3
    evidence_weight: float = 1.0
    kl_weight: float = 0.1
    learning_rate: float = 1e-4
    weight_decay: float = 1e-5
    gradient_clip: float = 1.0
    # build-specific knobs (not in the reference)
    compute_dtype: str = "fp32"   # "fp32": exact-fp32 MFMA (parity config); "bf16": bf16 MFMA, fp32 accumulate
    seed: int = 42                # parameter initialisation AND (unless dropout_seed is given) the dropout stream
    # Data parallel: every rank builds the model with the same `seed` (identical parameters) and indexes its rows from 0, so
    # with one dropout key all ranks would draw the SAME masks for their local rows -- correlated noise across the shards,
    # unlike DDP with per-rank RNG state.  Give each rank its own dropout_seed (bench.py: seed + 1000003 * rank).
    dropout_seed: Optional[int] = None


class _ParamTree(nn.Module):
    """Container whose parameters are registered under dotted (reference) names."""

    def add(self, dotted: str, tensor: torch.Tensor) -> nn.Parameter:
        parts = dotted.split(".")
        mod: nn.Module = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _ParamTree())
            mod = mod._modules[p]
        prm = nn.Parameter(tensor)
        mod.register_parameter(parts[-1], prm)
        return prm


def _as_cuda_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"mmdeer: {what} must be a CUDA (ROCm) tensor; the hot path has no CPU implementation")
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    return t.contiguous()


class _State:
    """Per-model launch state shared by the autograd functions."""

    def __init__(self):
        self.workspaces: Dict[Tuple[int, int, int], torch.Tensor] = {}     # activations / scratch, one per batch size
        self.weights: Dict[Tuple[int, int], torch.Tensor] = {}             # packed parameter copies: ONE per model (and device)
        self.generation = 0
        self.packed_key = None    # (weights buffer, parameter generation and versions) the packed copies were made from
        self.param_gen = 0        # bumped by in-place parameter updates torch cannot see (optim.FusedAdamW)


class _ForwardFn(torch.autograd.Function):
    """One call == mmdeer_forward; backward == mmdeer_backward in chain mode."""

    @staticmethod
    def forward(ctx, model: "MultimodalDEER", audio, video, text, targets, *params):
        out = model._launch_forward(audio, video, text, targets)
        ctx.model = model
        ctx.generation = model._st.generation
        ctx.inputs = (audio, video, text)
        ctx.meta = out["_meta"]
        nig = out["_nig"]
        res = tuple(nig[i] for i in range(7)) + (out["fused_features"], out["audiovisual_features"],
                                                 out["trimodal_features"], out["av_attention"],
                                                 out["trimodal_attention"])
        ctx.save_for_backward(nig)
        ctx.mark_non_differentiable(*res[8:])      # fused_features (res[7]) is differentiable: callers hang their own heads on it
        return res

    @staticmethod
    def backward(ctx, g_mu, g_nu, g_alpha, g_beta, g_alea, g_epis, g_unc, g_fused=None, *_unused):
        model = ctx.model
        (nig,) = ctx.saved_tensors
        # fold gradients that arrive through the derived uncertainties (deer.py:96-98) into nu / alpha / beta
        if g_alea is not None or g_epis is not None or g_unc is not None:
            nu, alpha, beta = nig[1], nig[2], nig[3]
            am1 = alpha - 1
            ga = (g_alea if g_alea is not None else 0) + (g_unc if g_unc is not None else 0)
            ge = (g_epis if g_epis is not None else 0) + (g_unc if g_unc is not None else 0)
            zero = torch.zeros_like(nu)
            ga = ga if torch.is_tensor(ga) else zero
            ge = ge if torch.is_tensor(ge) else zero
            d_beta = ga / am1 + ge / (nu * am1)
            d_alpha = -ga * beta / (am1 * am1) - ge * beta / (nu * am1 * am1)
            d_nu = -ge * beta / (nu * nu * am1)
            g_beta = d_beta if g_beta is None else g_beta + d_beta
            g_alpha = d_alpha if g_alpha is None else g_alpha + d_alpha
            g_nu = d_nu if g_nu is None else g_nu + d_nu
        grads = model._launch_backward(ctx, None, g_mu, g_nu, g_alpha, g_beta, g_fused=g_fused)
        return (None, None, None, None, None) + tuple(grads)


class _LossFn(torch.autograd.Function):
    """MultiTaskDEERLoss on (gamma, nu, alpha, beta): one pass computes the loss components and the
    gradients wrt the four inputs (mmdeer_nig_loss)."""

    @staticmethod
    def forward(ctx, gamma, nu, alpha, beta, targets, cfg: "_lib.LossCfg"):
        lib = _lib.load()
        B = gamma.shape[0]
        dev = gamma.device
        g, n, a, b = (t.contiguous().float() for t in (gamma, nu, alpha, beta))
        y = targets.contiguous().float()
        stats = torch.empty(lib.mmdeer_nig_stats_elems(B), dtype=torch.float32, device=dev)
        grads = torch.empty(4, B, 3, dtype=torch.float32, device=dev)
        loss_out = torch.empty(20, dtype=torch.float32, device=dev)
        bins = torch.empty(30, dtype=torch.int32, device=dev)
        _lib.check(lib.mmdeer_nig_loss(g.data_ptr(), n.data_ptr(), a.data_ptr(), b.data_ptr(), y.data_ptr(),
                                       stats.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                       grads[2].data_ptr(), grads[3].data_ptr(), loss_out.data_ptr(),
                                       bins.data_ptr(), B, C.byref(cfg), _lib.current_stream()))
        ctx.save_for_backward(grads)
        ctx.mark_non_differentiable(bins)
        return loss_out, bins

    @staticmethod
    def backward(ctx, g_loss_out, _g_bins):
        (grads,) = ctx.saved_tensors
        # only total_loss (element 16) is a training objective; the components are reported values
        s = g_loss_out[16]
        return grads[0] * s, grads[1] * s, grads[2] * s, grads[3] * s, None, None


def make_loss_cfg(reg_weight=0.1, kl_weight=0.01, ece_weight=0.05, cross_dim_weight=0.05,
                  task_weights=(1.0, 1.0, 1.0)) -> "_lib.LossCfg":
    cfg = _lib.LossCfg()
    cfg.reg_weight, cfg.kl_weight, cfg.ece_weight, cfg.cross_weight = reg_weight, kl_weight, ece_weight, cross_dim_weight
    for i in range(3):
        cfg.task_weight[i] = float(task_weights[i])
    return cfg


def loss_dict_from(loss_out: torch.Tensor, batch_size: int) -> Dict[str, torch.Tensor]:
    """Reference key layout of MultiTaskDEERLoss.forward (losses.py:303-318)."""
    d: Dict[str, torch.Tensor] = {}
    for i, dim in enumerate(DIM_NAMES):
        for j, k in enumerate(LOSS_KEYS):
            d[f"{dim}_{k}"] = loss_out[i * 5 + j]
        d[f"{dim}_batch_size"] = batch_size
    d["cross_dim_loss"] = loss_out[15]
    d["total_loss"] = loss_out[16]
    # keys the reference trainer accumulates when present (training.py:187-190)
    d["deer_loss"] = loss_out[16]
    d["nll_loss"], d["evidence_reg"], d["kl_reg"] = loss_out[17], loss_out[18], loss_out[19]   # computed in-kernel
    return d


class MultimodalDEER(nn.Module):
    def __init__(self, config: Optional[ModelConfig] = None, init: str = "reference"):
        super().__init__()
        self.config = config or ModelConfig()
        c = self.config
        d = DEFAULT_DIMS
        if (c.audio_dim, c.video_dim, c.text_dim, c.fusion_dim, c.emotion_dims, c.attention_heads) != (
                d.audio, d.video, d.text, d.fusion, d.ndim, d.heads):
            raise NotImplementedError(
                "libmmdeer_hip.so is specialised for the reference geometry 84/256/768 -> 512, 8 heads, 3 dims "
                f"(got {c})")
        if c.compute_dtype not in ("fp32", "bf16"):
            raise ValueError("compute_dtype must be 'fp32' or 'bf16'")
        self.dims = Dims(dropout=c.dropout)
        self.fusion = _ParamTree()
        self.head = _ParamTree()
        state = (synth.reference_init_state(self.dims, seed=c.seed) if init == "reference"
                 else synth.closed_form_state(self.dims, include_gate=True))
        self._live = []
        for name, _shape, _ in param_table(self.dims):
            tree, sub = (self.fusion, name[len("fusion."):]) if name.startswith("fusion.") else (self.head, name[len("head."):])
            self._live.append(tree.add(sub, torch.from_numpy(state[name].copy())))
        # kept for state_dict compatibility only: unreachable in the reference, never gets a gradient
        for name, _shape, _ in gate_param_table(self.dims):
            self.fusion.add(name[len("fusion."):], torch.from_numpy(state[name].copy()))
        self._offsets, self._flat_elems = param_offsets(self.dims)
        self.loss_cfg = make_loss_cfg()
        self._st = _State()
        self._step = 0          # dropout counter offset: advanced once per training forward
        self._flat_grad: Optional[torch.Tensor] = None
        self._step_flat: Optional[torch.Tensor] = None
        self._step_views = None
        self._grads_bound = False
        self._ptr_cache = None     # (tuple of data_ptrs, ctypes array) of the live parameters

    # ------------------------------------------------------------------ plumbing
    @property
    def compute_f32(self) -> int:
        return 1 if self.config.compute_dtype == "fp32" else 0

    @property
    def dropout_seed(self) -> int:
        ds = getattr(self.config, "dropout_seed", None)
        return int(self.config.seed if ds is None else ds)

    def live_parameters(self):
        return list(self._live)

    def mark_parameters_changed(self) -> None:
        """Force the next call to re-derive the packed weight copies.  Edits through ``load_state_dict``, optimisers and
        ordinary in-place ops are seen on their own (tensor version counters); writes through ``param.data`` are not."""
        self._st.param_gen += 1
        self._st.packed_key = None

    def load_reference_state_dicts(self, fusion_sd: Dict[str, torch.Tensor], head_sd: Dict[str, torch.Tensor]):
        """Load state_dicts saved from the reference's HierarchicalMultimodalFusion / MultiDimensionalDEER."""
        self.fusion.load_state_dict(fusion_sd)
        self.head.load_state_dict(head_sd)

    def load_numpy_state(self, state) -> None:
        sd = {k: torch.as_tensor(v) for k, v in state.items()}
        own = self.state_dict()
        for k in own:
            if k in sd:
                own[k].copy_(sd[k])

    def _workspace(self, B: int, dev: torch.device) -> torch.Tensor:
        key = (B, self.compute_f32, dev.index if dev.index is not None else 0)
        ws = self._st.workspaces.get(key)
        if ws is None:
            nbytes = _lib.load().mmdeer_workspace_bytes(B, self.compute_f32)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._st.workspaces[key] = ws
        return ws

    def _weights(self, dev: torch.device) -> torch.Tensor:
        """The packed parameter copies the kernels read: one buffer per model, shared by every batch size's workspace
        (a graph captured for one batch size and an eager step on another read the same, always current, copies)."""
        key = (self.compute_f32, dev.index if dev.index is not None else 0)
        wb = self._st.weights.get(key)
        if wb is None:
            wb = torch.empty(_lib.load().mmdeer_weights_bytes(self.compute_f32), dtype=torch.uint8, device=dev)
            self._st.weights[key] = wb
            self._st.packed_key = None
        return wb

    def _param_key(self, wb: torch.Tensor):
        return (wb.data_ptr(), self._st.param_gen) + tuple((p.data_ptr(), p._version) for p in self._live)

    def _launch_forward(self, audio, video, text, targets, prof_events=None, offset_dev=None, want_features=True, bump=False):
        lib = _lib.load()
        audio, video, text = (_as_cuda_f32(t, n) for t, n in ((audio, "audio"), (video, "video"), (text, "text")))
        B = audio.shape[0]
        if audio.shape != (B, self.dims.audio) or video.shape != (B, self.dims.video) or text.shape != (B, self.dims.text):
            raise ValueError(f"expected (B,84) (B,256) (B,768), got {tuple(audio.shape)} {tuple(video.shape)} {tuple(text.shape)}")
        in_bf16 = audio.dtype == torch.bfloat16
        if not (audio.dtype == video.dtype == text.dtype):
            raise ValueError("audio / video / text must share one dtype")
        if in_bf16 and self.compute_f32:
            audio, video, text = audio.float(), video.float(), text.float()
            in_bf16 = False
        dev = audio.device
        ptrs = tuple(p.data_ptr() for p in self._live)
        if self._ptr_cache is None or self._ptr_cache[0] != ptrs or self._ptr_cache[2] != dev:
            for p in self._live:
                if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("mmdeer: parameters must be contiguous fp32 tensors on the inputs' device")
            self._ptr_cache = (ptrs, (C.c_void_p * len(ptrs))(*ptrs), dev)
        ws = self._workspace(B, dev)
        wb = self._weights(dev)
        key = self._param_key(wb)
        repack = key != self._st.packed_key
        training = self.training
        if training and offset_dev is None:
            self._step += 1
        a = _lib.ForwardArgs()
        a.batch, a.compute_f32, a.training, a.inputs_bf16, a.repack = B, self.compute_f32, int(training), int(in_bf16), int(repack)
        # graph mode: the step counter lives on the device (offset_dev) and the host-side offset is 0
        step = 0 if offset_dev is not None else int(self._step)
        a.dropout_p, a.seed, a.offset = float(self.dims.dropout), self.dropout_seed, step
        a.offset_dev = _lib.ptr(offset_dev)
        a.bump_offset_dev = int(bump)
        a.audio, a.video, a.text = audio.data_ptr(), video.data_ptr(), text.data_ptr()
        a.params = self._ptr_cache[1]
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.weights, a.weights_bytes = wb.data_ptr(), wb.numel()
        f32 = dict(dtype=torch.float32, device=dev)
        nig = torch.empty(7, B, 3, **f32)
        a.nig_out = nig.data_ptr()
        fused = avf = trif = avw = triw = None
        if want_features:   # fp32 copies of the fusion features / attention weights of the reference's output dict
            fused = torch.empty(B, self.dims.fusion, **f32)
            avf = torch.empty(B, self.dims.inter, **f32)
            trif = torch.empty(B, self.dims.fusion, **f32)
            avw = torch.empty(B, 2, **f32)
            triw = torch.empty(B, 2, 2, **f32)
            a.fused_features, a.audiovisual_features = fused.data_ptr(), avf.data_ptr()
            a.trimodal_features, a.av_attention, a.trimodal_attention = trif.data_ptr(), avw.data_ptr(), triw.data_ptr()
        if targets is not None:
            targets = targets.contiguous().float()
            a.targets = targets.data_ptr()
        if prof_events is not None:
            a.prof_events[0], a.prof_events[1] = prof_events[0].cuda_event, prof_events[1].cuda_event
        a.stream = _lib.current_stream()
        _lib.check(lib.mmdeer_forward(C.byref(a)))
        self._st.packed_key = key
        self._st.generation += 1
        meta = dict(B=B, training=int(training), in_bf16=int(in_bf16), offset=step, offset_dev=offset_dev, bump=int(bump), ws=ws, wb=wb,
                    inputs=(audio, video, text), targets=targets)
        return {"_nig": nig, "_meta": meta, "fused_features": fused, "audiovisual_features": avf,
                "trimodal_features": trif, "av_attention": avw, "trimodal_attention": triw}

    def _launch_backward(self, ctx_or_meta, targets, g_mu=None, g_nu=None, g_alpha=None, g_beta=None,
                         loss_out=None, bin_counts=None, events=None, flat=None, want_views=True, phase=0,
                         global_stats=None, g_fused=None):
        lib = _lib.load()
        if isinstance(ctx_or_meta, dict):
            meta = ctx_or_meta
        else:
            meta = ctx_or_meta.meta
            if ctx_or_meta.generation != self._st.generation:
                raise RuntimeError("mmdeer: the activations of this forward were overwritten by a later forward of the "
                                   "same model; call backward() before the next forward (one workspace per batch size)")
        B = meta["B"]
        dev = meta["ws"].device
        if flat is None:
            # autograd path: a fresh buffer per backward so gradients handed to autograd never alias a later
            # backward; zero-filled because the 64-element alignment gaps between tensors are never written
            flat = torch.zeros(self._flat_elems, dtype=torch.float32, device=dev)
        a = _lib.BackwardArgs()
        a.batch, a.compute_f32, a.training, a.inputs_bf16 = B, self.compute_f32, meta["training"], meta["in_bf16"]
        a.dropout_p, a.seed, a.offset = float(self.dims.dropout), self.dropout_seed, meta["offset"]
        a.offset_dev = _lib.ptr(meta.get("offset_dev"))
        a.bump_offset_dev = int(meta.get("bump", 0))      # the forward ran with the pending increment: the backward's last launch stores it
        audio, video, text = meta["inputs"]
        a.audio, a.video, a.text = audio.data_ptr(), video.data_ptr(), text.data_ptr()
        a.workspace, a.workspace_bytes = meta["ws"].data_ptr(), meta["ws"].numel()
        a.weights, a.weights_bytes = meta["wb"].data_ptr(), meta["wb"].numel()
        keep = []
        if targets is not None:
            a.targets = targets.data_ptr()
        else:
            for name, g in (("g_mu", g_mu), ("g_nu", g_nu), ("g_alpha", g_alpha), ("g_beta", g_beta)):
                if g is not None:
                    g = g.contiguous().float()
                    keep.append(g)
                    setattr(a, name, g.data_ptr())
        a.loss = self.loss_cfg
        a.grads = flat.data_ptr()
        a.loss_out = _lib.ptr(loss_out)
        a.bin_counts = _lib.ptr(bin_counts)
        if events is not None:
            for i, ev in enumerate(events):
                a.bucket_events[i] = ev.cuda_event
        a.phase = int(phase)
        a.global_stats = _lib.ptr(global_stats)
        if g_fused is not None:
            g_fused = g_fused.contiguous().float()
            keep.append(g_fused)
            a.g_fused = g_fused.data_ptr()
        a.stream = _lib.current_stream()
        _lib.check(lib.mmdeer_backward(C.byref(a)))
        self._flat_grad = flat
        if not want_views:
            return None
        return [flat[off:off + p.numel()].view(p.shape) for p, off in zip(self._live, self._offsets)]

    # ------------------------------------------------------------------ reference-facing API
    def forward(self, audio_features, video_features=None, text_features=None, targets=None) -> Dict[str, torch.Tensor]:
        """``model(audio[B,84], video[B,256], text[B,768])`` (training.py:207, evaluation.py:171) or
        ``model({'audio':..., 'video':..., 'text':...})`` (run_multimodal_deer.py:414-427)."""
        if isinstance(audio_features, dict):
            d = audio_features
            audio_features = d.get("audio", d.get("audio_features"))
            video_features = d.get("video", d.get("video_features"))
            text_features = d.get("text", d.get("text_features"))
        if audio_features is None or video_features is None or text_features is None:
            raise ValueError("audio, video and text features are all required")
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._live)
        if need_grad:
            res = _ForwardFn.apply(self, audio_features, video_features, text_features, targets, *self._live)
            mu, nu, alpha, beta, alea, epis, unc, fused, avf, trif, avw, triw = res
        else:
            o = self._launch_forward(audio_features, video_features, text_features, targets)
            mu, nu, alpha, beta, alea, epis, unc = (o["_nig"][i] for i in range(7))
            fused, avf, trif, avw, triw = (o["fused_features"], o["audiovisual_features"], o["trimodal_features"],
                                           o["av_attention"], o["trimodal_attention"])
        out: Dict[str, torch.Tensor] = {}
        per_dim = (("mu", mu), ("nu", nu), ("alpha", alpha), ("beta", beta), ("aleatoric_uncertainty", alea),
                   ("epistemic_uncertainty", epis), ("uncertainty", unc))
        for i, dim in enumerate(DIM_NAMES):           # deer.py:254-255 -- (B,1) tensors
            for key, val in per_dim:
                out[f"{dim}_{key}"] = val[:, i:i + 1]
        out["mu_all"] = mu                            # deer.py:258-264
        out["uncertainty_all"] = unc
        # script / evaluator conventions (run_multimodal_deer.py:430,721; evaluation.py:175-176)
        out["gamma"], out["nu"], out["alpha"], out["beta"] = mu, nu, alpha, beta
        out["mu"] = out["predictions"] = mu
        out["uncertainties"] = out["total_uncertainty"] = unc
        out["aleatoric_uncertainty"], out["epistemic_uncertainty"] = alea, epis
        # fusion extras (fusion.py:164-171)
        out["fused_features"] = fused
        out["audiovisual_features"] = avf
        out["trimodal_features"] = trif
        out["av_attention_weights"] = {"audio_to_video": avw[:, 0:1], "video_to_audio": avw[:, 1:2]}
        out["trimodal_attention_weights"] = triw
        out["uncertainty_weights"] = None
        return out

    def get_predictions_and_uncertainties(self, outputs: Dict[str, torch.Tensor]):
        """complete_project.py:590-602."""
        return outputs["mu_all"], outputs.get("calibrated_uncertainty", outputs["uncertainty_all"])

    def compute_loss(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        """The trainer hook (training.py:210): MultiTaskDEERLoss defaults, dict with a backprop-able 'total_loss'."""
        return multitask_deer_loss(predictions, targets, self.loss_cfg)

    def train_step(self, audio, video, text, targets, events=None, prof_events=None, _offset_dev=None,
                   return_features: bool = False, _bump: bool = False, comm=None, stats_comm=None) -> Dict[str, torch.Tensor]:
        """Fused forward + MultiTaskDEERLoss + backward: two library calls, gradients land in one flat buffer
        (``.grad`` of every live parameter is a view of it).  Equivalent to
        ``compute_loss(model(a, v, t), y)['total_loss'].backward()``.

        ``stats_comm`` (data parallel, optional): a communicator with ``sum_small(tensor)`` (``parallel.BucketedAllReduce``).
        The 106 loss statistics of this rank's batch are summed across ranks between forward and backward, so the loss
        is the global batch's (its ECE and cross-dimension terms are non-linear in batch statistics) and the
        gradients are this rank's share of its gradient: exchange them with a SUM (``comm.exact_global = True``)."""
        # the fused step returns losses and gradients; the fp32 feature copies of forward()'s output dict are written
        # only on request (24 MB of stores per step at B = 4096 that nothing in training reads)
        o = self._launch_forward(audio, video, text, targets, prof_events, offset_dev=_offset_dev,
                                 want_features=return_features, bump=_bump)
        meta = o["_meta"]
        dev = meta["ws"].device
        loss_out = torch.empty(20, dtype=torch.float32, device=dev)
        bins = torch.empty(30, dtype=torch.int32, device=dev)
        # the fused path owns ONE persistent flat gradient buffer per device (zeroed once: the alignment gaps stay 0);
        # every step overwrites all live slices, and .grad of each live parameter is a view of it
        if self._step_flat is None or self._step_flat.device != dev:
            self._step_flat = torch.zeros(self._flat_elems, dtype=torch.float32, device=dev)
            self._step_views = None
        gstats = None
        if stats_comm is not None and getattr(stats_comm, "active", False):
            gstats = torch.empty(106, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().mmdeer_loss_stats(meta["ws"].data_ptr(), meta["ws"].numel(), meta["B"], self.compute_f32,
                                                     gstats.data_ptr(), _lib.current_stream()))
            stats_comm.sum_small(gstats)
        if comm is not None and getattr(comm, "active", False):
            # data parallel, overlapped: the backward pass in two calls; the all-reduce of buckets 0-1 (head, output
            # projection, trimodal fusion: 89 % of the gradient) runs on the communicator's side stream while the
            # audio-visual part of the pass, its weight gradients and their reduction are computed
            lo = int(_lib.load().mmdeer_bucket_end(2))
            self._launch_backward(meta, meta["targets"], loss_out=loss_out, bin_counts=bins, flat=self._step_flat,
                                  want_views=False, phase=1, global_stats=gstats)
            comm.launch_range(self._step_flat, lo, self._flat_elems)
            views = self._launch_backward(meta, meta["targets"], loss_out=loss_out, bin_counts=bins, flat=self._step_flat,
                                          want_views=self._step_views is None, phase=2, global_stats=gstats)
            comm.launch_range(self._step_flat, 0, lo)
            comm.join()
        else:
            views = self._launch_backward(meta, meta["targets"], loss_out=loss_out, bin_counts=bins, events=events,
                                          flat=self._step_flat, want_views=self._step_views is None, global_stats=gstats)
        if self._step_views is None:
            self._step_views = views
            self._grads_bound = False
        if (not self._grads_bound or self._live[0].grad is not self._step_views[0]
                or self._live[-1].grad is not self._step_views[-1]):
            for p, g in zip(self._live, self._step_views):
                if p.grad is not g:
                    p.grad = g
            self._grads_bound = True
        d = loss_dict_from(loss_out, meta["B"])
        d["ece_bin_counts"] = bins.view(3, 10)
        d["_outputs"] = o
        return d

    def flat_grad(self) -> Optional[torch.Tensor]:
        return self._flat_grad

    # launch plans autotune_launch_plan() chooses between (library options, include/mmdeer.h): identical results up to rounding
    LAUNCH_PLANS = (("layer chains", dict(chain=1, chain_bwd=1)),
                    ("forward chains only", dict(chain=1, chain_bwd=0)),
                    ("separate launches", dict(chain=0)))

    def autotune_launch_plan(self, audio, video, text, targets, replays: int = 20, reduce_max=None, stats_comm=None) -> Dict[str, object]:
        """Time the captured training step under each launch plan on THIS GPU and make the fastest the library's plan.

        The layer chains (csrc/chain.hip) are bound by the L2 -> CU path, whose rate differs from box to box far more than the
        MFMA or HBM rates do: on the boxes of one pool the same build gained 15 us per step from them on one and lost 12 us on
        another.  The plans compute the same step (forward bit for bit, gradients up to the summation
        order of a few partials), so the choice is a pure timing decision, made once before a run like the data-parallel
        exchange plan.  Every candidate is captured into a HIP graph and replayed ``replays`` times between two events;
        ``reduce_max`` (optional callable float -> float, e.g. an all-reduce MAX) makes every rank see the same timings.
        Returns {"plan": name, "options": {...}, "ms": {name: ms per step}}; plans that do not apply to this batch size /
        dtype collapse into one and nothing is timed."""
        B = int(audio.shape[0])
        lo, hi = _lib.get_option("chain_min"), _lib.get_option("chain_max")
        if self.compute_f32 or not (lo <= B <= hi) or not self.training:
            return {"plan": "separate launches", "options": {}, "ms": {}, "why": "the layer chains do not apply to this dtype / batch size"}
        graphs = []
        for name, opts in self.LAUNCH_PLANS:
            with _lib.options(**opts):
                graphs.append((name, self.capture_train_step(audio, video, text, targets, stats_comm=stats_comm)))
        # two interleaved rounds, the better one counts: the first plan timed must not pay for clocks that are still ramping up
        ms: Dict[str, float] = {}
        for _round in range(2):
            for name, r in graphs:
                for _ in range(5):
                    r()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(replays):
                    r()
                e1.record()
                torch.cuda.synchronize(audio.device)
                t = e0.elapsed_time(e1) / replays
                t = float(reduce_max(t)) if reduce_max is not None else t
                ms[name] = round(min(t, ms.get(name, t)), 4)
        del graphs
        self._graph = None
        best = min(ms, key=ms.get)
        opts = dict(self.LAUNCH_PLANS)[best]
        for k, v in opts.items():
            _lib.set_option(k, v)
        return {"plan": best, "options": dict(opts), "ms": ms,
                "why": "fastest of the launch plans on this GPU (best of two interleaved rounds of %d graph replays each)" % replays}

    def capture_train_step(self, audio, video, text, targets, events=None, after=None, comm=None, stats_comm=None):
        """Capture ``train_step`` on these (static) input tensors into a HIP graph and return ``replay()``.

        One step is ~45 kernel launches of 4-40 us each; enqueueing them from the host costs about as much as the GPU
        needs to run them, a graph replay costs ~15 us.  ``replay()`` returns the same loss dict every time (its
        tensors are overwritten in place); new data is fed by copying into ``audio/video/text/targets``.  Dropout
        masks advance through a device-side counter (``offset_dev``), so replays draw fresh masks.  The packed
        weight copies must be current at capture time and be kept current by ``optim.FusedAdamW`` (an eager call
        between replays); shapes, dtype and train mode are frozen into the graph.  ``after`` (optional callable) runs
        inside the capture right after the step -- e.g. the data-parallel gradient all-reduce, so that it is replayed
        with the step instead of being enqueued from the host every time."""
        if not self.training:
            raise RuntimeError("capture_train_step: call .train() first")
        for t in (audio, video, text, targets):
            if not t.is_cuda:
                raise RuntimeError("capture_train_step needs GPU tensors")
        dev = audio.device
        # eager warm-up: allocates the workspace and the persistent gradient buffer, packs the weights.  It is a real
        # step on the given batch: its loss dict is returned as ``replay.first`` and its gradients are in place
        first = self.train_step(audio, video, text, targets, comm=comm, stats_comm=stats_comm)
        self._graph_counter = torch.full((), int(self._step), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        in_kernel = not self.compute_f32     # bf16: the step's kernels add the pending 1 themselves and its last launch stores the counter
        # With a collective in the capture (or a process group alive at all) the process has other threads that talk to the runtime (the communicator's watchdog
        # polls the events of earlier eager collectives): in the default "global" mode such a call from another thread, landing
        # while this one captures, invalidates the capture.  Only this thread's calls matter here.
        import torch.distributed as _dist
        with_comm = after is not None or comm is not None or stats_comm is not None or (_dist.is_available() and _dist.is_initialized())
        mode = "thread_local" if with_comm else "global"
        with torch.cuda.graph(graph, capture_error_mode=mode):
            if not in_kernel:
                self._graph_counter.add_(1)
            out = self.train_step(audio, video, text, targets, events=events, _offset_dev=self._graph_counter, _bump=in_kernel,
                                  comm=comm, stats_comm=stats_comm)
            if after is not None:
                after()
        self._graph = graph

        counter = self._graph_counter
        shadow = [int(self._step)]                    # the value the device-side dropout counter holds

        def replay():
            # eager steps in between (a ragged tail batch) advanced the host-side dropout counter only: bring the device
            # counter in line, so a replay never reuses the offset of the step before it
            if self._step != shadow[0]:
                counter.fill_(int(self._step))
            # repack = 0 is frozen into the graph.  The packed copies are per model, so FusedAdamW steps taken after eager
            # steps on other batch sizes keep them current; only parameters changed behind the library's back
            # (load_state_dict, a torch optimiser, mark_parameters_changed) need a refresh before the replay.
            wb = self._weights(dev)
            if self._st.packed_key != self._param_key(wb):
                _lib.check(_lib.load().mmdeer_pack_weights(self._ptr_cache[1], wb.data_ptr(), wb.numel(), self.compute_f32,
                                                           _lib.current_stream()))
                self._st.packed_key = self._param_key(wb)
            graph.replay()
            self._step += 1          # keep the host-side step counter in line with the device-side one
            shadow[0] = int(self._step)
            return out
        replay.first = first
        return replay


def _stack_pred(predictions: Dict[str, torch.Tensor], names) -> torch.Tensor:
    for n in names:
        if n in predictions and torch.is_tensor(predictions[n]) and predictions[n].dim() == 2 and predictions[n].shape[1] == 3:
            return predictions[n]
    cols = []
    for dim in DIM_NAMES:
        for n in names:
            k = f"{dim}_{n}"
            if k in predictions:
                cols.append(predictions[k].reshape(-1, 1))
                break
        else:
            raise ValueError(f"Missing required NIG parameters in predictions ({dim}: {names})")
    return torch.cat(cols, dim=1)


def multitask_deer_loss(predictions: Dict[str, torch.Tensor], targets: torch.Tensor,
                        cfg: Optional["_lib.LossCfg"] = None) -> Dict[str, torch.Tensor]:
    """MultiTaskDEERLoss.forward (losses.py:268-318) on a prediction dict, key aliases of losses.py:286-291."""
    cfg = cfg or make_loss_cfg()
    gamma = _stack_pred(predictions, ("gamma", "mu"))
    nu = _stack_pred(predictions, ("nu", "lambda"))
    alpha = _stack_pred(predictions, ("alpha",))
    beta = _stack_pred(predictions, ("beta",))
    if targets.dim() != 2 or targets.shape[1] != 3:
        raise ValueError("targets must be (B, 3)")
    loss_out, bins = _LossFn.apply(gamma, nu, alpha, beta, targets, cfg)
    d = loss_dict_from(loss_out, gamma.shape[0])
    d["ece_bin_counts"] = bins.view(3, 10)
    return d


# names the reference's script imports (run_multimodal_deer.py:72-82)
CompleteDEERModel = MultimodalDEER


class HierarchicalMultimodalFusion(nn.Module):
    """``fusion.HierarchicalMultimodalFusion`` (src/models/fusion.py:35-185) for callers that use the fusion on its own:
    same constructor arguments, ``forward(audio, video, text, uncertainties=None)`` and output dictionary
    (fusion.py:164-171), and ``state_dict()`` has exactly the reference class's keys, so its checkpoints load.

    Default geometry (84 / 256 / 768 -> 256 -> 512, 8 heads): the arithmetic is the fused fusion + head pass of ``MultimodalDEER``
    (the head is a few launches and its outputs are dropped); ``fused_features`` is differentiable -- a caller's own head trains the
    fusion through it (``mmdeer_backward``'s ``g_fused``); ``audiovisual_features`` / ``trimodal_features`` are returned as values
    only.  Other ``audio_dim`` / ``video_dim`` / ``text_dim``: the operator path of ``generic_fusion.GenericHierarchicalFusion``
    (every output differentiable; pinned by tests/golden/fusion_geom.npz, captured from the reference at 40 / 128 / 300)."""

    KEYS = ("fused_features", "audiovisual_features", "trimodal_features", "av_attention_weights",
            "trimodal_attention_weights", "uncertainty_weights")

    def __init__(self, audio_dim: int, video_dim: int, text_dim: int, fusion_dim: int = 512, intermediate_dim: int = 256,
                 num_attention_heads: int = 8, dropout: float = 0.3, use_uncertainty_weighting: bool = True,
                 compute_dtype: str = "fp32", seed: int = 0):
        super().__init__()
        self.use_uncertainty_weighting = use_uncertainty_weighting
        default_geometry = (audio_dim, video_dim, text_dim, fusion_dim, intermediate_dim, num_attention_heads) == (
            DEFAULT_DIMS.audio, DEFAULT_DIMS.video, DEFAULT_DIMS.text, DEFAULT_DIMS.fusion, DEFAULT_DIMS.inter, 8)
        if not default_geometry:
            # other input widths (fusion.py:47-50 takes any): the operator path -- one C-ABI call per layer, forward and backward
            # (generic_fusion.py).  Same output dictionary, same state_dict keys for these dimensions.
            from .generic_fusion import GenericHierarchicalFusion
            gen = GenericHierarchicalFusion(audio_dim, video_dim, text_dim, fusion_dim, intermediate_dim, num_attention_heads, dropout,
                                            use_uncertainty_weighting, compute_dtype, seed)
            self.__dict__["_core"] = None
            self.__dict__["_generic"] = gen
            for name, child in gen.named_children():
                self.add_module(name, child)
            return
        self.__dict__["_generic"] = None
        core = MultimodalDEER(ModelConfig(audio_dim=audio_dim, video_dim=video_dim, text_dim=text_dim, fusion_dim=fusion_dim,
                                          attention_heads=num_attention_heads, dropout=dropout, compute_dtype=compute_dtype,
                                          seed=seed))
        self.__dict__["_core"] = core             # not a registered child: the head's parameters stay out of state_dict()
        for name, child in core.fusion.named_children():
            self.add_module(name, child)          # the fusion's own parameter tree, shared with the core

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if self._core is not None:
            self._core.head._apply(fn, *args, **kwargs)   # .to(device) / .float() reach the hidden head as well
        return self

    def train(self, mode: bool = True):
        super().train(mode)
        if self._core is not None:
            self._core.train(mode)
        else:
            self._generic.train(mode)
        return self

    def forward(self, audio_features, video_features, text_features, uncertainties=None) -> Dict[str, torch.Tensor]:
        if self.use_uncertainty_weighting and uncertainties is not None:
            # Same failure as the reference: with `uncertainties` given and use_uncertainty_weighting=True its forward calls
            # self.uncertainty_gate(audio, video, text, uncertainties) positionally (fusion.py:148-150) against
            # forward(self, *modality_features, uncertainties) (:384) and dies with exactly this TypeError.  Algebraically the
            # branch would scale the features by (1 + 0.1 / 3) (the mean of a softmax over three is 1/3, :173-185); it has no
            # working caller anywhere in the reference, so it is not built and parity stays unpinned.  With
            # use_uncertainty_weighting=False the reference never enters the branch (:147) and `uncertainties` is ignored.
            raise TypeError("forward() missing 1 required keyword-only argument: 'uncertainties' "
                            "[UncertaintyAwareGating is unreachable in the reference (fusion.py:148-150 vs :384); "
                            "call with uncertainties=None]")
        if self._generic is not None:
            return self._generic(audio_features, video_features, text_features)
        out = self._core(audio_features, video_features, text_features)
        return {k: out[k] for k in self.KEYS}


def create_fusion_module(fusion_type: str, config: Dict) -> nn.Module:
    """``fusion.create_fusion_module`` (src/models/fusion.py:557-592) with the reference's key lookups and defaults:
    ``'hierarchical'`` (:568-578) -> ``HierarchicalMultimodalFusion``, ``'attention'`` (:579-583) -> ``fusions.AttentionFusion``,
    anything else (:584-592) -> the Linear-ReLU-Dropout-LayerNorm block (``fusions.ConcatFusion``, an ``nn.Sequential`` with the
    same child indices).  Optional extra keys: ``compute_dtype`` ('fp32' | 'bf16'), ``seed``."""
    from . import fusions
    kind = str(fusion_type).lower()
    extra = {k: config[k] for k in ("compute_dtype",) if k in config}
    if kind == "hierarchical":
        return HierarchicalMultimodalFusion(
            audio_dim=config.get("audio_dim", 256), video_dim=config.get("video_dim", 256), text_dim=config.get("text_dim", 256),
            fusion_dim=config.get("fusion_dim", 512), intermediate_dim=config.get("intermediate_dim", 256),
            num_attention_heads=config.get("num_attention_heads", 8), dropout=config.get("dropout", 0.3),
            use_uncertainty_weighting=config.get("use_uncertainty_weighting", True),
            **{k: config[k] for k in ("compute_dtype", "seed") if k in config})
    if kind == "attention":
        return fusions.AttentionFusion(input_dims=config.get("input_dims", [256, 256, 256]), output_dim=config.get("fusion_dim", 512), **extra)
    return fusions.ConcatFusion(sum(config.get("input_dims", [256, 256, 256])), config.get("fusion_dim", 512), config.get("dropout", 0.3),
                                dropout_seed=config.get("seed", 0), **extra)


def create_model(config: Optional[ModelConfig] = None, device: Optional[str] = None) -> MultimodalDEER:
    m = MultimodalDEER(config)
    return m.to(device) if device else m
