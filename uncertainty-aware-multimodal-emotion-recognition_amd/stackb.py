"""Stack B: ``CompleteDEERModel`` of the reference's src/models/complete_project.py -- inference forward as
one library call, training forward (dropout) + backward as operator sequences (``stackb_train.py``).

Same constructor protocol (``ModelConfig``), ``state_dict()`` keys and shapes, output dictionary and
``get_predictions_and_uncertainties`` as the reference class (complete_project.py:33-56, 462-602), so a reference
checkpoint loads with ``load_state_dict`` and callers of ``model(audio, video, text)`` keep working.  The arithmetic
runs on the HIP library only -- one ``mmdeer_stackb_forward`` call per batch (25 kernel launches: grouped GEMMs plus
the four ``mmdeer_stackb_*`` row kernels); there is no CPU path.

How the batch is laid out (csrc/stackb.hip carves the intermediates out of one workspace, all row-major in HBM):
  * the three encoders write their (B, 256) outputs into column blocks of one (B, 768) matrix, which read as
    (3B, 256) is the (sample, modality)-interleaved row set that the shared-weight layers -- both attention blocks and
    the uncertainty estimator -- consume in a single GEMM each;
  * with one key per query the reference's MultiHeadAttention softmax is identically 1 (complete_project.py:160-171), so
    self / cross attention are ``output_proj(value_proj(.))``; the two value projections share one N = 512 GEMM, and the
    (3B, 256) self-attention output read as (B, 768) is already ``cat([audio_self, video_self, text_self])``;
  * ``weight_network.0`` has K = 771: its 768 feature columns run as a GEMM, the three uncertainty columns are added
    in ``mmdeer_stackb_attn_mix`` together with the ReLU, the 256 -> 3 layer, the softmax and the final mix;
  * the three encoders advance side by side: every encoder layer is one grouped / batched GEMM launch over the
    modality-major (3, B, 256) hidden state plus one residual-LayerNorm launch;
  * the first layer of the three prediction heads is one N = 768 GEMM, the other two are batched over the heads.

Training: in ``.train()`` mode ``forward`` runs ``stackb_train.forward_train`` (every ``nn.Dropout`` site of the reference
as counter-hash dropout in the GEMM epilogues) and returns (mu, nu, alpha, beta) attached to an autograd node whose
backward is ``stackb_train.backward`` -- so ``model.compute_loss(model(a, v, t), y)['total_loss'].backward()`` fills
``.grad`` of every parameter on the path, and ``torch.optim.AdamW`` (complete_project.py:640-650 builds exactly that)
steps them.  The uncertainty planes (aleatoric / epistemic / total / calibrated) are returned as values: no loss of
the reference reads them, and ``calibration_layer`` therefore never receives a gradient there either.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops

DIM_NAMES = ("valence", "arousal", "dominance")
HEAD_KEYS = ("mu", "nu", "alpha", "beta", "aleatoric_uncertainty", "epistemic_uncertainty", "uncertainty")


@dataclass
class ModelConfig:
    """Field names and defaults of complete_project.ModelConfig (complete_project.py:33-56)."""
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
    dropout_seed: int = 0        # (not a reference field) seed of the counter-hash dropout; give each data-parallel rank its own


# ---- parameter containers: the module tree (and therefore every state_dict key) of the reference classes.  The
#      activation / dropout slots hold no parameters; they only keep the Sequential indices aligned.
def _slot() -> nn.Module:
    return nn.Identity()


def _residual_block(dim: int) -> nn.Module:
    m = nn.Module()
    m.layers = nn.Sequential(nn.Linear(dim, dim), _slot(), _slot(), nn.LayerNorm(dim))
    return m


def _encoder(in_dim: int, dim: int, layers: int) -> nn.Module:
    m = nn.Module()
    m.input_projection = nn.Sequential(nn.Linear(in_dim, dim), _slot(), nn.LayerNorm(dim))
    m.encoder_layers = nn.ModuleList([_residual_block(dim) for _ in range(layers)])
    m.output_projection = nn.Linear(dim, dim)
    return m


def _mha(dim: int) -> nn.Module:
    m = nn.Module()
    for n in ("query_proj", "key_proj", "value_proj", "output_proj"):
        setattr(m, n, nn.Linear(dim, dim))
    return m


def _attention(dim: int) -> nn.Module:
    m = nn.Module()
    m.self_attention, m.cross_attention = _mha(dim), _mha(dim)
    m.uncertainty_estimator = nn.Module()
    m.uncertainty_estimator.estimator = nn.Sequential(nn.Linear(dim, dim // 2), _slot(), _slot(), nn.Linear(dim // 2, dim // 4),
                                                      _slot(), nn.Linear(dim // 4, 1), _slot())
    m.weight_network = nn.Sequential(nn.Linear(dim * 3 + 3, dim), _slot(), _slot(), nn.Linear(dim, 3), _slot())
    return m


def _fusion(dim: int, fdim: int) -> nn.Module:
    m = nn.Module()
    stage = lambda k: nn.Sequential(nn.Linear(k, fdim), _slot(), _slot(), nn.LayerNorm(fdim), nn.Linear(fdim, fdim), _slot())
    m.av_fusion = stage(2 * dim)
    m.trimodal_fusion = stage(fdim + dim)
    m.fusion_gate = nn.Sequential(nn.Linear(fdim + dim, fdim), _slot())
    return m


def _head(fdim: int, hidden: int = 256) -> nn.Module:
    m = nn.Module()
    m.evidence_network = nn.Sequential(nn.Linear(fdim, hidden), _slot(), _slot(), nn.Linear(hidden, hidden // 2), _slot(), _slot(),
                                       nn.Linear(hidden // 2, 4))
    return m


class _Calibration(nn.Module):
    def __init__(self, dims: int):
        super().__init__()
        self.temperature = nn.Parameter(torch.ones(dims))
        self.calibration_network = nn.Sequential(nn.Linear(1, 32), _slot(), nn.Linear(32, 16), _slot(), nn.Linear(16, 1), _slot())


class CompleteDEERModel(nn.Module):
    """Mirror of ``complete_project.CompleteDEERModel`` (complete_project.py:462-602); see the module docstring."""

    def __init__(self, config: Optional[ModelConfig] = None, compute_dtype: str = "fp32"):
        super().__init__()
        config = config or ModelConfig()
        if (config.encoder_dim, config.fusion_dim, config.emotion_dims) != (256, 512, 3):
            raise NotImplementedError("the HIP row kernels are specialised for encoder_dim=256, fusion_dim=512, emotion_dims=3")
        if config.encoder_dim % config.attention_heads:
            raise ValueError("feature_dim must be divisible by num_heads")          # complete_project.py:126
        for k in (config.audio_dim, config.video_dim, config.text_dim):
            if k % 4:
                raise NotImplementedError("input dimensions must be multiples of 4")
        ops._act_dtype(compute_dtype)
        self.config, self.compute_dtype = config, compute_dtype
        d, f = config.encoder_dim, config.fusion_dim
        self.audio_encoder = _encoder(config.audio_dim, d, config.encoder_layers)
        self.video_encoder = _encoder(config.video_dim, d, config.encoder_layers)
        self.text_encoder = _encoder(config.text_dim, d, config.encoder_layers)
        self.attention_module = _attention(d)
        self.fusion_module = _fusion(d, f)
        self.prediction_heads = nn.ModuleDict({n: _head(f) for n in DIM_NAMES})
        self.calibration_layer = _Calibration(config.emotion_dims)
        self._initialize_weights()
        self._packed = None
        self._train_step = 0          # dropout counter: one tick per training forward
        self._drop_counter = None     # device-side addend of the counter (int64[1]), created by capture_train_step

    def _initialize_weights(self) -> None:
        """Xavier-uniform Linear weights, zero biases, unit LayerNorm (complete_project.py:503-513)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def mark_parameters_changed(self) -> None:
        """Force the next forward to rebuild the operand image (needed only after writes through ``param.data``, which
        bypass the version counters the cache is keyed on)."""
        self._packed = None

    # ---- device-side operand image (include/mmdeer.h: mmdeer_stackb_weights), rebuilt when a parameter changes
    def _pack(self) -> dict:
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self.compute_dtype,)
        if self._packed is not None and self._packed["key"] == key:
            return self._packed
        cfg = self.config
        dt = ops._act_dtype(self.compute_dtype)
        W = lambda *ts: (torch.stack([t.detach() for t in ts]) if len(ts) > 1 else ts[0].detach()).to(dt).contiguous()
        V = lambda *ts: torch.stack([t.detach().float().reshape(-1) for t in ts]).contiguous()
        encs = (self.audio_encoder, self.video_encoder, self.text_encoder)
        keep = {}          # name -> tensor: keeps every image alive while the struct points at it
        sw = _lib.StackBWeights()
        sw.audio_dim, sw.video_dim, sw.text_dim, sw.encoder_layers = cfg.audio_dim, cfg.video_dim, cfg.text_dim, cfg.encoder_layers
        # bf16: 84-wide rows are not 16-byte aligned -> the audio weight image is zero-padded to 128 columns
        sw.audio_ld = cfg.audio_dim if dt == torch.float32 else (cfg.audio_dim + 127) // 128 * 128
        for m, enc in enumerate(encs):
            w = enc.input_projection[0].weight.detach()
            if m == 0 and sw.audio_ld != cfg.audio_dim:
                w = torch.nn.functional.pad(w, (0, sw.audio_ld - cfg.audio_dim))
            keep[f"enc_in_w{m}"] = w.to(dt).contiguous()
            sw.enc_in_w[m] = keep[f"enc_in_w{m}"].data_ptr()
        keep["enc_in_vec"] = V(*[t for e in encs for t in (e.input_projection[0].bias, e.input_projection[2].weight, e.input_projection[2].bias)])
        L = cfg.encoder_layers
        if L:
            keep["enc_res_w"] = W(*[e.encoder_layers[l].layers[0].weight for l in range(L) for e in encs])
            keep["enc_res_vec"] = V(*[t for l in range(L) for e in encs
                                      for t in (e.encoder_layers[l].layers[0].bias, e.encoder_layers[l].layers[3].weight, e.encoder_layers[l].layers[3].bias)])
        keep["enc_out_w"] = W(*[e.output_projection.weight for e in encs])
        keep["enc_out_b"] = V(*[e.output_projection.bias for e in encs])
        att = self.attention_module
        sa, ca = att.self_attention, att.cross_attention
        keep["value_w"], keep["value_b"] = W(sa.value_proj.weight, ca.value_proj.weight), V(sa.value_proj.bias, ca.value_proj.bias)
        keep["attn_out_w"], keep["attn_out_b"] = W(sa.output_proj.weight, ca.output_proj.weight), V(sa.output_proj.bias, ca.output_proj.bias)
        est = att.uncertainty_estimator.estimator
        keep["est_w1"], keep["est_b1"] = W(est[0].weight), V(est[0].bias)
        keep["est_w2"], keep["est_b2"] = W(est[3].weight), V(est[3].bias)
        keep["est_w3"], keep["est_b3"] = V(est[5].weight), V(est[5].bias)
        wn, D3 = att.weight_network, 3 * cfg.encoder_dim
        keep["wn_w1"], keep["wn_b1"] = W(wn[0].weight[:, :D3]), V(wn[0].bias)
        keep["wn_w1_unc"] = wn[0].weight.detach()[:, D3:].float().contiguous()
        keep["wn_w2"], keep["wn_b2"] = wn[3].weight.detach().float().contiguous(), V(wn[3].bias)
        fu = self.fusion_module
        for name, seq in (("av", fu.av_fusion), ("tri", fu.trimodal_fusion)):
            keep[name + "_w0"], keep[name + "_w4"] = W(seq[0].weight), W(seq[4].weight)
            keep[name + "_vec"] = V(seq[0].bias, seq[3].weight, seq[3].bias, seq[4].bias)
        keep["gate_w"], keep["gate_b"] = W(fu.fusion_gate[0].weight), V(fu.fusion_gate[0].bias)
        nets = [self.prediction_heads[n].evidence_network for n in DIM_NAMES]
        for k in (0, 3, 6):
            keep[f"head_w{k}"], keep[f"head_b{k}"] = W(*[n[k].weight for n in nets]), V(*[n[k].bias for n in nets])
        cal = self.calibration_layer
        cn = cal.calibration_network
        keep["cal"] = [t.detach().float().reshape(-1).contiguous() for t in
                       (cal.temperature, cn[0].weight, cn[0].bias, cn[2].weight, cn[2].bias, cn[4].weight, cn[4].bias)]
        for name, _ in _lib.StackBWeights._fields_:
            if name in keep and name != "cal":
                setattr(sw, name, keep[name].data_ptr())
        for i, t in enumerate(keep["cal"]):
            sw.calibration[i] = t.data_ptr()
        self._packed = {"key": key, "keep": keep, "struct": sw}
        return self._packed

    def forward(self, audio_features: torch.Tensor, video_features: torch.Tensor, text_features: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self.training:
            return self.forward_train(audio_features, video_features, text_features, dropout=True)
        with torch.no_grad():
            return self._forward_eval(audio_features, video_features, text_features)

    def _check_inputs(self, xs):
        cfg = self.config
        ops._check_dev(*xs)
        B = xs[0].shape[0]
        for x, k in zip(xs, (cfg.audio_dim, cfg.video_dim, cfg.text_dim)):
            if x.dim() != 2 or x.shape != (B, k):
                raise ValueError(f"expected features of shape ({B}, {k}), got {tuple(x.shape)}")
        return B

    def forward_train(self, audio_features: torch.Tensor, video_features: torch.Tensor, text_features: torch.Tensor,
                      dropout: bool = True) -> Dict[str, torch.Tensor]:
        """Differentiable forward (what ``forward`` runs in ``.train()`` mode).  ``dropout=False`` keeps every site open --
        the reference in ``.eval()`` mode with gradients, which is how the golden gradients were taken."""
        from . import stackb_train
        xs = (audio_features, video_features, text_features)
        B = self._check_inputs(xs)
        if B == 0:
            raise ValueError("training forward on an empty batch")
        xs = [x.detach().float().contiguous() for x in xs]
        drop = None
        if dropout:
            # mask key = hash(seed, step, site, row, column).  Eager forwards hash the host-side step (and advance it); inside a
            # captured step (capture_train_step) the step lives in a device counter the graph bumps itself, with 0 frozen in
            # as the host part -- replay() keeps the two in line, so a replay / eager / replay sequence draws three
            # different masks (the old scheme added host + device and could repeat an offset)
            if getattr(self, "_in_graph_step", False):
                drop = (self.config.dropout, int(self.config.dropout_seed), 0, self._drop_counter)
            else:
                drop = (self.config.dropout, int(self.config.dropout_seed), self._train_step, None)
                self._train_step += 1
        names = [n for n, _ in self.named_parameters()]
        core, tape = _StackBFn.apply(self, xs, drop, names, *[p for _, p in self.named_parameters()])
        planes = tape["planes"]
        out: Dict[str, torch.Tensor] = {}
        for d, name in enumerate(DIM_NAMES):
            for k, key in enumerate(HEAD_KEYS):
                out[f"{name}_{key}"] = core[k, :, d] if k < 4 else planes[k, :, d]
        out["mu_all"], out["uncertainty_all"], out["calibrated_uncertainty"] = core[0], planes[6], planes[7]
        out["attention_weights"], out["modality_uncertainties"] = tape["w4"][:, :3], tape["u4"][:, :3]
        out["fused_features"] = tape["fused32"]
        return out

    def compute_loss(self, predictions: Dict[str, torch.Tensor], targets: torch.Tensor) -> Dict[str, torch.Tensor]:
        """The trainer hook (training.py:210): ``MultiTaskDEERLoss`` defaults on this model's ``{dim}_{mu|nu|alpha|beta}``
        keys (src/utils/losses.py:286-291 reads exactly those)."""
        from .model import multitask_deer_loss
        return multitask_deer_loss(predictions, targets)

    def train_step(self, audio, video, text, targets) -> Dict[str, torch.Tensor]:
        """``compute_loss(model(a, v, t), y)['total_loss'].backward()`` -- gradients accumulate into ``.grad``."""
        was = self.training
        self.train()
        try:
            loss = self.compute_loss(self(audio, video, text), targets)
            loss["total_loss"].backward()
        finally:
            self.train(was)
        return loss

    def capture_train_step(self, audio, video, text, targets, optimizer: Optional[torch.optim.Optimizer] = None, max_grad_norm: Optional[float] = None):
        """Capture forward (dropout live) + ``compute_loss`` + backward (+ ``clip_grad_norm_`` + ``optimizer.step()`` when an optimiser
        built with ``capturable=True`` is given) for this batch shape as ONE HIP graph: the ~150 operator launches and the autograd
        bookkeeping replay without host work.  Returns ``replay(audio, video, text, targets) -> loss dict`` (static tensors: clone what
        must outlive the next replay); ``.grad`` of the parameters are the graph's static gradient buffers.  Every replay draws fresh
        dropout masks: a device counter, bumped inside the graph, is added to the step the library hashes."""
        dev = audio.device
        static = [x.detach().float().contiguous().clone() for x in (audio, video, text, targets)]
        # the device counter holds the step the LAST graph-side forward used; host invariant: _train_step == counter + 1
        self._drop_counter = torch.full((1,), int(self._train_step) - 1, dtype=torch.int64, device=dev)
        counter = self._drop_counter
        was = self.training
        self.train()

        def step():
            counter.add_(1)
            self._drop_counter, self._in_graph_step = counter, True
            try:
                loss = self.compute_loss(self(*static[:3]), static[3])
            finally:
                self._in_graph_step = False
            loss["total_loss"].backward()
            if optimizer is not None:
                if max_grad_norm is not None:
                    torch.nn.utils.clip_grad_norm_(self.parameters(), max_grad_norm)
                optimizer.step()
            return loss

        if optimizer is not None:
            optimizer.zero_grad(set_to_none=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                                # warm-up: library load, allocator pools, optimiser state
                for p in self.parameters():
                    p.grad = None
                step()
                self._train_step += 1                         # each warm-up step used one step of the stream
        torch.cuda.current_stream().wait_stream(side)
        for p in self.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = step()                                     # recorded, not executed: the counter is not advanced here
        self.train(was)
        shadow = [int(self._train_step) - 1]                  # what the device counter holds

        def replay(a=None, v=None, t=None, y=None):
            for dst, src in zip(static, (a, v, t, y)):
                if src is not None:
                    dst.copy_(src, non_blocking=True)
            if int(self._train_step) - 1 != shadow[0]:        # eager forwards in between advanced the host step only
                counter.fill_(int(self._train_step) - 1)
            graph.replay()                                    # uses step == _train_step (counter + 1)
            self._train_step += 1
            shadow[0] = int(self._train_step) - 1
            return loss

        replay.graph, replay.static_inputs, replay.loss, replay.counter = graph, static, loss, counter
        return replay

    # ---- fused training step: flat parameter / gradient / compute-dtype buffers, no autograd, no casts
    def _flat(self, dev: torch.device) -> dict:
        """Move the parameters into ONE flat fp32 buffer (each ``nn.Parameter`` becomes a view at a 64-element-aligned offset,
        values unchanged), with a flat gradient buffer of the same layout (``.grad`` of every parameter is a view of it) and
        a flat compute-dtype copy the GEMMs read (maintained by ``optim.FlatAdamW`` / refreshed by one ``mmdeer_convert``).
        Idempotent; redone if the parameters were moved (``.to``) or re-created."""
        st = getattr(self, "_flat_state", None)
        named = list(self.named_parameters())
        if st is not None and st["dev"] == dev and all(p.data_ptr() == q for (_, p), q in zip(named, st["ptrs"])):
            return st
        offs, cur = {}, 0
        for n, p in named:
            offs[n] = cur
            cur += (p.numel() + 63) // 64 * 64
        flat_p = torch.zeros(cur, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(cur, dtype=torch.float32, device=dev)
        by_id, gview = {}, {}
        for n, p in named:
            v = flat_p[offs[n]:offs[n] + p.numel()].view(p.shape)
            v.copy_(p.detach().to(dev))
            p.data = v
            gview[n] = flat_g[offs[n]:offs[n] + p.numel()].view(p.shape)
            p.grad = gview[n]
            by_id[id(p)] = (offs[n], p.numel())
        f32 = self.compute_dtype == "fp32"
        packed = flat_p if f32 else torch.empty(cur, dtype=torch.bfloat16, device=dev)
        # transposed compute-dtype copies of the matrices whose dX the backward needs, at the same offsets; the three operands
        # that are concatenations / a column slice of parameters get areas of their own behind the last parameter
        att, fu, cfgm = self.attention_module, self.fusion_module, self.config
        sa, ca, wn = att.self_attention, att.cross_attention, att.weight_network
        nets = [self.prediction_heads[n].evidence_network for n in DIM_NAMES]
        E_, D3 = cfgm.encoder_dim, 3 * cfgm.encoder_dim
        extra, tab = {}, []                       # tab: (parameter, dst_off, ld_dst, dst_col)
        def area(key, elems):
            nonlocal cur_t
            extra[key] = cur_t
            cur_t += (elems + 63) // 64 * 64
        cur_t = cur
        area("wv", E_ * 2 * E_); area("wn1", (D3 + 3) * E_); area("wh0", cfgm.fusion_dim * 3 * 256)
        plain = [e.output_projection.weight for e in (self.audio_encoder, self.video_encoder, self.text_encoder)]
        plain += [blk.layers[0].weight for e in (self.audio_encoder, self.video_encoder, self.text_encoder) for blk in e.encoder_layers]
        est = att.uncertainty_estimator.estimator
        plain += [sa.output_proj.weight, ca.output_proj.weight, est[0].weight, est[3].weight, fu.av_fusion[0].weight, fu.av_fusion[4].weight,
                  fu.trimodal_fusion[0].weight, fu.trimodal_fusion[4].weight, fu.fusion_gate[0].weight] + [n[3].weight for n in nets]
        for w in plain:
            tab.append((w, by_id[id(w)][0], 0, 0))
        tab.append((sa.value_proj.weight, extra["wv"], 2 * E_, 0)); tab.append((ca.value_proj.weight, extra["wv"], 2 * E_, E_))
        tab.append((wn[0].weight, extra["wn1"], 0, 0))
        for d, n in enumerate(nets):
            tab.append((n[0].weight, extra["wh0"], 3 * 256, d * 256))
        packed_t = torch.empty(cur_t, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        nt = len(tab)
        vp = C.c_void_p
        tt = {"n": nt, "src": (vp * nt)(*[w.data_ptr() for w, _, _, _ in tab]), "rows": (C.c_int32 * nt)(*[w.shape[0] for w, _, _, _ in tab]),
              "cols": (C.c_int32 * nt)(*[w.shape[1] for w, _, _, _ in tab]), "off": (C.c_longlong * nt)(*[o for _, o, _, _ in tab]),
              "ld": (C.c_int32 * nt)(*[l for _, _, l, _ in tab]), "col": (C.c_int32 * nt)(*[c for _, _, _, c in tab])}
        st = {"dev": dev, "offs": offs, "n": cur, "p": flat_p, "g": flat_g, "packed": packed, "by_id": by_id, "gview": gview,
              "ptrs": [p.data_ptr() for _, p in named], "versions": None, "names": [n for n, _ in named],
              "packed_t": packed_t, "extra_t": extra, "ttab": tt}
        st["frag"] = None
        if not f32:
            from . import stackb_train
            st["frag"] = stackb_train.build_frag_images(self, st)
        self._flat_state = st
        self._packed = None
        return st

    def _flat_refresh(self, st: dict) -> None:
        """The compute-dtype copy follows the parameters: after an update by anything but FlatAdamW (which writes it itself)
        one conversion of the whole buffer."""
        ver = tuple(p._version for p in self.parameters())
        if st["versions"] == ver:
            return
        if st["packed"] is not st["p"]:
            lib = _lib.load()
            _lib.check(lib.mmdeer_convert(st["p"].data_ptr(), 1, st["packed"].data_ptr(), 0, st["n"], _lib.current_stream()))
        self._flat_pack_t(st)
        st["versions"] = ver

    def _flat_pack_t(self, st: dict) -> None:
        """The transposed copies follow the parameters: one launch (``mmdeer_pack_transposed_batch``)."""
        t = st["ttab"]
        _lib.check(_lib.load().mmdeer_pack_transposed_batch(t["n"], t["src"], t["rows"], t["cols"], st["packed_t"].data_ptr(), t["off"], t["ld"],
                                                            t["col"], int(st["packed"] is st["p"]), _lib.current_stream()))
        if st.get("frag") is not None:       # bf16: the fragment-major images the layer chains stream (stackb_train.py), from the copies above
            st["frag"].refresh()

    def train_step_fused(self, audio, video, text, targets) -> Dict[str, torch.Tensor]:
        """forward (dropout live) + ``MultiTaskDEERLoss`` + backward as library launches only: the same operator sequence as
        ``compute_loss(model(a, v, t), y)['total_loss'].backward()`` (same kernels, same gradients) without autograd
        bookkeeping, per-step weight casts or gradient copies -- the gradients land in the flat buffer ``.grad`` views of
        which the parameters hold.  Follow with ``optim.FlatAdamW(model).step()``."""
        from . import stackb_train
        from .model import loss_dict_from, make_loss_cfg
        xs = (audio, video, text)
        B = self._check_inputs(xs)
        if B == 0:
            raise ValueError("training step on an empty batch")
        dev = audio.device
        st = self._flat(dev)
        self._flat_refresh(st)
        # bf16 compute: the feature blocks are rounded to bf16 once here (the GEMM loaders would round them on the fly anyway)
        # so that the input-projection weight gradients run on the LDS-DMA kernel with the rest of their group
        xdt = torch.float32 if self.compute_dtype == "fp32" else torch.bfloat16
        lean = self.compute_dtype == "bf16"       # (both plans of the bf16 step: the same weight-gradient kernel for the input projection)
        xs = [x.detach() if (lean and x.shape[1] % 64) else x.detach().to(xdt).contiguous() for x in xs]
        if lean:
            # bf16 step: rows of a width the chain's DMA cannot take (the 84-wide audio block) live in a persistent buffer
            # zero-padded to a multiple of 64 columns -- one converting copy per step, and the input projection's weight gradient
            # reads 16-byte aligned rows
            for i, x in enumerate(xs):
                K0 = (x.shape[1] + 63) // 64 * 64
                if K0 != x.shape[1]:
                    pad = st.get(("xpad", i))
                    if pad is None or pad.shape[0] != B:
                        pad = st[("xpad", i)] = torch.zeros(B, K0, dtype=xdt, device=dev)
                    pad[:, :x.shape[1]].copy_((audio, video, text)[i].detach())
                    xs[i] = pad[:, :x.shape[1]]
        if getattr(self, "_in_graph_step", False):
            drop = (self.config.dropout, int(self.config.dropout_seed), 0, self._drop_counter)
        else:
            drop = (self.config.dropout, int(self.config.dropout_seed), self._train_step, None)
            self._train_step += 1
        T = stackb_train.forward_train(self, xs, drop, flat=st)
        planes = T["planes"]
        lib = _lib.load()
        f32o = dict(dtype=torch.float32, device=dev)
        g4, loss_out = torch.empty(4, B, 3, **f32o), torch.empty(20, **f32o)
        bins = torch.empty(30, dtype=torch.int32, device=dev)
        stats = torch.empty(int(lib.mmdeer_nig_stats_elems(B)), **f32o)
        y = targets.detach().float().contiguous()
        cfg = make_loss_cfg()
        _lib.check(lib.mmdeer_nig_loss(planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), planes[3].data_ptr(), y.data_ptr(),
                                       stats.data_ptr(), g4[0].data_ptr(), g4[1].data_ptr(), g4[2].data_ptr(), g4[3].data_ptr(),
                                       loss_out.data_ptr(), bins.data_ptr(), B, C.byref(cfg), _lib.current_stream()))
        stackb_train.backward(self, T, g4, flat=st)
        d = loss_dict_from(loss_out, B)
        d["ece_bin_counts"] = bins.view(3, 10)
        d["_planes"] = planes
        d["_keep"] = (T, g4, stats, y)
        return d

    def capture_train_step_fused(self, audio, video, text, targets, warmup: int = 2):
        """``train_step_fused`` for this batch shape as ONE HIP graph (fresh dropout masks per replay through the device
        counter).  ``replay(a, v, t, y)`` copies new data into the static inputs and returns the (static) loss dict; the
        optimiser step (``optim.FlatAdamW.step``: two launches) follows eagerly, as for Stack C.  The capture is preceded by
        ``warmup`` eager steps on this batch (each a real training step with the dropout step it would have had); the last
        one's loss dict is ``replay.first`` and its gradients are in the flat buffer -- a trainer that captures on the first batch of
        a shape (``warmup=1``) takes them as that batch's step, and the replays continue the same mask stream."""
        dev = audio.device
        # feature blocks handed over in the compute dtype stay in it (bf16 blocks resident in HBM: no conversion launch inside the graph)
        keep = torch.bfloat16 if self.compute_dtype == "bf16" else None
        static = [x.detach().contiguous().clone() if x.dtype == keep else x.detach().float().contiguous().clone() for x in (audio, video, text)]
        static.append(targets.detach().float().contiguous().clone())
        self._drop_counter = torch.full((1,), int(self._train_step) - 1, dtype=torch.int64, device=dev)
        counter = self._drop_counter
        was = self.training
        self.train()

        def step():
            counter.add_(1)
            self._drop_counter, self._in_graph_step = counter, True
            try:
                return self.train_step_fused(*static)
            finally:
                self._in_graph_step = False

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        first = None
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                first = step()
                self._train_step += 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = step()
        self.train(was)
        shadow = [int(self._train_step) - 1]
        st = self._flat(dev)

        def replay(a=None, v=None, t=None, y=None):
            for dst, src in zip(static, (a, v, t, y)):
                if src is not None:
                    dst.copy_(src, non_blocking=True)
            self._flat_refresh(st)
            if int(self._train_step) - 1 != shadow[0]:
                counter.fill_(int(self._train_step) - 1)
            graph.replay()
            self._train_step += 1
            shadow[0] = int(self._train_step) - 1
            return loss

        replay.graph, replay.static_inputs, replay.loss, replay.counter, replay.first = graph, static, loss, counter, first
        return replay

    def _forward_eval(self, audio_features, video_features, text_features) -> Dict[str, torch.Tensor]:
        cfg = self.config
        xs = (audio_features, video_features, text_features)
        B = self._check_inputs(xs)
        xs = [x.detach().float().contiguous() for x in xs]
        dev = xs[0].device
        P = self._pack()
        lib = _lib.load()
        f32 = int(self.compute_dtype == "fp32")
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        planes, weights, unc, fused = new(8, B, 3), new(B, 3), new(B, 3), new(B, cfg.fusion_dim)
        if B:
            nbytes = lib.mmdeer_stackb_workspace_bytes(B, f32, P["struct"].audio_ld)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            a = _lib.StackBForwardArgs()
            a.batch, a.compute_f32 = B, f32
            a.audio, a.video, a.text = (x.data_ptr() for x in xs)
            a.weights = C.pointer(P["struct"])
            a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
            a.planes, a.attention_weights, a.modality_uncertainties = planes.data_ptr(), weights.data_ptr(), unc.data_ptr()
            a.fused_features = fused.data_ptr()
            a.stream = _lib.current_stream()
            _lib.check(lib.mmdeer_stackb_forward(C.byref(a)))

        out: Dict[str, torch.Tensor] = {}
        for d, name in enumerate(DIM_NAMES):
            for k, key in enumerate(HEAD_KEYS):
                out[f"{name}_{key}"] = planes[k, :, d]
        out["mu_all"], out["uncertainty_all"], out["calibrated_uncertainty"] = planes[0], planes[6], planes[7]
        out["attention_weights"], out["modality_uncertainties"] = weights, unc
        out["fused_features"] = fused
        return out

    def capture(self, audio_features: torch.Tensor, video_features: torch.Tensor, text_features: torch.Tensor):
        """Capture the forward for this batch shape as a HIP graph (the 25 launches replay without host work).

        Returns ``replay(audio, video, text) -> outputs``: the inputs are copied into the graph's static buffers and the
        returned dictionary is the same set of static tensors every call (clone what must outlive the next replay).
        Parameters are baked in as packed at capture time; capture again after changing them."""
        static = [x.detach().float().contiguous().clone() for x in (audio_features, video_features, text_features)]
        self._pack()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.forward(*static)                      # warm-up: library load, workspace pool
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outputs = self.forward(*static)

        def replay(audio, video, text):
            for dst, src in zip(static, (audio, video, text)):
                dst.copy_(src, non_blocking=True)
            graph.replay()
            return outputs

        replay.graph, replay.outputs, replay.static_inputs = graph, outputs, static
        return replay

    def get_predictions_and_uncertainties(self, outputs: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """(mu_all, calibrated_uncertainty or uncertainty_all) -- complete_project.py:591-602."""
        return outputs["mu_all"], outputs.get("calibrated_uncertainty", outputs["uncertainty_all"])


class _StackBFn(torch.autograd.Function):
    """(mu, nu, alpha, beta) of a training forward; backward = stackb_train.backward.  The parameters are inputs so that
    autograd accumulates their gradients; the tape (saved activations) rides on ctx."""

    @staticmethod
    def forward(ctx, model, xs, drop, names, *params):
        from . import stackb_train
        tape = stackb_train.forward_train(model, xs, drop)
        ctx.model, ctx.tape, ctx.names = model, tape, names
        ctx.set_materialize_grads(False)
        core = tape["planes"][:4].clone()
        ctx.mark_non_differentiable()
        return core, _Tape(tape)

    @staticmethod
    def backward(ctx, g_core, _g_tape):
        from . import stackb_train
        if g_core is None:
            return (None,) * (4 + len(ctx.names))
        G = stackb_train.backward(ctx.model, ctx.tape, g_core.float().contiguous())
        params = dict(ctx.model.named_parameters())
        grads = []
        for n in ctx.names:
            g = G.get(n)
            grads.append(None if g is None else g.to(params[n].dtype).view_as(params[n]))
        ctx.tape = None
        return (None, None, None, None) + tuple(grads)


class _Tape(dict):
    """The forward's saved tensors as a (non-tensor) second output of the autograd node."""


def create_complete_deer_model(config: Optional[ModelConfig] = None, compute_dtype: str = "fp32") -> CompleteDEERModel:
    """Factory of complete_project.py:605-632 (without its parameter-count printout)."""
    return CompleteDEERModel(config or ModelConfig(), compute_dtype=compute_dtype)
