"""Stack B training path: forward with dropout and backward of ``complete_project.CompleteDEERModel``
(reference src/models/complete_project.py:462-602) as a sequence of C-ABI operator calls.

This file is host logic only -- it owns buffers and the ORDER of the launches; every number is produced by
``libmmdeer_hip.so``: the Linear layers by ``mmdeer_gemm`` (forward with bias / ReLU / hash dropout in the epilogue;
``dX = dY W`` with the ``(Y > 0) / (1 - p)`` mask of the layer below in the epilogue; ``dW = dY^T X`` with the bias gradient
from the same launch), LayerNorm by ``mmdeer_layernorm_fwd`` / ``_bwd`` (the backward folds the ReLU + dropout mask of the
``Linear-ReLU-Dropout-LayerNorm`` blocks, complete_project.py:65-70, 315-333), and the row operators of
``csrc/stackb_train.hip`` (attention tail forward / backward, gate mix backward, softplus backward, masked adds).

Dropout sites (all ``nn.Dropout``; masks are the library's counter hash keyed by (seed, step, site, row, column), so the
backward regenerates them): ResidualBlock (:68), MultiHeadAttention attention weights (:141, 172 -- one decision per (row,
head) because every softmax is over a single key), UncertaintyEstimator (:186, p = 0.2 whatever the config says),
weight_network (:237), the two fusion stages (:318, 328) and the prediction heads (:379, 382).

Two launch plans write the same tape and the same gradients, bit for bit (tests/test_gpu_stackb.py): launch by launch as described above
(``model.train_plan = 'ops'``; fp32; geometries the chain kernel does not instantiate), and -- the default of the bf16 fused step -- every
sample-local run of layers as ONE launch of the layer-chain kernel (``mmdeer_chain``, csrc/chain.hip; host side ``chainops.py``): the three
encoders, the attention blocks' value / output projections, the estimator, the two fusion stages, the heads' first two layers, and the dX
run of each of them: 16 chain launches, 49 launches per step instead of ~170.

Layout conventions are those of the inference executor (csrc/stackb.hip): encoder outputs in column blocks of one (B, 768)
matrix whose (3B, 256) reading is the row set of the shared-weight attention layers; heads stacked along N.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .chainops import K_OK, Chain, FragImages
from .opseq import Exec, _ptr

ENC, FUS, HID = 256, 512, 256
SITE_RES, SITE_ATTN_S, SITE_ATTN_C, SITE_EST, SITE_WN, SITE_AV, SITE_TRI, SITE_H0, SITE_H3 = 32, 64, 65, 66, 67, 68, 69, 70, 71     # H3: 71..73


def build_frag_images(model, st) -> Optional[FragImages]:
    """Fragment-major images of every matrix the layer chains of the bf16 training step stream (W for the forward runs, W^T for
    the dX runs), sourced from the flat bf16 copy / the transposed copies the optimiser step maintains.  None when the model's
    geometry is outside what the chain kernel instantiates (the step then runs launch by launch)."""
    cfg = model.config
    L = cfg.encoder_layers
    if cfg.encoder_dim != ENC or cfg.fusion_dim != FUS or L > 4 or cfg.audio_dim > 128 or cfg.audio_dim % 2:
        return None
    if cfg.video_dim not in K_OK or cfg.text_dim not in K_OK or cfg.attention_heads != 8 or cfg.emotion_dims != 3:
        return None
    packed, packed_t, by_id, extra = st["packed"], st["packed_t"], st["by_id"], st["extra_t"]
    F = FragImages(st["dev"])

    def W(t):
        off, n = by_id[id(t)]
        return packed[off:off + n].view(t.shape)

    def both(key, t):
        F.add(key, W(t), t.shape[0], t.shape[1])
        F.add(key + ".T", W(t), t.shape[0], t.shape[1], transpose=1)

    for m, e in enumerate((model.audio_encoder, model.video_encoder, model.text_encoder)):
        w0 = e.input_projection[0].weight
        K0 = (w0.shape[1] + 63) // 64 * 64
        F.add(f"enc{m}.w0", W(w0), ENC, K0, ld_src=w0.shape[1], cols_valid=w0.shape[1])
        for l, blk in enumerate(e.encoder_layers):
            both(f"enc{m}.res{l}", blk.layers[0].weight)
        both(f"enc{m}.wo", e.output_projection.weight)
    att, fu = model.attention_module, model.fusion_module
    sa, ca, est = att.self_attention, att.cross_attention, att.uncertainty_estimator.estimator
    F.add("wv_s", W(sa.value_proj.weight), ENC, ENC); F.add("wv_c", W(ca.value_proj.weight), ENC, ENC)
    both("wos", sa.output_proj.weight); both("woc", ca.output_proj.weight)
    both("we1", est[0].weight); both("we2", est[3].weight)
    for name, seq in (("av", fu.av_fusion), ("tri", fu.trimodal_fusion)):
        both(name + ".w0", seq[0].weight); both(name + ".w4", seq[4].weight)
    for d, nm in enumerate(("valence", "arousal", "dominance")):
        net = model.prediction_heads[nm].evidence_network
        F.add(f"wh0.{d}", W(net[0].weight), HID, FUS)
        both(f"wh3.{d}", net[3].weight)
    # W^T of the stacked operands, from the transposed row-major copies (their areas behind the last parameter)
    F.add("wv.T", packed_t[extra["wv"]:extra["wv"] + ENC * 2 * ENC].view(ENC, 2 * ENC), ENC, 2 * ENC)
    F.add("wh0.T", packed_t[extra["wh0"]:extra["wh0"] + FUS * 3 * HID].view(FUS, 3 * HID), FUS, 3 * HID)
    # row-major restatements the remaining GEMM launches read: the feature columns of weight_network.0 (a column slice of a
    # 771-wide matrix), the heads' last layers zero-padded from 4 to 8 rows
    wn0 = att.weight_network[0].weight
    F.area("wn1", ENC, 3 * ENC)
    F.place("wn1", W(wn0), ENC, 3 * ENC, ld_src=wn0.shape[1])
    F.area("wh6bd", 24, 3 * HID // 2)          # the three last head layers block-diagonally: d H3 of all heads in one dX launch
    for d, nm in enumerate(("valence", "arousal", "dominance")):
        w6 = model.prediction_heads[nm].evidence_network[6].weight
        F.area(f"wh6p.{d}", 8, HID // 2)
        F.place(f"wh6p.{d}", W(w6), 4, HID // 2)
        F.place("wh6bd", W(w6), 4, HID // 2, row0=8 * d, col0=d * (HID // 2))
    F.finish()
    return F


def chain_plan(model, flat, dt) -> Optional[FragImages]:
    """The fragment-major weight images when this step runs its sample-local layer runs as chains: the fused bf16 step on a
    geometry the chain kernel instantiates, unless ``model.train_plan == 'ops'`` (the launch-by-launch sequence, kept as the
    reference the chains are tested against bit for bit)."""
    if flat is None or dt != torch.bfloat16 or getattr(model, "train_plan", "auto") == "ops":
        return None
    return flat.get("frag")


def _params(model, dt, flat=None, Fg=None):
    """Compute-dtype matrices and fp32 vectors, keyed like the module tree.  Default: a cast per matrix per step (torch here
    is memory plumbing, the casts carry no arithmetic of the path).  With ``flat`` (CompleteDEERModel._flat: the fused
    training step) the matrices are VIEWS of the flat compute-dtype copy the optimiser step maintains -- no cast at all; only
    the few operands that are concatenations / column slices of parameters still cost a small copy."""
    if flat is None:
        Wp = lambda t: t.detach().to(dt).contiguous()
    else:
        def Wp(t):
            off, n = flat["by_id"][id(t)]
            return flat["packed"][off:off + n].view(t.shape)
    V = lambda t: t.detach().float().contiguous()
    catW = lambda ts: torch.cat([Wp(t) for t in ts], 0)
    catV = lambda ts: torch.cat([V(t) for t in ts], 0)
    P = {"enc": []}
    cfg = model.config
    for e in (model.audio_encoder, model.video_encoder, model.text_encoder):
        d = {"w0": Wp(e.input_projection[0].weight), "b0": V(e.input_projection[0].bias), "g0": V(e.input_projection[2].weight),
             "be0": V(e.input_projection[2].bias), "res": [], "wo": Wp(e.output_projection.weight), "bo": V(e.output_projection.bias)}
        for blk in e.encoder_layers:
            d["res"].append({"w": Wp(blk.layers[0].weight), "b": V(blk.layers[0].bias), "g": V(blk.layers[3].weight), "be": V(blk.layers[3].bias)})
        P["enc"].append(d)
    att = model.attention_module
    sa, ca, est, wn = att.self_attention, att.cross_attention, att.uncertainty_estimator.estimator, att.weight_network
    if Fg is None:
        P["wv"] = catW([sa.value_proj.weight, ca.value_proj.weight])                       # (512, 256)
        P["bv"] = catV([sa.value_proj.bias, ca.value_proj.bias])
    P["bvs"], P["bvc"] = V(sa.value_proj.bias), V(ca.value_proj.bias)
    P["wos"], P["bos"], P["woc"], P["boc"] = Wp(sa.output_proj.weight), V(sa.output_proj.bias), Wp(ca.output_proj.weight), V(ca.output_proj.bias)
    P["we1"], P["be1"], P["we2"], P["be2"] = Wp(est[0].weight), V(est[0].bias), Wp(est[3].weight), V(est[3].bias)
    P["we3"], P["be3"] = V(est[5].weight).reshape(-1), V(est[5].bias)
    D3 = 3 * cfg.encoder_dim
    # chain plan (Fg): the operands that are slices / concatenations / paddings of parameters are images the optimiser step
    # maintains (build_frag_images) or strided views -- no per-step copies
    P["wn1"], P["bn1"] = (Fg.mat("wn1") if Fg is not None else Wp(wn[0].weight)[:, :D3].contiguous()), V(wn[0].bias)
    P["wn1u"] = wn[0].weight.detach()[:, D3:] if Fg is not None else wn[0].weight.detach()[:, D3:].float().contiguous()   # (256, 3) fp32
    P["wn2"], P["bn2"] = V(wn[3].weight), V(wn[3].bias)
    fu = model.fusion_module
    for name, seq in (("av", fu.av_fusion), ("tri", fu.trimodal_fusion)):
        P[name] = {"w0": Wp(seq[0].weight), "b0": V(seq[0].bias), "g": V(seq[3].weight), "be": V(seq[3].bias), "w4": Wp(seq[4].weight), "b4": V(seq[4].bias)}
    P["wg"], P["bg"] = Wp(fu.fusion_gate[0].weight), V(fu.fusion_gate[0].bias)
    nets = [model.prediction_heads[n].evidence_network for n in ("valence", "arousal", "dominance")]
    if Fg is None:
        P["wh0"], P["bh0"] = catW([n[0].weight for n in nets]), catV([n[0].bias for n in nets])                        # (768, 512)
    P["bh0s"] = [V(n[0].bias) for n in nets]
    P["wh3"], P["bh3"] = [Wp(n[3].weight) for n in nets], [V(n[3].bias) for n in nets]
    P["wh6"], P["bh6"] = [Wp(n[6].weight) for n in nets], [V(n[6].bias) for n in nets]
    P["wh6p"] = ([Fg.mat(f"wh6p.{d}") for d in range(3)] if Fg is not None else
                 [torch.nn.functional.pad(w, (0, 0, 0, 4)) for w in P["wh6"]])             # (8, 128): rows 4..7 zero (8-column gradient blocks)
    if flat is not None and flat.get("packed_t") is not None:
        # transposed copies (maintained with the compute-dtype copy): dX = dY W runs as an NT GEMM on the LDS-DMA kernel
        def Wt(t):
            off, n = flat["by_id"][id(t)]
            return flat["packed_t"][off:off + n].view(t.shape[1], t.shape[0])
        X = lambda key, r, c: flat["packed_t"][flat["extra_t"][key]:flat["extra_t"][key] + r * c].view(r, c)
        encs = (model.audio_encoder, model.video_encoder, model.text_encoder)
        P["t"] = {"enc": [{"wo": Wt(e.output_projection.weight), "res": [Wt(blk.layers[0].weight) for blk in e.encoder_layers]} for e in encs],
                  "wv": X("wv", cfg.encoder_dim, 2 * cfg.encoder_dim), "wos": Wt(sa.output_proj.weight), "woc": Wt(ca.output_proj.weight),
                  "we1": Wt(est[0].weight), "we2": Wt(est[3].weight), "wn1": X("wn1", D3 + 3, cfg.encoder_dim)[:D3],
                  "av": {"w0": Wt(fu.av_fusion[0].weight), "w4": Wt(fu.av_fusion[4].weight)},
                  "tri": {"w0": Wt(fu.trimodal_fusion[0].weight), "w4": Wt(fu.trimodal_fusion[4].weight)},
                  "wg": Wt(fu.fusion_gate[0].weight), "wh0": X("wh0", cfg.fusion_dim, 3 * HID), "wh3": [Wt(n[3].weight) for n in nets]}
    return P


def forward_train(model, xs: List[torch.Tensor], drop, flat=None) -> Dict:
    """Forward with every intermediate the backward needs kept on a tape.  Returns the tape (incl. planes (8, B, 3))."""
    ex = Exec(model.compute_dtype, drop)
    dt, f32, dev = ex.dt, ex.f32, xs[0].device
    cfg = model.config
    pc = cfg.dropout
    B = xs[0].shape[0]
    # bf16 fused step: the sample-local runs of layers go through the layer-chain kernel (mmdeer_chain), ONE launch per run, writing
    # the same tape the launch-by-launch sequence below writes
    Fg = chain_plan(model, flat, dt)
    P = _params(model, dt, flat, Fg)
    new = lambda *s, d=None: torch.empty(*s, dtype=d or dt, device=dev)
    T: Dict = {"P": P, "B": B, "xs": xs, "ex": ex}
    E = new(B, 3 * ENC)
    T["enc"] = []
    T["chain"] = Fg is not None
    pch = ex.p_of(pc)
    stat = lambda n: (torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev))
    for m, (x, pe) in enumerate(zip(xs, P["enc"])):
        if Fg is None:
            break
        # encoder m: Linear-ReLU-LayerNorm stem, the residual blocks x + LayerNorm(Dropout(ReLU(Linear x))), output projection
        # (complete_project.py:77-118) -- L + 2 layers, one launch
        K = x.shape[1]
        K0 = (K + 63) // 64 * 64
        xin = x
        if x.stride(0) != K0:                                            # the 84-wide audio rows: zero-padded copy (memory plumbing;
            xin = torch.zeros(B, K0, dtype=dt, device=dev)               # train_step_fused hands over rows that are padded already)
            xin[:, :K].copy_(x)
            x = xin[:, :K]
        t = {"xin": x, "y0": new(B, ENC), "y": [], "st": [], "h": [new(B, ENC)]}
        t["m0"], t["r0"] = stat(B)
        ch = Chain(ex, xin, xin.stride(0), K0, B, p=pch, ts=16 if K0 > 512 else 0)
        ch.seg(Fg(f"enc{m}.w0"), ENC, K0, bias=pe["b0"], relu=1).end(ENC, stash=t["y0"], ld_stash=ENC, ln=(pe["g0"], pe["be0"], t["h"][0], t["m0"], t["r0"]))
        for l, pr in enumerate(pe["res"]):
            y, hn, (mean, rstd) = new(B, ENC), new(B, ENC), stat(B)
            ch.seg(Fg(f"enc{m}.res{l}"), ENC, ENC, bias=pr["b"], relu=1, site=SITE_RES + 3 * l + m)
            ch.end(ENC, stash=y, ld_stash=ENC, ln=(pr["g"], pr["be"], hn, mean, rstd), residual=1)
            t["y"].append(y); t["st"].append((mean, rstd)); t["h"].append(hn)
        ch.seg(Fg(f"enc{m}.wo"), ENC, ENC, bias=pe["bo"]).end(ENC, stash=E[:, m * ENC:(m + 1) * ENC], ld_stash=3 * ENC)
        ch.launch()
        T["enc"].append(t)
    for m, (x, pe) in enumerate(zip(xs, P["enc"])):
        if Fg is not None:
            break
        t = {}
        y0 = new(B, ENC)
        # inputs are read as fp32 (the loader's dtype) and converted while staging; the 84-wide bf16 audio weight rows are
        # 8-byte aligned, which the GEMM's narrow-vector operand mode takes
        w0 = pe["w0"]
        t["xin"] = x
        ex.gemm(x, w0, y0, B, ENC, x.shape[1], x.stride(0), w0.stride(0), ENC, bias=pe["b0"], relu=1)
        t["y0"] = y0
        h, t["m0"], t["r0"] = ex.ln_fwd(y0, pe["g0"], pe["be0"])
        t["h"], t["y"], t["st"] = [h], [], []
        for l, pr in enumerate(pe["res"]):
            y = new(B, ENC)
            ex.linear(h, ENC, pr["w"], pr["b"], y, ENC, B, relu=1, site=SITE_RES + 3 * l + m, p=pc)
            ln, mean, rstd = ex.ln_fwd(y, pr["g"], pr["be"])
            h = ex.add(new(B, ENC), h, ln)                                             # x + LayerNorm(...)      (:73)
            t["y"].append(y); t["st"].append((mean, rstd)); t["h"].append(h)
        ex.linear(h, ENC, pe["wo"], pe["bo"], E[:, m * ENC:(m + 1) * ENC], 3 * ENC, B)
        T["enc"].append(t)
    T["E"] = E
    E3 = E.view(3 * B, ENC)
    # attention: every softmax is over ONE key, so a block is output_proj(drop(value_proj(x))) with one dropout decision per
    # (row, 32-column head) (:141, 172); the two blocks write the halves of one (3B, 512) matrix
    VV = new(3 * B, 2 * ENC)
    S, X = new(3 * B, ENC), new(3 * B, ENC)
    H1, H2 = new(3 * B, ENC // 2), new(3 * B, ENC // 4)
    if Fg is not None:
        # every one of the 3 B (sample, modality) rows is a sample of its own here: value projections of both blocks (one layer,
        # two column ranges), both output projections (one layer, each reading its half); the estimator's two layers (p = 0.2)
        ch = Chain(ex, E3, ENC, ENC, 3 * B, p=pch)
        ch.seg(Fg("wv_s"), ENC, ENC, bias=P["bvs"], site=SITE_ATTN_S, shift=5)
        ch.seg(Fg("wv_c"), ENC, ENC, bias=P["bvc"], site=SITE_ATTN_C, shift=5, nout_off=ENC).end(2 * ENC, stash=VV, ld_stash=2 * ENC)
        ch.seg(Fg("wos"), ENC, ENC, bias=P["bos"])
        ch.seg(Fg("woc"), ENC, ENC, bias=P["boc"], kin=ENC, nout_off=ENC).end(2 * ENC, stash=S, ld_stash=ENC, stash2=X, split=ENC)
        ch.launch()
        ch = Chain(ex, E3, ENC, ENC, 3 * B, p=ex.p_of(0.2))
        ch.seg(Fg("we1"), ENC // 2, ENC, bias=P["be1"], relu=1, site=SITE_EST).end(ENC // 2, stash=H1, ld_stash=ENC // 2)
        ch.seg(Fg("we2"), ENC // 4, ENC // 2, bias=P["be2"], relu=1).end(ENC // 4, stash=H2, ld_stash=ENC // 4)
        ch.launch()
    else:
        ex.linear(E3, ENC, P["wv"][:ENC], P["bv"][:ENC], VV[:, :ENC], 2 * ENC, 3 * B, site=SITE_ATTN_S, p=pc, shift=5)
        ex.linear(E3, ENC, P["wv"][ENC:], P["bv"][ENC:], VV[:, ENC:], 2 * ENC, 3 * B, site=SITE_ATTN_C, p=pc, shift=5)
        ex.linear(VV[:, :ENC], 2 * ENC, P["wos"], P["bos"], S, ENC, 3 * B)
        ex.linear(VV[:, ENC:], 2 * ENC, P["woc"], P["boc"], X, ENC, 3 * B)
        ex.linear(E3, ENC, P["we1"], P["be1"], H1, ENC // 2, 3 * B, relu=1, site=SITE_EST, p=0.2)       # (:186): p = 0.2 always
        ex.linear(H1, ENC // 2, P["we2"], P["be2"], H2, ENC // 4, 3 * B, relu=1)
    pre = new(B, ENC)
    ex.linear(S.view(B, 3 * ENC), 3 * ENC, P["wn1"], P["bn1"], pre, ENC, B)
    AV, Tt = new(B, 2 * ENC), new(B, FUS + ENC)
    r, w4, u4 = new(B, ENC), new(B, 4, d=torch.float32), new(B, 4, d=torch.float32)
    a = _lib.StackBAttnTrainArgs()
    a.h2, a.pre, a.self_out, a.cross_out = H2.data_ptr(), pre.data_ptr(), S.data_ptr(), X.data_ptr()
    a.est_w3, a.est_b3, a.wn_w1_unc, a.wn_w2, a.wn_b2 = (P[k].data_ptr() for k in ("we3", "be3", "wn1u", "wn2", "bn2"))
    a.out_av, a.out_text = AV.data_ptr(), Tt[:, FUS:].data_ptr()
    a.r, a.weights4, a.unc4 = r.data_ptr(), w4.data_ptr(), u4.data_ptr()
    a.ld_w1_unc, a.ld_av, a.ld_text, a.B, a.act_f32 = P["wn1u"].stride(0), 2 * ENC, FUS + ENC, B, f32
    a.training, a.drop_site, a.dropout_p = int(drop is not None), SITE_WN, ex.p_of(pc)
    if drop is not None:
        a.seed, a.offset = drop[1], drop[2]
        if len(drop) > 3 and drop[3] is not None:
            a.offset_dev = drop[3].data_ptr()
    a.stream = ex.s
    if B:
        _lib.check(ex.lib.mmdeer_stackb_attn_mix_train_fwd(C.byref(a)))
    T.update(VV=VV, S=S, X=X, H1=H1, H2=H2, pre=pre, AV=AV, T=Tt, r=r, w4=w4, u4=u4, attn_args=a)
    # fusion
    def stage(inp, ldi, K, pp, site, out, ldo):
        a1 = new(B, FUS)
        ex.gemm(inp, pp["w0"], a1, B, FUS, K, ldi, pp["w0"].stride(0), FUS, bias=pp["b0"], relu=1,
                drop_site=site if ex.p_of(pc) > 0 else -1, p=ex.p_of(pc))
        n1, mean, rstd = ex.ln_fwd(a1, pp["g"], pp["be"])
        ex.linear(n1, FUS, pp["w4"], pp["b4"], out, ldo, B, relu=1)
        return a1, n1, mean, rstd

    def stage_chain(name, inp, ldi, K, pp, site, out, ldo):
        # Linear-ReLU-Dropout-LayerNorm-Linear-ReLU (complete_project.py:315-333) as one launch
        a1, n1, (mean, rstd) = new(B, FUS), new(B, FUS), stat(B)
        ch = Chain(ex, inp, ldi, K, B, p=pch, ts=16 if K > 512 else 0)
        ch.seg(Fg(name + ".w0"), FUS, K, bias=pp["b0"], relu=1, site=site).end(FUS, stash=a1, ld_stash=FUS, ln=(pp["g"], pp["be"], n1, mean, rstd))
        ch.seg(Fg(name + ".w4"), FUS, FUS, bias=pp["b4"], relu=1).end(FUS, stash=out, ld_stash=ldo)
        ch.launch()
        return a1, n1, mean, rstd
    R2 = new(B, FUS)
    if Fg is not None:
        T["av"] = stage_chain("av", AV, 2 * ENC, 2 * ENC, P["av"], SITE_AV, Tt[:, :FUS], FUS + ENC)
        T["tri"] = stage_chain("tri", Tt, FUS + ENC, FUS + ENC, P["tri"], SITE_TRI, R2, FUS)
    else:
        T["av"] = stage(AV, 2 * ENC, 2 * ENC, P["av"], SITE_AV, Tt[:, :FUS], FUS + ENC)
        T["tri"] = stage(Tt, FUS + ENC, FUS + ENC, P["tri"], SITE_TRI, R2, FUS)
    G = new(B, FUS)
    ex.linear(Tt, FUS + ENC, P["wg"], P["bg"], G, FUS, B)
    fused = new(B, FUS)
    fused32 = new(B, FUS, d=torch.float32)
    if B:
        _lib.check(ex.lib.mmdeer_stackb_gate_mix(G.data_ptr(), FUS, R2.data_ptr(), FUS, Tt.data_ptr(), FUS + ENC, fused.data_ptr(), FUS,
                                                 fused32.data_ptr(), B, FUS, f32, ex.s))
    T.update(R2=R2, G=G, fused=fused, fused32=fused32)
    # heads
    H0, H3 = new(B, 3 * HID), new(B, 3 * HID // 2)
    ev = new(B, 12, d=torch.float32)
    if Fg is not None:
        # the first two layers of the three evidence networks (complete_project.py:376-384): 512 -> 3 x 256 -> 3 x 128, one launch
        ch = Chain(ex, fused, FUS, FUS, B, p=pch, ts=16)
        for d in range(3):
            ch.seg(Fg(f"wh0.{d}"), HID, FUS, bias=P["bh0s"][d], relu=1, site=SITE_H0, dcol=d * HID, nout_off=d * HID)
        ch.end(3 * HID, stash=H0, ld_stash=3 * HID)
        for d in range(3):
            ch.seg(Fg(f"wh3.{d}"), HID // 2, HID, bias=P["bh3"][d], relu=1, site=SITE_H3 + d, kin=d * HID, nout_off=d * (HID // 2))
        ch.end(3 * HID // 2, stash=H3, ld_stash=3 * HID // 2)
        ch.launch()
    else:
        ex.linear(fused, FUS, P["wh0"], P["bh0"], H0, 3 * HID, B, relu=1, site=SITE_H0, p=pc)
    for d in range(3):
        if Fg is None:
            ex.linear(H0[:, d * HID:(d + 1) * HID], 3 * HID, P["wh3"][d], P["bh3"][d], H3[:, d * 128:(d + 1) * 128], 3 * HID // 2, B, relu=1,
                      site=SITE_H3 + d, p=pc)
        ex.linear(H3[:, d * 128:(d + 1) * 128], 3 * HID // 2, P["wh6"][d], P["bh6"][d], ev[:, 4 * d:4 * d + 4], 12, B)
    planes = new(8, B, 3, d=torch.float32)
    cal = model.calibration_layer
    cn = cal.calibration_network
    cps = [t.detach().float().reshape(-1).contiguous() for t in (cal.temperature, cn[0].weight, cn[0].bias, cn[2].weight, cn[2].bias, cn[4].weight, cn[4].bias)]
    if B:
        _lib.check(ex.lib.mmdeer_stackb_head(ev.data_ptr(), 12, *[t.data_ptr() for t in cps], planes.data_ptr(), B, ex.s))
    T.update(H0=H0, H3=H3, ev=ev, planes=planes, cal_keep=cps)
    return T


def backward(model, T: Dict, g4: torch.Tensor, flat=None) -> Dict[str, torch.Tensor]:
    """Gradients of every parameter on the path from d (mu, nu, alpha, beta) = g4 (4, B, 3) fp32.  Returns
    {state_dict key: fp32 gradient}; the calibration layer (not on the path of mu / nu / alpha / beta) gets none.

    With ``flat`` (the fused training step) every gradient is written straight into its slice of the model's flat gradient
    buffer (``.grad`` of each parameter is a view of it), the weight-gradient GEMMs are recorded and run as grouped launches
    with one slab fold per group at the end (``Exec.flush_dw`` -> ``mmdeer_gemm_batch``), and nothing is returned through
    autograd; the query / key projections keep the zeros the buffer was created with."""
    ex: Exec = T["ex"]
    P, B, dt, f32 = T["P"], T["B"], ex.dt, ex.f32
    cfg = model.config
    pc = cfg.dropout
    dev = g4.device
    new = lambda *s, d=None: torch.empty(*s, dtype=d or dt, device=dev)
    # every fp32 gradient / scratch matrix of the pass is a slice of ONE zeroed buffer (one memset instead of ~100)
    pool = torch.zeros((0 if flat is not None else sum(p.numel() for p in model.parameters())) + (1 << 20), dtype=torch.float32, device=dev)
    cursor = [0]

    def z32(*shape):
        n = 1
        for d in shape:
            n *= d
        lo = cursor[0]
        cursor[0] = (lo + n + 63) // 64 * 64                       # 256-byte aligned slices
        if cursor[0] > pool.numel():
            return torch.zeros(*shape, dtype=torch.float32, device=dev)
        return pool[lo:lo + n].view(*shape)

    G: Dict[str, torch.Tensor] = {}
    sc = ex.scale_of(pc)
    Fg = flat.get("frag") if (flat is not None and T.get("chain")) else None      # the layer-chain plan (see forward_train)
    late = []                      # flat mode: (destination view, source) copies that must wait for the grouped dW launches
    PT = P.get("t")                # transposed weight copies (flat mode) or None
    if flat is not None:
        ex.deferred, ex.folds = [], []
        fview = lambda name: flat["gview"][name]

    def grads(prefix, N, K):
        if flat is not None:
            return fview(prefix + ".weight"), fview(prefix + ".bias")
        G[prefix + ".weight"], G[prefix + ".bias"] = z32(N, K), z32(N)
        return G[prefix + ".weight"], G[prefix + ".bias"]

    def vec(name, n):              # a vector gradient the row kernels write directly (LayerNorm gamma / beta)
        if flat is not None:
            return fview(name)
        G[name] = z32(n)
        return G[name]

    def put(name, src, pad_from=None):            # a gradient that is a slice of a wider scratch matrix
        # pad_from (flat mode): `src` is the head of that zero-tailed scratch vector (the rows behind it are exact zeros: the products of the
        # zero columns the row kernels pad their 8-column outputs with) and has fewer than 4 elements -- four are copied, so that the copy
        # rides in the one batched fold launch; the extra zeros land in the parameter's alignment gap of the flat buffer (64-element slots)
        if flat is not None:
            n = src.numel()
            if pad_from is not None and n % 4:
                off = flat["offs"][name]
                late.append((flat["g"][off:off + (n + 3) // 4 * 4], pad_from.reshape(-1)[:(n + 3) // 4 * 4]))
            else:
                late.append((fview(name), src))
        else:
            G[name] = src

    # ---- heads
    dev_ = new(B, 24)                                                   # head d in columns 8 d .. 8 d + 3, zeros in 8 d + 4 .. 8 d + 7
    _lib.check(ex.lib.mmdeer_stackb_head_bwd(T["ev"].data_ptr(), 12, g4.contiguous().data_ptr(), dev_.data_ptr(), 24, B, f32, ex.s))
    H0, H3 = T["H0"], T["H3"]
    dH3, dH0 = new(B, 3 * HID // 2), new(B, 3 * HID)
    names = ("valence", "arousal", "dominance")
    for d, nm in enumerate(names):
        pre = f"prediction_heads.{nm}.evidence_network"
        h3, dh3, ev_d = H3[:, d * 128:(d + 1) * 128], dH3[:, d * 128:(d + 1) * 128], dev_[:, 8 * d:8 * d + 8]
        if Fg is None:
            ex.dx(ev_d, 24, P["wh6p"][d], dh3, 3 * HID // 2, B, mask=h3, ldm=3 * HID // 2, mask_scale=sc)
        elif d == 0:        # chain plan: one launch for the three heads (the block-diagonal image of their last layers)
            ex.dx(dev_, 24, Fg.mat("wh6bd"), dH3, 3 * HID // 2, B, mask=H3, ldm=3 * HID // 2, mask_scale=sc)
        w8, b8 = z32(8, 128), z32(8)
        ex.dw(ev_d, 24, h3, 3 * HID // 2, w8, b8, B, 8, 128)
        put(pre + ".6.weight", w8[:4]); put(pre + ".6.bias", b8[:4])
        h0, dh0 = H0[:, d * HID:(d + 1) * HID], dH0[:, d * HID:(d + 1) * HID]
        if Fg is None:
            ex.dx(dh3, 3 * HID // 2, P["wh3"][d], dh0, 3 * HID, B, mask=h0, ldm=3 * HID, mask_scale=sc, wt=PT and PT["wh3"][d])
        ex.dw(dh3, 3 * HID // 2, h0, 3 * HID, *grads(pre + ".3", 128, HID), B, 128, HID)
    dfused = new(B, FUS)
    if Fg is not None:         # d H0 = d H3 W3 (masked by H0) per head, d fused = d H0 W0 (the stacked first layers): one launch
        ch = Chain(ex, dH3, 3 * HID // 2, 3 * HID // 2, B, ts=16)
        for d in range(3):
            ch.seg(Fg(f"wh3.{d}.T"), HID, HID // 2, kin=d * (HID // 2), nout_off=d * HID, mask=H0, ldm=3 * HID, mcol=d * HID, mscale=sc)
        ch.end(3 * HID, stash=dH0, ld_stash=3 * HID)
        ch.seg(Fg("wh0.T"), FUS, 3 * HID).end(FUS, stash=dfused, ld_stash=FUS)
        ch.launch()
    else:
        ex.dx(dH0, 3 * HID, P["wh0"], dfused, FUS, B, wt=PT and PT["wh0"])
    gw0, gb0 = z32(3 * HID, FUS), z32(3 * HID)
    ex.dw(dH0, 3 * HID, T["fused"], FUS, gw0, gb0, B, 3 * HID, FUS)
    for d, nm in enumerate(names):
        put(f"prediction_heads.{nm}.evidence_network.0.weight", gw0[d * HID:(d + 1) * HID])
        put(f"prediction_heads.{nm}.evidence_network.0.bias", gb0[d * HID:(d + 1) * HID])
    # ---- fusion: gate mix, trimodal stage, gate, audio-visual stage
    Tt, R2, Gt = T["T"], T["R2"], T["G"]
    dG, dZ4, dav_a = new(B, FUS), new(B, FUS), new(B, FUS)
    _lib.check(ex.lib.mmdeer_stackb_gate_mix_bwd(dfused.data_ptr(), FUS, Gt.data_ptr(), FUS, R2.data_ptr(), FUS, Tt.data_ptr(), FUS + ENC,
                                                 dG.data_ptr(), FUS, dZ4.data_ptr(), FUS, dav_a.data_ptr(), FUS, B, FUS, f32, ex.s))

    def stage_bwd(name, dz4, st, inp, ldi, K):
        a1, n1, mean, rstd = st
        pp = P[name]
        pre = f"fusion_module.{'av_fusion' if name == 'av' else 'trimodal_fusion'}"
        dn1 = ex.dx(dz4, FUS, pp["w4"], new(B, FUS), FUS, B, wt=PT and PT[name]["w4"])
        ex.dw(dz4, FUS, n1, FUS, *grads(pre + ".4", FUS, FUS), B, FUS, FUS)
        dz0 = ex.ln_bwd(dn1, a1, mean, rstd, pp["g"], vec(pre + ".3.weight", FUS), vec(pre + ".3.bias", FUS), sc)
        ex.dw(dz0, FUS, inp, ldi, *grads(pre + ".0", FUS, K), B, FUS, K)
        return dz0

    def stage_bwd_chain(name, dz4, st, inp, ldi, K):
        # d n1 = d z4 W4, LayerNorm backward (masked) -> d z0, d input = d z0 W0: one launch (the weight gradients read dz4 / dz0)
        a1, n1, mean, rstd = st
        pp = P[name]
        pre = f"fusion_module.{'av_fusion' if name == 'av' else 'trimodal_fusion'}"
        ch = Chain(ex, dz4, FUS, FUS, B, ts=16 if K > 512 else 0)
        nwg = ch.workgroups()
        dz0, din = new(B, FUS), new(B, K)
        part = torch.empty(nwg * 2 * FUS, dtype=torch.float32, device=dev)
        ch.seg(Fg(name + ".w4.T"), FUS, FUS).end(FUS, lnb=(pp["g"], a1, mean, rstd, dz0, part, sc))
        for c0 in range(0, K, 384 if K > 512 else K):          # W0^T is [K][512]: at most four 128-column tiles per segment
            n = min(384 if K > 512 else K, K - c0)
            ch.seg(Fg(name + ".w0.T", c0), n, FUS, nout_off=c0)
        ch.end(K, stash=din, ld_stash=K)
        ch.launch()
        ex.folds.append((part, vec(pre + ".3.weight", FUS), nwg, FUS, 2 * FUS))
        ex.folds.append((part[FUS:], vec(pre + ".3.bias", FUS), nwg, FUS, 2 * FUS))
        ex.dw(dz4, FUS, n1, FUS, *grads(pre + ".4", FUS, FUS), B, FUS, FUS)
        ex.dw(dz0, FUS, inp, ldi, *grads(pre + ".0", FUS, K), B, FUS, K)
        return din

    if Fg is not None:
        dT = stage_bwd_chain("tri", dZ4, T["tri"], Tt, FUS + ENC, FUS + ENC)
    else:
        dz0_tri = stage_bwd("tri", dZ4, T["tri"], Tt, FUS + ENC, FUS + ENC)
        dT = ex.dx(dz0_tri, FUS, P["tri"]["w0"], new(B, FUS + ENC), FUS + ENC, B, wt=PT and PT["tri"]["w0"])
    dT2 = ex.dx(dG, FUS, P["wg"], new(B, FUS + ENC), FUS + ENC, B, wt=PT and PT["wg"])
    ex.dw(dG, FUS, Tt, FUS + ENC, *grads("fusion_module.fusion_gate.0", FUS, FUS + ENC), B, FUS, FUS + ENC)
    dtext = ex.add(new(B, ENC), dT[:, FUS:], dT2[:, FUS:])
    # d av_fused: through the trimodal input, the gate input and the gate mix; av_fused = relu(.)
    tmp = ex.add(new(B, FUS), dT[:, :FUS], dT2[:, :FUS])
    dz4_av = ex.add(new(B, FUS), tmp, dav_a, mask=Tt[:, :FUS], scale=1.0)
    if Fg is not None:
        dAV = stage_bwd_chain("av", dz4_av, T["av"], T["AV"], 2 * ENC, 2 * ENC)
    else:
        dz0_av = stage_bwd("av", dz4_av, T["av"], T["AV"], 2 * ENC, 2 * ENC)
        dAV = ex.dx(dz0_av, FUS, P["av"]["w0"], new(B, 2 * ENC), 2 * ENC, B, wt=PT and PT["av"]["w0"])
    # ---- attention tail
    a = T["attn_args"]
    dS, dpre = new(3 * B, ENC), new(B, ENC)
    dSX = new(3 * B, 2 * ENC) if Fg is not None else None        # chain plan: [d self | d cross], the input rows of the attention dX run
    dX = dSX[:, ENC:] if Fg is not None else new(3 * B, ENC)
    a.ld_dcross = 2 * ENC if Fg is not None else 0
    unc8 = new(B, 8) if (flat is not None and not f32 and B >= 2) else None     # the uncertainties as a bf16 GEMM operand (a by-product of the kernel below; both plans of the bf16 fused step)
    a.unc8 = _ptr(unc8)
    dlog8, dz8e, dh2 = new(B, 8), new(3 * B, 8), new(3 * B, ENC // 4)
    a.d_av, a.d_text = dAV.data_ptr(), dtext.data_ptr()
    a.ld_text = ENC                                  # the text gradient is a dense (B, 256) matrix here
    a.d_self, a.d_cross, a.d_pre = dS.data_ptr(), dX.data_ptr(), dpre.data_ptr()
    a.d_logits8, a.d_z8, a.d_h2 = dlog8.data_ptr(), dz8e.data_ptr(), dh2.data_ptr()
    if B:
        _lib.check(ex.lib.mmdeer_stackb_attn_mix_bwd(C.byref(a)))
    att = "attention_module"
    t4w, t4b = z32(8, ENC), z32(8)
    ex.dw(dlog8, 8, T["r"], ENC, t4w, t4b, B, 8, ENC)                                      # weight_network.3 (3 x 256)
    put(att + ".weight_network.3.weight", t4w[:3]); put(att + ".weight_network.3.bias", t4b[:3], pad_from=t4b)
    if flat is not None:
        gwn, gbn = fview(att + ".weight_network.0.weight"), fview(att + ".weight_network.0.bias")
    else:
        gwn = z32(ENC, 3 * ENC + 3)
        G[att + ".weight_network.0.weight"], G[att + ".weight_network.0.bias"] = gwn, z32(ENC)
        gbn = G[att + ".weight_network.0.bias"]
    feat = z32(ENC, 3 * ENC)
    ex.dw(dpre, ENC, T["S"].view(B, 3 * ENC), 3 * ENC, feat, gbn, B, ENC, 3 * ENC)
    unc = z32(ENC, 8 if unc8 is not None else 4)
    if unc8 is not None:
        ex.dw(dpre, ENC, unc8, 8, unc, None, B, ENC, 8)
    elif B >= 2:
        ex.dw(dpre, ENC, T["u4"], 4, unc, None, B, ENC, 4)
    else:           # a (1, 4) operand is below the GEMM's 8-element minimum: the same product over two rows, the second zero
        d2, u2 = torch.zeros(2, ENC, dtype=dt, device=dev), z32(2, 4)
        d2[:1].copy_(dpre); u2[:1].copy_(T["u4"])
        ex.dw(d2, ENC, u2, 4, unc, None, 2, ENC, 4)
    if flat is not None:
        late.append((gwn[:, :3 * ENC], feat)); late.append((gwn[:, 3 * ENC:], unc[:, :3]))
    else:
        gwn[:, :3 * ENC].copy_(feat); gwn[:, 3 * ENC:].copy_(unc[:, :3])                   # (memory plumbing: two column blocks of one parameter)
    dS_b = ex.dx(dpre, ENC, P["wn1"], new(B, 3 * ENC), 3 * ENC, B, wt=PT and PT["wn1"])
    dS = ex.add(dSX[:, :ENC] if Fg is not None else new(3 * B, ENC), dS, dS_b.view(3 * B, ENC))
    # uncertainty estimator
    est = att + ".uncertainty_estimator.estimator"
    t4w2, t4b2 = z32(8, ENC // 4), z32(8)
    ex.dw(dz8e, 8, T["H2"], ENC // 4, t4w2, t4b2, 3 * B, 8, ENC // 4)
    put(est + ".5.weight", t4w2[:1]); put(est + ".5.bias", t4b2[:1], pad_from=t4b2)
    E3 = T["E"].view(3 * B, ENC)
    if Fg is not None:         # the estimator's two dX products on the 3 B rows: one launch
        dH1, dE_est = new(3 * B, ENC // 2), new(3 * B, ENC)
        ch = Chain(ex, dh2, ENC // 4, ENC // 4, 3 * B)
        ch.seg(Fg("we2.T"), ENC // 2, ENC // 4, mask=T["H1"], ldm=ENC // 2, mscale=ex.scale_of(0.2)).end(ENC // 2, stash=dH1, ld_stash=ENC // 2)
        ch.seg(Fg("we1.T"), ENC, ENC // 2).end(ENC, stash=dE_est, ld_stash=ENC)
        ch.launch()
    else:
        dH1 = ex.dx(dh2, ENC // 4, P["we2"], new(3 * B, ENC // 2), ENC // 2, 3 * B, mask=T["H1"], ldm=ENC // 2, mask_scale=ex.scale_of(0.2), wt=PT and PT["we2"])
        dE_est = ex.dx(dH1, ENC // 2, P["we1"], new(3 * B, ENC), ENC, 3 * B, wt=PT and PT["we1"])
    ex.dw(dh2, ENC // 4, T["H1"], ENC // 2, *grads(est + ".3", ENC // 4, ENC // 2), 3 * B, ENC // 4, ENC // 2)
    ex.dw(dH1, ENC // 2, E3, ENC, *grads(est + ".0", ENC // 2, ENC), 3 * B, ENC // 2, ENC)
    # the two attention blocks: output_proj, then the value projections (attention-dropout factor regenerated)
    VV = T["VV"]
    dVV = new(3 * B, 2 * ENC)
    if Fg is not None:
        # both output-projection dX products (each on its half of the [d self | d cross] rows, the attention-dropout factor of the
        # forward regenerated per (row, head)), then the stacked value projections' dX: one launch
        dE_v = new(3 * B, ENC)
        ch = Chain(ex, dSX, 2 * ENC, 2 * ENC, 3 * B, p=ex.p_of(pc))
        ch.seg(Fg("wos.T"), ENC, ENC, site=SITE_ATTN_S, shift=5)
        ch.seg(Fg("woc.T"), ENC, ENC, site=SITE_ATTN_C, shift=5, kin=ENC, nout_off=ENC).end(2 * ENC, stash=dVV, ld_stash=2 * ENC)
        ch.seg(Fg("wv.T"), ENC, 2 * ENC).end(ENC, stash=dE_v, ld_stash=ENC)
        ch.launch()
    else:
        ex.dx(dS, ENC, P["wos"], dVV[:, :ENC], 2 * ENC, 3 * B, regen_site=SITE_ATTN_S, shift=5, p=pc, wt=PT and PT["wos"])
        ex.dx(dX, ENC, P["woc"], dVV[:, ENC:], 2 * ENC, 3 * B, regen_site=SITE_ATTN_C, shift=5, p=pc, wt=PT and PT["woc"])
    ex.dw(dS, 2 * ENC if Fg is not None else ENC, VV[:, :ENC], 2 * ENC, *grads(att + ".self_attention.output_proj", ENC, ENC), 3 * B, ENC, ENC)
    ex.dw(dX, 2 * ENC if Fg is not None else ENC, VV[:, ENC:], 2 * ENC, *grads(att + ".cross_attention.output_proj", ENC, ENC), 3 * B, ENC, ENC)
    gwv, gbv = z32(2 * ENC, ENC), z32(2 * ENC)
    ex.dw(dVV, 2 * ENC, E3, ENC, gwv, gbv, 3 * B, 2 * ENC, ENC)
    for i, blk in enumerate(("self_attention", "cross_attention")):
        put(f"{att}.{blk}.value_proj.weight", gwv[i * ENC:(i + 1) * ENC]); put(f"{att}.{blk}.value_proj.bias", gbv[i * ENC:(i + 1) * ENC])
        # one key per query: the softmax is the constant 1, so query / key projections get exact zeros (as autograd gives;
        # in flat mode their slices of the gradient buffer are never written and keep the zeros it was created with)
        if flat is None:
            for q in ("query_proj", "key_proj"):
                G[f"{att}.{blk}.{q}.weight"], G[f"{att}.{blk}.{q}.bias"] = z32(ENC, ENC), z32(ENC)
    if Fg is None:
        dE_v = ex.dx(dVV, 2 * ENC, P["wv"], new(3 * B, ENC), ENC, 3 * B, wt=PT and PT["wv"])
    dE = ex.add(new(3 * B, ENC), dE_v, dE_est).view(B, 3 * ENC)
    # ---- encoders
    for m, (t, pe, ename) in enumerate(zip(T["enc"], P["enc"], ("audio_encoder", "video_encoder", "text_encoder"))):
        if Fg is None:
            break
        # one launch per encoder: dh = dE W_out, then per residual block (top down) dz = LayerNorm'(dh) masked, dh += dz W -- the
        # bypass gradient rides in columns [256, 512) of the panel while the LayerNorm backward rewrites [0, 256) -- and the stem's
        # LayerNorm backward at the end; the weight gradients read the stored dz / h rows in the grouped launches
        dEm = dE[:, m * ENC:(m + 1) * ENC]
        hs, L = t["h"], len(pe["res"])
        ch = Chain(ex, dEm, 3 * ENC, ENC, B)
        nwg = ch.workgroups()
        ex.dw(dEm, 3 * ENC, hs[-1], ENC, *grads(ename + ".output_projection", ENC, ENC), B, ENC, ENC)
        ch.seg(Fg(f"enc{m}.wo.T"), ENC, ENC, res_dup=int(L > 0))

        def lnb_end(y, mean, rstd, gamma, gname, bname, ms):
            dz = new(B, ENC)
            part = torch.empty(nwg * 2 * ENC, dtype=torch.float32, device=dev)
            ch.end(ENC, lnb=(gamma, y, mean, rstd, dz, part, ms))
            ex.folds.append((part, vec(gname, ENC), nwg, ENC, 2 * ENC))
            ex.folds.append((part[ENC:], vec(bname, ENC), nwg, ENC, 2 * ENC))
            return dz

        for l in reversed(range(L)):
            pr, pre = pe["res"][l], f"{ename}.encoder_layers.{l}.layers"
            dz = lnb_end(t["y"][l], *t["st"][l], pr["g"], pre + ".3.weight", pre + ".3.bias", sc)
            ex.dw(dz, ENC, hs[l], ENC, *grads(pre + ".0", ENC, ENC), B, ENC, ENC)
            ch.seg(Fg(f"enc{m}.res{l}.T"), ENC, ENC, res_add=1, res_dup=int(l > 0))
        pre = ename + ".input_projection"
        dz0 = lnb_end(t["y0"], t["m0"], t["r0"], pe["g0"], pre + ".2.weight", pre + ".2.bias", 1.0)
        ch.launch()
        K = t["xin"].shape[1]
        ex.dw(dz0, ENC, t["xin"], t["xin"].stride(0), *grads(pre + ".0", ENC, K), B, ENC, K)
    for m, (t, pe, ename) in enumerate(zip(T["enc"], P["enc"], ("audio_encoder", "video_encoder", "text_encoder"))):
        if Fg is not None:
            break
        dEm = dE[:, m * ENC:(m + 1) * ENC]
        hs = t["h"]
        dh = ex.dx(dEm, 3 * ENC, pe["wo"], new(B, ENC), ENC, B, wt=PT and PT["enc"][m]["wo"])
        ex.dw(dEm, 3 * ENC, hs[-1], ENC, *grads(ename + ".output_projection", ENC, ENC), B, ENC, ENC)
        for l in reversed(range(len(pe["res"]))):
            pr, pre = pe["res"][l], f"{ename}.encoder_layers.{l}.layers"
            dz = ex.ln_bwd(dh, t["y"][l], *t["st"][l], pr["g"], vec(pre + ".3.weight", ENC), vec(pre + ".3.bias", ENC), sc)
            ex.dw(dz, ENC, hs[l], ENC, *grads(pre + ".0", ENC, ENC), B, ENC, ENC)
            dh = ex.add(new(B, ENC), dh, ex.dx(dz, ENC, pr["w"], new(B, ENC), ENC, B, wt=PT and PT["enc"][m]["res"][l]))
        pre = ename + ".input_projection"
        dz0 = ex.ln_bwd(dh, t["y0"], t["m0"], t["r0"], pe["g0"], vec(pre + ".2.weight", ENC), vec(pre + ".2.bias", ENC), 1.0)
        K = t["xin"].shape[1]
        ex.dw(dz0, ENC, t["xin"], t["xin"].stride(0), *grads(pre + ".0", ENC, K), B, ENC, K)
    if flat is not None:
        ex.flush_dw()
        ex.deferred = None
        for dst, src in late:      # after the weight-gradient launches; dense runs ride in the one fold launch below
            ex.copy1d(dst, src)
        ex.flush_folds()
        ex.folds = None
    return G
