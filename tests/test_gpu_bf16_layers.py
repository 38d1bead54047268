"""Every launch of the BENCH CONFIGURATION (B = 4096, bf16, dropout 0.3, fused plan) checked on its own stored inputs.

A bf16 chain is chaotic at the rounding level (tests/test_gpu_bf16_parity.py), so an end-to-end tolerance cannot be tighter
than a few per cent on the gradients.  Here the chain is cut at every launch boundary instead ("teacher forcing"): the
activations and activation-gradients one HIP training step left in its workspace are read back through
mmdeer_workspace_offset, and each kernel's output is compared with the oracle's restatement of THAT kernel
(oracle.k_linear / k_dx / k_ln_fwd / k_ln_bwd / k_tri_fwd / k_tri_bwd / k_nig_bwd / k_dw) applied to the kernel's own inputs.
What remains is one kernel's fp32 summation order:

 * bf16 tensors (35 stored activations / activation-gradients): every element within ONE bf16 ulp of the oracle's, and
   at most FLIP_FRAC of them different at all (an fp32 sum that lands on the other side of a rounding boundary) --
   measured at B = 4096: 0 to 2.6e-4 of the elements, never more than one ulp (the test prints the table);
 * fp32 tensors (evidence, LayerNorm statistics, every weight / bias / LayerNorm gradient of the step -- i.e. the split-K
   weight-gradient launch and its slab fold at full size): <= 1e-5 of the tensor's largest element against an fp64 sum
   (measured <= 4.6e-7);
 * the one internal rounding (q, k to bf16 inside the fused projection + attention kernel) is handled where it occurs.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmdeer import _lib, synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from oracle import deer_oracle as O  # noqa: E402

from .test_gpu_model import dump_masks  # noqa: E402

DEV = "cuda:0"
FLIP_FRAC = 1.5e-3     # measured worst 2.9e-4


def ordered(x):
    """bf16 values (as an fp32 tensor) -> integers in which neighbouring bf16 values differ by 1 (+0 and -0 coincide)."""
    bits = x.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    mag = bits & 0x7FFF
    return torch.where(bits >= 0x8000, -mag, mag)


class Report:
    def __init__(self):
        self.rows = []

    def bf16(self, name, got, ref):
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert torch.isfinite(got).all(), name
        d = (ordered(got) - ordered(ref)).abs()
        frac = float((d > 0).float().mean())
        # an fp32 sum carries an ABSOLUTE error of ~1e-6 of the tensor's scale: an element that small itself (cancellation)
        # may sit many of ITS ulps away, so differences below 1e-5 of the largest element are not counted in ulps
        tiny = (got - ref).abs() <= 1e-5 * float(ref.abs().max())
        worst = int(d[~tiny].max()) if (~tiny).any() else 0
        self.rows.append(f"  {name:34s} bf16 {tuple(got.shape)!s:14s} differing {frac:.2e}   max ulp distance {worst}")
        assert worst <= 1, (name, worst)
        assert frac <= FLIP_FRAC, (name, frac)

    def f32(self, name, got, ref, tol=1e-5):
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = max(float(ref.abs().max()), 1e-30)
        err = float((got.double() - ref.double()).abs().max()) / scale
        self.rows.append(f"  {name:34s} fp32 {tuple(got.shape)!s:14s} max error / max element {err:.2e}")
        assert err <= tol, (name, err)


@pytest.mark.parametrize("B,dropout", [(4096, 0.3), (515, 0.3), (256, 0.0), (8192, 0.3)])   # 8192: the 32-sample chain workgroups
def test_every_launch_of_a_bf16_step_on_its_own_inputs(B, dropout):
    p = dropout
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(B, seed=42).items()}
    m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=p, seed=43)).to(DEV).train()
    a_d, v_d, t_d = (b[k].to(DEV).bfloat16() for k in ("audio", "video", "text"))
    ld = m.train_step(a_d, v_d, t_d, b["targets"].to(DEV))
    torch.cuda.synchronize()
    masks = dump_masks(m, B, m._step) if p > 0 else {}
    lib = _lib.load()
    ws = m._workspace(B, torch.device(DEV))

    def buf(name, shape, dtype=torch.bfloat16):
        off = lib.mmdeer_workspace_offset(B, 0, name.encode())
        assert off >= 0, name
        n = int(np.prod(shape)) * (2 if dtype == torch.bfloat16 else 4)
        return ws[off:off + n].view(dtype).view(*shape).float().cpu()

    assert lib.mmdeer_workspace_offset(B, 0, b"no_such_buffer") == -1
    P = {k: w.detach().cpu() for k, w in m.state_dict().items()}
    G = {k: w.grad.detach().cpu() for k, w in m.named_parameters() if w.grad is not None}
    audio, video, text = a_d.float().cpu(), v_d.float().cpu(), t_d.float().cpu()
    R = Report()
    av, tf, hd = "fusion.audio_visual_fusion.", "fusion.trimodal_fusion.", "head."
    Wb = lambda pre: (P[pre + ".weight"], P[pre + ".bias"])
    mk = masks.get

    # ------------------------------------------------------------------ forward
    avin = buf("avin", (2 * B, 256))
    R.bf16("F1 video_projection", avin[:B], O.k_linear(video, *Wb(av + "video_projection")))
    R.bf16("F1 audio_projection", avin[B:], O.k_linear(audio, *Wb(av + "audio_projection")))
    xtok = buf("xtok", (B, 2, 512))
    R.bf16("F1 text_projection", xtok[:, 1], O.k_linear(text, *Wb(tf + "text_projection")))
    w_in, b_in = P[av + "cross_attention.in_proj_weight"], P[av + "cross_attention.in_proj_bias"]
    avv = buf("avv", (2 * B, 256))
    head_mask = None
    if p > 0:
        head_mask = torch.cat([mk("av_attn_a2v"), mk("av_attn_v2a")], dim=0).repeat_interleave(32, dim=1)   # one decision per (row, head)
    R.bf16("F2 AV value projection", avv, O.k_linear(avin, w_in[512:], b_in[512:], mask=head_mask, p=p))
    cat = buf("cat", (B, 512))
    R.bf16("F3 AV out_proj (audio_att)", cat[:, :256], O.k_linear(avv[:B], *Wb(av + "cross_attention.out_proj")))
    R.bf16("F3 AV out_proj (video_att)", cat[:, 256:], O.k_linear(avv[B:], *Wb(av + "cross_attention.out_proj")))
    y_a2 = buf("y_a2", (B, 256))
    R.bf16("F4 fusion_layers.0", y_a2, O.k_linear(cat, *Wb(av + "fusion_layers.0"), relu=True, mask=mk("av_fuse"), p=p))
    avf = buf("av", (B, 256))
    ref, mu, rs = O.k_ln_fwd(y_a2, P[av + "fusion_layers.3.weight"], P[av + "fusion_layers.3.bias"])
    R.bf16("F5 LayerNorm 256", avf, ref)
    mean_a2, rstd_a2 = buf("mean_a2", (B,), torch.float32), buf("rstd_a2", (B,), torch.float32)
    R.f32("F5 mean", mean_a2, mu, 1e-5); R.f32("F5 rstd", rstd_a2, rs, 1e-5)
    R.bf16("F6 audiovisual_projection", xtok[:, 0], O.k_linear(avf, *Wb(tf + "audiovisual_projection")))
    w_t, b_t = P[tf + "modality_attention.in_proj_weight"], P[tf + "modality_attention.in_proj_bias"]
    probs = buf("probs", (B, 8, 2, 2), torch.float32)
    obar = buf("obar", (B, 512))
    # the fused kernel rounds q, k to bf16 INSIDE (for the score MFMAs): an element of q / k that rounds the other way moves
    # a probability by up to ~5e-4 (measured 5.4e-4 at B = 4096: ~1e-4 of the 2 M elements flip), so the probabilities get
    # that band plus a bound on how many differ at all, and the context is checked from the kernel's own probabilities
    prob_ref, _ = O.k_tri_fwd(xtok, w_t, b_t, mask=mk("tri_attn"), p=p)
    R.f32("F7-8 softmax probabilities", probs, prob_ref, 2e-3)
    moved = float(((probs - prob_ref).abs() > 2e-5).float().mean())
    R.rows.append(f"  {'F7-8 probabilities off by > 2e-5':34s} fraction {moved:.2e}")
    assert moved < 0.1, moved
    assert float((probs.sum(-1) - 1).abs().max()) < 1e-6
    R.bf16("F7-8 fused in_proj + attention", obar, O.k_tri_fwd(xtok, w_t, b_t, mask=mk("tri_attn"), p=p, probs=probs)[1])
    pool = buf("pool", (B, 512))
    R.bf16("F9 attention out_proj", pool, O.k_linear(obar, *Wb(tf + "modality_attention.out_proj")))
    y_t3 = buf("y_t3", (B, 512))
    R.bf16("F10 final_fusion.0", y_t3, O.k_linear(pool, *Wb(tf + "final_fusion.0"), relu=True, mask=mk("tri_fuse"), p=p))
    tri = buf("tri", (B, 512))
    R.bf16("F11 LayerNorm 512", tri, O.k_ln_fwd(y_t3, P[tf + "final_fusion.3.weight"], P[tf + "final_fusion.3.bias"])[0])
    y_o1 = buf("y_o1", (B, 512))
    R.bf16("F12 output_projection.0", y_o1, O.k_linear(tri, *Wb("fusion.output_projection.0"), relu=True, mask=mk("out_proj"), p=p))
    fused = buf("fused", (B, 512))
    R.bf16("F13 LayerNorm 512", fused, O.k_ln_fwd(y_o1, P["fusion.output_projection.3.weight"], P["fusion.output_projection.3.bias"])[0])
    h1 = buf("h1", (B, 256))
    R.bf16("F14 feature_processor.0", h1, O.k_linear(fused, *Wb(hd + "feature_processor.0"), relu=True, mask=mk("fp0"), p=p))
    h2 = buf("h2", (B, 256))
    R.bf16("F15 feature_processor.3", h2, O.k_linear(h1, *Wb(hd + "feature_processor.3"), relu=True, mask=mk("fp1"), p=p))
    e1, e2 = buf("e1", (B, 3, 128)), buf("e2", (B, 3, 64))
    evid = buf("evid", (B, 3, 4), torch.float32)
    ev = lambda i, l: hd + f"deer_heads.{i}.evidence_net.{l}"
    for i in range(3):
        m0 = None if p == 0 else mk("ev0")[:, i]
        m1 = None if p == 0 else mk("ev1")[:, i]
        R.bf16(f"F16 evidence_net.0 head {i}", e1[:, i], O.k_linear(h2, *Wb(ev(i, 0)), relu=True, mask=m0, p=p))
        R.bf16(f"F17 evidence_net.3 head {i}", e2[:, i], O.k_linear(e1[:, i], *Wb(ev(i, 3)), relu=True, mask=m1, p=p))
        R.f32(f"F18 evidence head {i}", evid[:, i], e2[:, i] @ O._bf16_round(P[ev(i, 6) + ".weight"]).t() + P[ev(i, 6) + ".bias"], 2e-6)

    # ------------------------------------------------------------------ backward: activation-gradient chain
    dz2 = buf("dz2", (B, 3, 64))
    dev_ref, dz2_ref = O.k_nig_bwd(evid, b["targets"], e2.reshape(B, 192), [P[ev(i, 6) + ".weight"] for i in range(3)], p=p)
    R.bf16("B1 nig_bwd dz2", dz2.reshape(B, 192), dz2_ref)
    de1 = buf("de1", (B, 3, 128))
    for i in range(3):
        R.bf16(f"B2 dX evidence_net.3 head {i}", de1[:, i], O.k_dx(dz2[:, i], P[ev(i, 3) + ".weight"], ymask=e1[:, i], p=p))
    dh2 = buf("dh2", (B, 256))
    w0 = torch.cat([P[ev(i, 0) + ".weight"] for i in range(3)], dim=0)                    # stacked (384, 256)
    R.bf16("B3 dX evidence_net.0 (stacked)", dh2, O.k_dx(de1.reshape(B, 384), w0, ymask=h2, p=p))
    dh1 = buf("dh1", (B, 256))
    R.bf16("B4 dX feature_processor.3", dh1, O.k_dx(dh2, P[hd + "feature_processor.3.weight"], ymask=h1, p=p))
    dfused = buf("dfused", (B, 512))
    R.bf16("B5 dX feature_processor.0", dfused, O.k_dx(dh1, P[hd + "feature_processor.0.weight"]))
    dz_o1 = buf("dz_o1", (B, 512))
    mean_o1, rstd_o1 = buf("mean_o1", (B,), torch.float32), buf("rstd_o1", (B,), torch.float32)
    ref, dg_o1, db_o1 = O.k_ln_bwd(dfused, y_o1, mean_o1, rstd_o1, P["fusion.output_projection.3.weight"], p=p)
    R.bf16("B6 LayerNorm bwd (output_proj)", dz_o1, ref)
    dtri = buf("dtri", (B, 512))
    R.bf16("B7 dX output_projection.0", dtri, O.k_dx(dz_o1, P["fusion.output_projection.0.weight"]))
    dz_t3 = buf("dz_t3", (B, 512))
    mean_t3, rstd_t3 = buf("mean_t3", (B,), torch.float32), buf("rstd_t3", (B,), torch.float32)
    ref, dg_t3, db_t3 = O.k_ln_bwd(dtri, y_t3, mean_t3, rstd_t3, P[tf + "final_fusion.3.weight"], p=p)
    R.bf16("B8 LayerNorm bwd (final_fusion)", dz_t3, ref)
    dpool = buf("dpool", (B, 512))
    R.bf16("B9 dX final_fusion.0", dpool, O.k_dx(dz_t3, P[tf + "final_fusion.0.weight"]))
    dobar = buf("dobar", (B, 512))
    R.bf16("B10 dX attention out_proj", dobar, O.k_dx(dpool, P[tf + "modality_attention.out_proj.weight"]))
    dqkv = buf("dqkv", (B, 2, 1536))
    R.bf16("B11 fused attention bwd (dqkv)", dqkv, O.k_tri_bwd(xtok, w_t, b_t, dobar, probs, mask=mk("tri_attn"), p=p))
    dxtok = buf("dxtok", (B, 2, 512))
    R.bf16("B12 dX in_proj", dxtok.reshape(2 * B, 512), O.k_dx(dqkv.reshape(2 * B, 1536), w_t))
    dav = buf("dav", (B, 256))
    R.bf16("B13 dX audiovisual_projection", dav, O.k_dx(dxtok[:, 0], P[tf + "audiovisual_projection.weight"]))
    dz_a2 = buf("dz_a2", (B, 256))
    ref, dg_a2, db_a2 = O.k_ln_bwd(dav, y_a2, mean_a2, rstd_a2, P[av + "fusion_layers.3.weight"], p=p)
    R.bf16("B14 LayerNorm bwd (fusion_layers)", dz_a2, ref)
    dcats = buf("dcats", (2 * B, 256))
    wf = P[av + "fusion_layers.0.weight"]
    R.bf16("B15 dX fusion_layers.0 (audio_att)", dcats[:B], O.k_dx(dz_a2, wf[:, :256]))
    R.bf16("B15 dX fusion_layers.0 (video_att)", dcats[B:], O.k_dx(dz_a2, wf[:, 256:]))
    davv = buf("davv", (2 * B, 256))
    R.bf16("B16 dX AV out_proj", davv, O.k_dx(dcats, P[av + "cross_attention.out_proj.weight"], p=p, keep=head_mask))
    davin = buf("davin", (2 * B, 256))
    R.bf16("B17 dX AV value projection", davin, O.k_dx(davv, w_in[512:]))

    # ------------------------------------------------------------------ backward: every parameter gradient of the step
    def dw(tag, name, dy, x, rows=None):
        gw, gb = O.k_dw(dy, x)
        if name.endswith("in_proj"):                      # packed MultiheadAttention parameters: in_proj_weight / in_proj_bias
            got_w, got_b = G[name + "_weight"], G[name + "_bias"]
        else:
            got_w, got_b = G[name + ".weight"], G[name + ".bias"]
        if rows is not None:
            got_w, got_b = got_w[rows], got_b[rows]
        R.f32(f"dW {tag}", got_w, gw); R.f32(f"db {tag}", got_b, gb)

    for i in range(3):
        gw = (dev_ref[:, i].double().t() @ e2[:, i].double()).float()
        R.f32(f"dW evidence_net.6 head {i}", G[ev(i, 6) + ".weight"], gw, 2e-5)     # d evidence from the kernel's own loss-gradient formula
        R.f32(f"db evidence_net.6 head {i}", G[ev(i, 6) + ".bias"], dev_ref[:, i].double().sum(0).float(), 2e-5)
        dw(f"evidence_net.3 head {i}", ev(i, 3), dz2[:, i], e1[:, i])
        dw(f"evidence_net.0 head {i}", ev(i, 0), de1[:, i], h2)
    dw("feature_processor.3", hd + "feature_processor.3", dh2, h1)
    dw("feature_processor.0", hd + "feature_processor.0", dh1, fused)
    dw("output_projection.0", "fusion.output_projection.0", dz_o1, tri)
    dw("final_fusion.0", tf + "final_fusion.0", dz_t3, pool)
    dw("attention out_proj", tf + "modality_attention.out_proj", dpool, obar)
    dw("trimodal in_proj", tf + "modality_attention.in_proj", dqkv.reshape(2 * B, 1536), xtok.reshape(2 * B, 512))
    dw("audiovisual_projection", tf + "audiovisual_projection", dxtok[:, 0], avf)
    dw("text_projection", tf + "text_projection", dxtok[:, 1], text)
    dw("fusion_layers.0", av + "fusion_layers.0", dz_a2, cat)
    dw("AV out_proj", av + "cross_attention.out_proj", dcats, avv)
    dw("AV in_proj (value rows)", av + "cross_attention.in_proj", davv, avin, rows=slice(512, 768))
    assert float(G[av + "cross_attention.in_proj_weight"][:512].abs().max()) == 0.0           # dead q / k rows: exact zeros
    dw("video_projection", av + "video_projection", davin[:B], video)
    dw("audio_projection", av + "audio_projection", davin[B:], audio)
    for tag, gname, dg, db in (("output_projection.3", "fusion.output_projection.3", dg_o1, db_o1), ("final_fusion.3", tf + "final_fusion.3", dg_t3, db_t3),
                               ("fusion_layers.3", av + "fusion_layers.3", dg_a2, db_a2)):
        R.f32(f"d gamma {tag}", G[gname + ".weight"], dg); R.f32(f"d beta {tag}", G[gname + ".bias"], db)
    print(f"\nteacher-forced launch checks, B = {B}, dropout {p} (loss {float(ld['total_loss']):.6f}):")
    print("\n".join(R.rows))
