"""GPU parity of the single operators of libmmdeer_hip.so, called through the C ABI.

fp32 compute (v_mfma_f32_16x16x4_f32) is compared with torch fp32 math on the same
inputs (tolerance 1e-4 abs after normalising by the reduction length); bf16 compute
is compared with the same math on bf16-rounded operands.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmdeer import _lib  # noqa: E402


def dev():
    return torch.device("cuda:0")


def stream():
    return torch.cuda.current_stream().cuda_stream


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def run_gemm(A, W, M, N, K, *, bias=None, relu=0, trans_a=0, trans_w=0, compute_f32=1, tile=-1, Y=None,
             mask_scale=1.0, bias_grad=None, c_dtype=None, drop_site=-1, drop_shift=0, regen_site=-1, p=0.0,
             seed=1, offset=0, lda=None, ldw=None, accumulate=0, C_init=None, splitk=1):
    lib = _lib.load()
    c_dtype = c_dtype or (torch.float32 if compute_f32 else torch.bfloat16)
    Cm = torch.zeros(M, N, dtype=c_dtype, device=dev()) if C_init is None else C_init.clone()
    a = _lib.GemmArgs()
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), Cm.data_ptr()
    a.bias = _lib.ptr(bias)
    a.bias_grad = _lib.ptr(bias_grad)
    a.Y = _lib.ptr(Y)
    a.M, a.N, a.K = M, N, K
    a.lda = lda if lda is not None else A.shape[1]
    a.ldw = ldw if ldw is not None else W.shape[1]
    a.ldc = N
    a.ldy = Y.shape[1] if Y is not None else 0
    a.a_f32 = int(A.dtype == torch.float32)
    a.w_f32 = int(W.dtype == torch.float32)
    a.c_f32 = int(c_dtype == torch.float32)
    a.y_f32 = int(Y is not None and Y.dtype == torch.float32)
    a.trans_a, a.trans_w, a.relu, a.accumulate = trans_a, trans_w, relu, accumulate
    a.compute_f32, a.tile = compute_f32, tile
    a.drop_site, a.drop_shift, a.regen_site = drop_site, drop_shift, regen_site
    a.dropout_p, a.mask_scale, a.seed, a.offset = p, mask_scale, seed, offset
    slab = None
    if splitk > 1:
        slab = torch.full((splitk * ((M * N + M + 3) // 4 * 4),), float("nan"), device=dev())
        a.splitk, a.slab = splitk, slab.data_ptr()
    a.stream = stream()
    _lib.check(lib.mmdeer_gemm(C.byref(a)))
    torch.cuda.synchronize()
    return Cm


def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("tile", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (130, 256, 84), (7, 512, 768), (257, 384, 256), (1, 64, 128)])
def test_gemm_fp32_nt(tile, M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    out = run_gemm(A, W, M, N, K, bias=b, relu=1, tile=tile)
    ref = torch.relu(A.double() @ W.double().t() + b.double()).float()
    assert (out - ref).abs().max().item() < 1e-4 * max(1.0, math.sqrt(K) / 8)


@pytest.mark.parametrize("tile", [0, 2])
def test_gemm_fp32_asymmetric_identity(tile):
    # A = I against an asymmetric W catches a transposed accumulator layout
    M = N = K = 128
    A = torch.eye(M, device=dev())
    W = (torch.arange(N * K, dtype=torch.float32, device=dev()).reshape(N, K) % 251) / 16.0
    out = run_gemm(A, W, M, N, K, tile=tile)
    assert torch.equal(out, W.t().contiguous())


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (130, 256, 84), (33, 512, 768), (300, 1536, 512), (1000, 520, 96),
                                   (2048, 768, 32)])
def test_gemm_bf16_nt(tile, M, N, K):
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.1), rnd(N, seed=6)
    # fp32 activations converted on the fly by the loader, bf16 weights
    out = run_gemm(A, bf(W), M, N, K, bias=b, compute_f32=0, tile=tile, c_dtype=torch.float32)
    ref = (bf(A).double() @ bf(W).double().t() + b.double()).float()
    assert (out - ref).abs().max().item() < 2e-3
    # bf16 activations in, bf16 out
    out2 = run_gemm(bf(A), bf(W), M, N, K, bias=b, compute_f32=0, tile=tile)
    assert (out2.float() - ref).abs().max().item() < 4e-2


@pytest.mark.parametrize("compute_f32", [1, 0])
@pytest.mark.parametrize("tile", [0, 2])
@pytest.mark.parametrize("M,N,K", [(64, 84, 128), (130, 256, 512), (7, 128, 64)])
def test_gemm_dx(compute_f32, tile, M, N, K):
    """dX[M,N] = dY[M,K] W[K,N], masked by (Y > 0) * scale: W is stored [K][N] -> trans_w."""
    dY, W, Y = rnd(M, K, seed=7), rnd(K, N, seed=8, scale=0.1), rnd(M, N, seed=9)
    if not compute_f32:
        dY, W, Y = bf(dY), bf(W), bf(Y)
    out = run_gemm(dY, W, M, N, K, trans_w=1, compute_f32=compute_f32, tile=tile, Y=Y, mask_scale=1.0 / 0.7, ldw=N)
    ref = (dY.double() @ W.double()) * (Y.double() > 0) / 0.7
    tol = 1e-4 if compute_f32 else 3e-2
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, math.sqrt(K) / 8)


@pytest.mark.parametrize("compute_f32", [1, 0])
@pytest.mark.parametrize("splitk", [1, 3, 8])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])     # bf16: 2 / 3 / 4 = the LDS-DMA kernel on 128x128 / 256x256 / 256x128 tiles
@pytest.mark.parametrize("Bt,Nl,Kl", [(64, 128, 64), (100, 256, 84), (7, 64, 256), (513, 384, 256), (1100, 64, 128),
                                      (1024, 512, 768), (2048, 384, 256), (96, 64, 128), (4096, 520, 264)])
def test_gemm_dw_and_bias_grad(compute_f32, splitk, tile, Bt, Nl, Kl):
    """dW[Nl,Kl] = dY[Bt,Nl]^T X[Bt,Kl] and db = column sums of dY (both operands transposed by the loader)."""
    dY, X = rnd(Bt, Nl, seed=10), rnd(Bt, Kl, seed=11)
    if not compute_f32:
        dY = bf(dY)
    dbias = torch.zeros(Nl, device=dev())
    # X stays fp32 in bf16 mode for the Kl=84 case: the loader converts (first-layer inputs are user fp32 tensors)
    Xs = X if (compute_f32 or Kl % 8) else bf(X)
    out = run_gemm(dY, Xs, Nl, Kl, Bt, trans_a=1, trans_w=1, compute_f32=compute_f32, tile=tile, bias_grad=dbias,
                   c_dtype=torch.float32, lda=Nl, ldw=Kl, splitk=splitk)
    Xr = Xs.double() if compute_f32 else bf(Xs).double()
    ref = dY.double().t() @ Xr
    tol = 1e-4 if compute_f32 else 2e-2
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, math.sqrt(Bt) / 4)
    assert (dbias.double() - dY.double().sum(0)).abs().max().item() < tol * max(1.0, math.sqrt(Bt) / 4)


@pytest.mark.parametrize("tile", [2, 3, 4])
@pytest.mark.parametrize("splitk", [1, 4])
def test_gemm_dw_padded_rows(tile, splitk):
    """dW against an activation block whose 84 valid columns sit in 128-element rows (the padded audio block):
    the columns beyond 84 must not reach the result."""
    Bt, Nl, Kl, ld = 1024, 256, 84, 128
    dY = bf(rnd(Bt, Nl, seed=20))
    X = bf(rnd(Bt, ld, seed=21))          # columns 84..127 hold garbage on purpose
    dbias = torch.zeros(Nl, device=dev())
    out = run_gemm(dY, X, Nl, Kl, Bt, trans_a=1, trans_w=1, compute_f32=0, tile=tile, bias_grad=dbias,
                   c_dtype=torch.float32, lda=Nl, ldw=ld, splitk=splitk)
    ref = dY.double().t() @ X[:, :Kl].double()
    assert out.shape == (Nl, Kl)
    assert (out.double() - ref).abs().max().item() < 2e-2 * math.sqrt(Bt) / 4
    assert (dbias.double() - dY.double().sum(0)).abs().max().item() < 2e-2 * math.sqrt(Bt) / 4


@pytest.mark.parametrize("N", [1024, 768])
def test_gemm_256_tile_epilogue_matches_64_tile(N):
    """The 256-row forward kernel (bias + ReLU + dropout, bf16 in/out) against the 64x64 kernel on the same
    inputs: the same products summed in a different order, the same dropout decisions.  N = 1024 runs as 256x256
    tiles, N = 768 as 256x192 tiles (16 of them against 12)."""
    M, K = 777, 512
    A, W, b = bf(rnd(M, K, seed=30)), bf(rnd(N, K, seed=31, scale=0.1)), rnd(N, seed=32)
    kw = dict(bias=b, relu=1, compute_f32=0, drop_site=4, p=0.3, seed=99, offset=7)
    o3 = run_gemm(A, W, M, N, K, tile=3, **kw).float()
    o0 = run_gemm(A, W, M, N, K, tile=0, **kw).float()
    assert ((o3 == 0) == (o0 == 0)).float().mean().item() > 0.999   # identical keep decisions (up to relu ties)
    assert (o3 - o0).abs().max().item() < 5e-2
    ref = torch.relu(A.double() @ W.double().t() + b.double())
    kept = o3 != 0
    assert ((o3.double() - ref / 0.7).abs() * kept).max().item() < 8e-2
    # fp32 output: the register-direct epilogue of the same kernel
    o3f = run_gemm(A, W, M, N, K, tile=3, c_dtype=torch.float32, **kw)
    assert ((o3f == 0) == (o3 == 0)).float().mean().item() > 0.999
    assert ((o3f - o3).abs() <= o3f.abs() * 2.0 ** -8 + 1e-6).all()        # o3 is o3f rounded to bf16
    assert ((o3f.double() - ref / 0.7).abs() * (o3f != 0)).max().item() < 2e-2


def test_gemm_dropout_matches_mask_dump():
    lib = _lib.load()
    M, N, K, p, seed, off, site = 96, 256, 64, 0.3, 1234, 5, 4
    A, W = rnd(M, K, seed=12), rnd(N, K, seed=13)
    out = run_gemm(A, W, M, N, K, relu=1, drop_site=site, p=p, seed=seed, offset=off)
    mask = torch.empty(M, N, dtype=torch.uint8, device=dev())
    _lib.check(lib.mmdeer_dropout_mask(site, M, N, p, seed, off, mask.data_ptr(), stream()))
    torch.cuda.synchronize()
    ref = torch.relu(A @ W.t()) * mask.float() / (1 - p)
    assert (out - ref).abs().max().item() < 1e-4
    keep = mask.float().mean().item()
    assert abs(keep - 0.7) < 0.02
    # a different offset gives a different mask
    mask2 = torch.empty_like(mask)
    _lib.check(lib.mmdeer_dropout_mask(site, M, N, p, seed, off + 1, mask2.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert (mask != mask2).float().mean().item() > 0.2
    # head-granular dropout: one decision per 32 columns
    out3 = run_gemm(A, W, M, N, K, drop_site=1, drop_shift=5, p=p, seed=seed, offset=off)
    m3 = torch.empty(M, N // 32, dtype=torch.uint8, device=dev())
    _lib.check(lib.mmdeer_dropout_mask(1, M, N // 32, p, seed, off, m3.data_ptr(), stream()))
    torch.cuda.synchronize()
    ref3 = (A @ W.t()) * m3.float().repeat_interleave(32, dim=1) / (1 - p)
    assert (out3 - ref3).abs().max().item() < 1e-4


@pytest.mark.parametrize("act_f32", [1, 0])
@pytest.mark.parametrize("M,N", [(5, 256), (64, 512), (1000, 512)])
def test_layernorm_fwd_bwd(act_f32, M, N):
    lib = _lib.load()
    dt = torch.float32 if act_f32 else torch.bfloat16
    y = torch.relu(rnd(M, N, seed=20)).to(dt)
    g, b = 1 + 0.1 * rnd(N, seed=21), 0.05 * rnd(N, seed=22)
    out = torch.empty(M, N, dtype=dt, device=dev())
    out32 = torch.empty(M, N, device=dev())
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    _lib.check(lib.mmdeer_layernorm_fwd(y.data_ptr(), out.data_ptr(), out32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        g.data_ptr(), b.data_ptr(), M, N, act_f32, stream()))
    yr = y.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(yr, (N,), g.double(), b.double(), 1e-5)
    torch.cuda.synchronize()
    assert (out32.double() - ref).abs().max().item() < 2e-5
    assert (out.double() - ref).abs().max().item() < (2e-5 if act_f32 else 3e-2)
    # backward (+ ReLU/dropout mask of y)
    dout = rnd(M, N, seed=23).to(dt)
    dz = torch.empty(M, N, dtype=dt, device=dev())
    dgam, dbet = torch.empty(N, device=dev()), torch.empty(N, device=dev())
    nparts = lib.mmdeer_layernorm_bwd_nparts(M)
    partial = torch.empty(nparts * 2 * N, device=dev())
    _lib.check(lib.mmdeer_layernorm_bwd(dout.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(),
                                        dz.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), partial.data_ptr(), M, N,
                                        act_f32, 1.0 / 0.7, stream()))
    torch.cuda.synchronize()
    gg = torch.nn.Parameter(g.double())
    bb = torch.nn.Parameter(b.double())
    ref = torch.nn.functional.layer_norm(yr, (N,), gg, bb, 1e-5)
    ref.backward(dout.double())
    dz_ref = yr.grad * (y.double() > 0) / 0.7
    tol = 1e-4 if act_f32 else 5e-2
    assert (dz.double() - dz_ref).abs().max().item() < tol
    assert (dgam.double() - gg.grad).abs().max().item() < tol * max(1.0, math.sqrt(M) / 4)
    assert (dbet.double() - bb.grad).abs().max().item() < tol * max(1.0, math.sqrt(M) / 4)


def _attn_ref(qkv, p_drop=None):
    """explicit 2-token attention on a (2B,1536) matrix, returns pooled context, probs, head-mean weights"""
    B = qkv.shape[0] // 2
    x = qkv.view(B, 2, 3, 8, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)  # (B,8,2,64)
    s = (q * math.sqrt(1 / 64)) @ k.transpose(-1, -2)
    p = torch.softmax(s, dim=-1)
    pd = p if p_drop is None else p * p_drop
    o = pd @ v                                    # (B,8,2,64)
    obar = o.mean(dim=2).reshape(B, 512)
    return obar, p, pd.mean(dim=1)


@pytest.mark.parametrize("act_f32", [1, 0])
@pytest.mark.parametrize("training", [0, 1])
@pytest.mark.parametrize("B", [1, 37, 256])
def test_trimodal_attention(act_f32, training, B):
    lib = _lib.load()
    dt = torch.float32 if act_f32 else torch.bfloat16
    qkv = rnd(2 * B, 1536, seed=30).to(dt)
    obar = torch.empty(B, 512, dtype=dt, device=dev())
    probs = torch.empty(B, 8, 4, device=dev())
    w = torch.empty(B, 2, 2, device=dev())
    avw = torch.empty(B, 2, device=dev())
    p, seed, off = 0.3, 77, 3
    _lib.check(lib.mmdeer_trimodal_attn_fwd(qkv.data_ptr(), obar.data_ptr(), probs.data_ptr(), w.data_ptr(), avw.data_ptr(),
                                            B, act_f32, training, p, seed, off, stream()))
    drop = None
    if training:
        m = torch.empty(B, 32, dtype=torch.uint8, device=dev())
        _lib.check(lib.mmdeer_dropout_mask(3, B, 32, p, seed, off, m.data_ptr(), stream()))
        drop = m.double().view(B, 8, 2, 2) / (1 - p)
        mav = torch.empty(2 * B, 8, dtype=torch.uint8, device=dev())
        _lib.check(lib.mmdeer_dropout_mask(1, 2 * B, 8, p, seed, off, mav.data_ptr(), stream()))
    torch.cuda.synchronize()
    qr = qkv.double().requires_grad_(True)
    obar_ref, p_ref, w_ref = _attn_ref(qr, drop)
    tol = 2e-5 if act_f32 else 3e-2
    assert (obar.double() - obar_ref).abs().max().item() < tol
    assert (probs.double().view(B, 8, 2, 2) - p_ref).abs().max().item() < 1e-5
    assert (w.double() - w_ref).abs().max().item() < 1e-5
    if training:
        ref_av = torch.stack([mav[:B].double().mean(1), mav[B:].double().mean(1)], dim=1) / (1 - p)
        assert (avw.double() - ref_av).abs().max().item() < 1e-6
    else:
        assert torch.equal(avw, torch.ones_like(avw))
    # backward
    dob = rnd(B, 512, seed=31).to(dt)
    dqkv = torch.empty(2 * B, 1536, dtype=dt, device=dev())
    _lib.check(lib.mmdeer_trimodal_attn_bwd(qkv.data_ptr(), dob.data_ptr(), probs.data_ptr(), dqkv.data_ptr(), B, act_f32,
                                            training, p, seed, off, stream()))
    torch.cuda.synchronize()
    obar_ref.backward(dob.double())
    assert (dqkv.double() - qr.grad).abs().max().item() < (5e-5 if act_f32 else 5e-2)


@pytest.mark.parametrize("training", [0, 1])
@pytest.mark.parametrize("B", [1, 37, 128, 300, 1024])
def test_fused_projection_attention(training, B):
    """tri_fused.hip (in_proj + 2-token attention in one kernel, q|k|v on chip) against fp64 math on the same bf16
    operands: obar, probabilities, returned weights, the optional q|k|v tile; backward (recompute) dqkv against autograd.
    B = 1 / 37 / 300: ragged row tiles (rows beyond 2B are never stored); 1024: several row tiles per XCD group."""
    lib = _lib.load()
    x = rnd(2 * B, 512, seed=50, scale=1.0).bfloat16()
    w = rnd(1536, 512, seed=51, scale=0.06)                      # fp32 master in_proj_weight
    bias = rnd(1536, seed=52, scale=0.2)
    whm = torch.empty(1536 * 512, dtype=torch.bfloat16, device=dev())
    _lib.check(lib.mmdeer_pack_qkv_headmajor(w.data_ptr(), whm.data_ptr(), stream()))
    obar = torch.full((B, 512), float("nan"), dtype=torch.bfloat16, device=dev())
    probs = torch.full((B, 8, 4), float("nan"), device=dev())
    attn_w = torch.empty(B, 2, 2, device=dev())
    avw = torch.empty(B, 2, device=dev())
    qkv = torch.full((2 * B, 1536), float("nan"), dtype=torch.bfloat16, device=dev())
    p, seed, off = 0.3, 91, 5
    _lib.check(lib.mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar.data_ptr(), probs.data_ptr(),
                                             qkv.data_ptr(), attn_w.data_ptr(), avw.data_ptr(), B, training, p, seed, off, stream()))
    drop = None
    if training:
        m = torch.empty(B, 32, dtype=torch.uint8, device=dev())
        _lib.check(lib.mmdeer_dropout_mask(3, B, 32, p, seed, off, m.data_ptr(), stream()))
        drop = m.double().view(B, 8, 2, 2) / (1 - p)
    torch.cuda.synchronize()
    # head-major image: row 96 wn + 32 part + dd of head h  <-  in_proj_weight[part * 512 + 64 h + 32 wn + dd]
    img = whm.view(8, 2, 3, 32, 512)
    want = w.bfloat16().view(3, 8, 2, 32, 512).permute(1, 2, 0, 3, 4)
    assert torch.equal(img, want)
    xr = x.double().requires_grad_(True)
    wr = w.bfloat16().double()
    qkv_ref = xr @ wr.t() + bias.double()
    obar_ref, p_ref, w_ref = _attn_ref(qkv_ref, drop)
    assert torch.isfinite(obar.float()).all() and torch.isfinite(probs).all() and torch.isfinite(qkv.float()).all()
    assert (qkv.double() - qkv_ref).abs().max().item() < 3e-2            # bf16 rounding of O(1) values
    assert (probs.double().view(B, 8, 2, 2) - p_ref).abs().max().item() < 6e-3   # scores on the matrix pipe: q, k rounded to bf16
    assert (obar.double() - obar_ref).abs().max().item() < 1.2e-2 * max(1.0, obar_ref.abs().max().item())   # bf16 output
    assert (attn_w.double() - w_ref).abs().max().item() < 6e-3
    # without the optional q|k|v tile the results are the same bits
    obar2 = torch.empty_like(obar); probs2 = torch.empty_like(probs)
    _lib.check(lib.mmdeer_trimodal_fused_fwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), obar2.data_ptr(), probs2.data_ptr(),
                                             None, None, None, B, training, p, seed, off, stream()))
    torch.cuda.synchronize()
    assert torch.equal(obar.view(torch.int16), obar2.view(torch.int16)) and torch.equal(probs, probs2)
    # backward: recomputed q|k|v, d(obar) -> dqkv
    dob = rnd(B, 512, seed=53).bfloat16()
    dqkv = torch.full((2 * B, 1536), float("nan"), dtype=torch.bfloat16, device=dev())
    _lib.check(lib.mmdeer_trimodal_fused_bwd(x.data_ptr(), whm.data_ptr(), bias.data_ptr(), dob.data_ptr(), probs.data_ptr(),
                                             dqkv.data_ptr(), B, training, p, seed, off, stream()))
    torch.cuda.synchronize()
    qr = qkv_ref.detach().requires_grad_(True)
    o2, _, _ = _attn_ref(qr, drop)
    o2.backward(dob.double())
    assert torch.isfinite(dqkv.float()).all()
    assert (dqkv.double() - qr.grad).abs().max().item() < 3e-2 * max(1.0, qr.grad.abs().max().item())
    # and it agrees with the unfused attention-backward kernel run on the stored q|k|v tile (bf16 rounding of q, k, v apart)
    dq2 = torch.empty_like(dqkv)
    _lib.check(lib.mmdeer_trimodal_attn_bwd(qkv.data_ptr(), dob.data_ptr(), probs.data_ptr(), dq2.data_ptr(), B, 0, training,
                                            p, seed, off, stream()))
    torch.cuda.synchronize()
    assert (dqkv.double() - dq2.double()).abs().max().item() < 3e-2 * max(1.0, qr.grad.abs().max().item())


def test_nig_loss_vs_oracle(golden_dir):
    import os
    from oracle import deer_oracle as O
    lib = _lib.load()
    g = dict(np.load(os.path.join(golden_dir, "loss_cases.npz")))
    e = torch.from_numpy(g["evidence"])
    y = torch.from_numpy(g["targets"])
    for tag, sl in (("reg", slice(5, 64)), ("one", slice(7, 8)), ("two", slice(9, 11)), ("all", slice(0, 64))):
        ec = e[sl].clone().double().requires_grad_(True)
        mu, nu, alpha, beta, *_ = O.nig_activations(ec.float())
        nig = [t.detach().to(dev()).contiguous() for t in (mu, nu, alpha, beta)]
        yt = y[sl].to(dev()).contiguous()
        B = yt.shape[0]
        stats = torch.empty(lib.mmdeer_nig_stats_elems(B), device=dev())
        grads = torch.zeros(4, B, 3, device=dev())
        loss_out = torch.empty(20, device=dev())
        bins = torch.empty(30, dtype=torch.int32, device=dev())
        from mmdeer.model import make_loss_cfg
        cfg = make_loss_cfg()
        _lib.check(lib.mmdeer_nig_loss(*(t.data_ptr() for t in nig), yt.data_ptr(), stats.data_ptr(),
                                       grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(),
                                       loss_out.data_ptr(), bins.data_ptr(), B, C.byref(cfg), stream()))
        torch.cuda.synchronize()
        lo = loss_out.cpu().numpy()
        # golden values captured from the reference
        names = ["total_loss", "nll_loss", "reg_loss", "kl_loss", "ece_loss"]
        for i, d in enumerate(O.DIM_NAMES):
            for j, n in enumerate(names):
                ref = float(g[f"multitask.{tag}.{d}_{n}"])
                if np.isfinite(ref):
                    assert lo[i * 5 + j] == pytest.approx(ref, rel=1e-4, abs=1e-5), (tag, d, n)
                else:   # the extreme-evidence rows (+-25, -104): inf / nan must come out as the same inf / nan
                    got = float(lo[i * 5 + j])
                    assert (np.isnan(ref) and np.isnan(got)) or got == ref, (tag, d, n, got, ref)
        if np.isfinite(float(g[f"multitask.{tag}.total_loss"])):
            assert lo[16] == pytest.approx(float(g[f"multitask.{tag}.total_loss"]), rel=1e-4)
            assert lo[15] == pytest.approx(float(g[f"multitask.{tag}.cross_dim_loss"]), rel=1e-4, abs=1e-7)
        # bin populations: exact integers vs the oracle's boolean masks on the same fp32 inputs
        pred = {}
        mu32, nu32, al32, be32 = (t.detach() for t in (mu, nu, alpha, beta))
        for i, d in enumerate(O.DIM_NAMES):
            pred[f"{d}_mu"], pred[f"{d}_nu"] = mu32[:, i:i + 1], nu32[:, i:i + 1]
            pred[f"{d}_alpha"], pred[f"{d}_beta"] = al32[:, i:i + 1], be32[:, i:i + 1]
        ld = O.multitask_loss(pred, y[sl])
        ref_counts = np.array([ld[f"{d}__bin_counts"] for d in O.DIM_NAMES]).reshape(-1)
        np.testing.assert_array_equal(bins.cpu().numpy(), ref_counts)
        if tag == "reg":
            # gradients wrt (gamma, nu, alpha, beta) against autograd over the oracle in fp64
            t64 = [t.detach().double().requires_grad_(True) for t in (mu32, nu32, al32, be32)]
            p64 = {}
            for i, d in enumerate(O.DIM_NAMES):
                p64[f"{d}_mu"], p64[f"{d}_nu"] = t64[0][:, i:i + 1], t64[1][:, i:i + 1]
                p64[f"{d}_alpha"], p64[f"{d}_beta"] = t64[2][:, i:i + 1], t64[3][:, i:i + 1]
            O.multitask_loss(p64, y[sl].double())["total_loss"].backward()
            for k in range(4):
                ref = t64[k].grad
                got = grads[k].cpu().double()
                assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), k


def test_convert_roundtrip():
    lib = _lib.load()
    x = rnd(1000, 4, seed=40)
    h = torch.empty(1000, 4, dtype=torch.bfloat16, device=dev())
    _lib.check(lib.mmdeer_convert(x.data_ptr(), 1, h.data_ptr(), 0, x.numel(), stream()))
    torch.cuda.synchronize()
    assert torch.equal(h, x.to(torch.bfloat16))   # round-to-nearest-even, bit-exact


def test_errors_are_reported():
    lib = _lib.load()
    a = _lib.GemmArgs()
    assert lib.mmdeer_gemm(C.byref(a)) == -1
    assert b"gemm" in lib.mmdeer_last_error()
