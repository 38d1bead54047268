"""The layer-chain kernel and the weight-image writer as C-ABI operators (include/mmdeer.h: mmdeer_chain, mmdeer_repack; host side
mmdeer/chainops.py): against plain PyTorch / numpy restatements, on ragged row counts, both workgroup sizes and randomly drawn segment
tables.  (Their use by Stack B's training step -- and its bit-for-bit equality with the launch-by-launch plan -- is in
test_gpu_stackb.py; their use by Stack C's step in test_gpu_model.py and test_gpu_bf16_layers.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,ts", [(37, 0), (300, 32)])
def test_layer_chain_operator_against_torch(rows, ts):
    """mmdeer_chain on its own, against plain PyTorch fp32 on the same bf16-rounded operands: a Linear-ReLU-LayerNorm stem, a residual
    block x + LayerNorm(ReLU(Linear x)) and an output Linear whose 512 columns are stashed into two tensors -- ragged row counts,
    both workgroup sizes.  bf16 storage between the layers: 2e-2 of each tensor's largest element."""
    from mmdeer.chainops import Chain, FragImages
    from mmdeer.opseq import Exec
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(rows)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = rnd(rows, 256).bfloat16()
    W0, W1, W2 = (rnd(256, 256, sc=0.08).bfloat16(), rnd(256, 256, sc=0.08).bfloat16(), rnd(512, 256, sc=0.08).bfloat16())
    b0, b1, b2 = rnd(256, sc=0.1), rnd(256, sc=0.1), rnd(512, sc=0.1)
    g0, be0, g1, be1 = 1 + rnd(256, sc=0.1), rnd(256, sc=0.1), 1 + rnd(256, sc=0.1), rnd(256, sc=0.1)
    F = FragImages(dev)
    F.add("w0", W0, 256, 256); F.add("w1", W1, 256, 256); F.add("w2", W2, 512, 256)
    F.finish(); F.refresh()
    new = lambda *s, d=torch.bfloat16: torch.zeros(*s, dtype=d, device=dev)
    y0, h0, y1, h1, oa, ob = new(rows, 256), new(rows, 256), new(rows, 256), new(rows, 256), new(rows, 256), new(rows, 256)
    m0, r0, m1, r1 = (new(rows, d=torch.float32) for _ in range(4))
    ex = Exec("bf16", None)
    ch = Chain(ex, x, 256, 256, rows, ts=ts)
    ch.seg(F("w0"), 256, 256, bias=b0, relu=1).end(256, stash=y0, ld_stash=256, ln=(g0, be0, h0, m0, r0))
    ch.seg(F("w1"), 256, 256, bias=b1, relu=1).end(256, stash=y1, ld_stash=256, ln=(g1, be1, h1, m1, r1), residual=1)
    ch.seg(F("w2"), 512, 256, bias=b2).end(512, stash=oa, ld_stash=256, stash2=ob, split=256)
    assert ch.workgroups() == (rows + (ts or 16) - 1) // (ts or 16)
    ch.launch()
    torch.cuda.synchronize()
    f = lambda t: t.float()
    bf = lambda t: t.bfloat16().float()
    ry0 = bf(torch.relu(f(x) @ f(W0).T + b0))
    rh0 = bf(torch.nn.functional.layer_norm(ry0, (256,), g0, be0, 1e-5))
    ry1 = bf(torch.relu(rh0 @ f(W1).T + b1))
    rh1 = bf(bf(torch.nn.functional.layer_norm(ry1, (256,), g1, be1, 1e-5)) + rh0)
    ro = rh1 @ f(W2).T + b2
    for name, got, ref in (("y0", y0, ry0), ("h0", h0, rh0), ("y1", y1, ry1), ("h1", h1, rh1), ("out[:256]", oa, ro[:, :256]), ("out[256:]", ob, ro[:, 256:])):
        scale = float(ref.abs().max())
        assert float((f(got) - ref).abs().max()) <= 2e-2 * scale, (name, float((f(got) - ref).abs().max()), scale)
    assert torch.allclose(m0, ry0.mean(1), atol=1e-4) and torch.allclose(r0, 1 / torch.sqrt(ry0.var(1, unbiased=False) + 1e-5), rtol=1e-4)


@pytest.mark.parametrize("seed", list(range(10)))
def test_layer_chain_operator_random_tables(seed):
    """mmdeer_chain over randomly drawn segment tables (every instantiated depth K / 64 in {1, 2, 4, 6, 8, 12}, 64- and 128-column tiles,
    layers of one to three segments writing column ranges of one panel and reading column ranges of the previous one, ReLU, masks,
    LayerNorm with and without the residual, ragged row counts, both workgroup sizes) against plain PyTorch on the same
    bf16-rounded operands."""
    from mmdeer.chainops import Chain, FragImages
    from mmdeer.opseq import Exec
    dev = torch.device("cuda")
    rng = np.random.default_rng(100 + seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ts = int(rng.choice([16, 32]))
    cap = 768 if ts == 16 else 512
    rows = int(rng.integers(1, 5 * ts))
    K0 = int(rng.choice([64, 128, 256, 384, 512] + ([768] if ts == 16 else [])))
    x = rnd(rows, K0).bfloat16()
    F = FragImages(dev)
    layers, width = [], K0
    for li in range(int(rng.integers(2, 5))):
        segs, nout = [], 0
        ln = bool(rng.integers(0, 2))
        nseg = 1 if ln else int(rng.integers(1, 4))
        for si in range(nseg):
            # K: a window of the input panel; N: multiples of 64 (64-column tiles, N % 128 != 0, exist for K = 128 and 256)
            ks = [k for k in (64, 128, 256, 384, 512, 768) if k <= width and (k != 768 or ts == 16)]
            K = int(rng.choice(ks))
            kin = int(rng.integers(0, (width - K) // 64 + 1)) * 64
            ns = [n for n in ((256, 512) if ln else (64, 128, 192, 256, 384, 512)) if nout + n <= cap and (n % 128 == 0 or K in (128, 256)) and (n // (128 if n % 128 == 0 else 64)) <= 4]
            if not ns:
                break
            N = int(rng.choice(ns))
            W = rnd(N, K, sc=1.0 / np.sqrt(K)).bfloat16()
            b = rnd(N, sc=0.2) if rng.integers(0, 2) else None
            relu = int(rng.integers(0, 2))
            mask = rnd(rows, N).bfloat16() if (rng.integers(0, 3) == 0 and not ln) else None
            key = f"w{li}.{si}"
            F.add(key, W, N, K)
            segs.append(dict(key=key, W=W, b=b, N=N, K=K, kin=kin, nout_off=nout, relu=relu, mask=mask, ms=float(rng.choice([1.0, 1.25]))))
            nout += N
        if not segs:
            break
        residual = bool(ln and nout == width and rng.integers(0, 2))
        gam, bet = (1 + rnd(nout, sc=0.1), rnd(nout, sc=0.1)) if ln else (None, None)
        layers.append(dict(segs=segs, nout=nout, ln=ln, residual=residual, gam=gam, bet=bet))
        width = nout
    F.finish(); F.refresh()
    ex = Exec("bf16", None)
    ch = Chain(ex, x, K0, K0, rows, ts=ts)
    new = lambda *s, d=torch.bfloat16: torch.zeros(*s, dtype=d, device=dev)
    outs = []
    for L in layers:
        for sg in L["segs"]:
            ch.seg(F(sg["key"]), sg["N"], sg["K"], bias=sg["b"], relu=sg["relu"], kin=sg["kin"], nout_off=sg["nout_off"],
                   mask=sg["mask"], ldm=sg["N"], mscale=sg["ms"])
        y = new(rows, L["nout"])
        if L["ln"]:
            h, m, r = new(rows, L["nout"]), new(rows, d=torch.float32), new(rows, d=torch.float32)
            ch.end(L["nout"], stash=y, ld_stash=L["nout"], ln=(L["gam"], L["bet"], h, m, r), residual=int(L["residual"]))
            outs.append((y, h))
        else:
            ch.end(L["nout"], stash=y, ld_stash=L["nout"])
            outs.append((y, None))
    ch.launch()
    torch.cuda.synchronize()
    bf = lambda t: t.bfloat16().float()
    cur = x.float()
    for L, (y, h) in zip(layers, outs):
        ref = torch.zeros(rows, L["nout"], device=dev)
        for sg in L["segs"]:
            v = cur[:, sg["kin"]:sg["kin"] + sg["K"]] @ sg["W"].float().T
            if sg["b"] is not None:
                v = v + sg["b"]
            if sg["relu"]:
                v = torch.relu(v)
            if sg["mask"] is not None:
                v = torch.where(sg["mask"].float() > 0, v * sg["ms"], torch.zeros_like(v))
            ref[:, sg["nout_off"]:sg["nout_off"] + sg["N"]] = v
        ref = bf(ref)
        scale = max(float(ref.abs().max()), 1e-3)
        assert float((y.float() - ref).abs().max()) <= 2e-2 * scale, ("pre", seed, float((y.float() - ref).abs().max()), scale)
        if L["ln"]:
            n = bf(torch.nn.functional.layer_norm(y.float(), (L["nout"],), L["gam"], L["bet"], 1e-5))      # from the chain's own rows: no compounding
            if L["residual"]:
                n = bf(n + cur)
            assert float((h.float() - n).abs().max()) <= 2e-2 * max(float(n.abs().max()), 1e-3), ("ln", seed)
            cur = h.float()
        else:
            cur = y.float()


def test_repack_operator_writes_the_documented_layouts():
    """mmdeer_repack against the index formulas of include/mmdeer.h / csrc/chain.h, restated in numpy: fragment-major images of S and
    of S^T (16-byte granule ((wt * (K / 64) + kt) * 2 + c) * 64 + lane = M[16 wt + (lane & 15)][64 kt + 32 c + 8 (lane >> 4) .. + 8)),
    zero columns past cols_valid with an odd source stride (the 84-wide audio projection), row-major placements into a zero-filled
    area (a column slice, padded rows, a block-diagonal stack) -- what mmdeer/chainops.py: FragImages asks of it."""
    from mmdeer.chainops import FragImages
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(3)
    S = (torch.randn(192, 128, generator=g)).bfloat16().to(dev)
    A = (torch.randn(32, 84, generator=g)).bfloat16().to(dev)             # 84 valid columns, rows of 84 elements (not 16-byte aligned)
    Wn = (torch.randn(16, 67, generator=g)).bfloat16().to(dev)            # a 64-column slice of 67-wide rows
    F = FragImages(dev)
    F.add("S", S, 192, 128)
    F.add("S.T", S, 192, 128, transpose=1)                                 # the image of S^T [128][192]
    F.add("A", A, 32, 128, ld_src=84, cols_valid=84)
    F.area("slice", 16, 64); F.place("slice", Wn, 16, 64, ld_src=67)
    F.area("bd", 16, 256)
    for d in range(2):
        F.place("bd", S[8 * d:8 * d + 4], 4, 128, row0=8 * d, col0=128 * d)
    F.finish(); F.refresh()
    torch.cuda.synchronize()

    def frag(M):                                                            # numpy restatement of the fragment-major order
        R, K = M.shape
        out = np.zeros(R * K, dtype=M.dtype)
        for wt in range(R // 16):
            for kt in range(K // 64):
                for c in range(2):
                    for lane in range(64):
                        l = ((wt * (K // 64) + kt) * 2 + c) * 64 + lane
                        r, k0 = 16 * wt + (lane & 15), 64 * kt + 32 * c + 8 * (lane >> 4)
                        out[8 * l:8 * l + 8] = M[r, k0:k0 + 8]
        return out
    raw = lambda t: t.cpu().view(torch.int16).numpy()
    assert np.array_equal(raw(F("S")), frag(raw(S)))
    assert np.array_equal(raw(F("S.T")), frag(np.ascontiguousarray(raw(S).T)))
    assert np.array_equal(raw(F("S", 64)), frag(raw(S)[64:]))              # a sub-image: 16-row blocks of the matrix are contiguous
    Ap = np.zeros((32, 128), dtype=np.int16); Ap[:, :84] = raw(A)
    assert np.array_equal(raw(F("A")), frag(Ap))
    assert np.array_equal(raw(F.mat("slice")), raw(Wn)[:, :64])
    bd = np.zeros((16, 256), dtype=np.int16)
    for d in range(2):
        bd[8 * d:8 * d + 4, 128 * d:128 * d + 128] = raw(S)[8 * d:8 * d + 4]
    assert np.array_equal(raw(F.mat("bd")), bd)


def test_layer_chain_operator_refuses_what_it_does_not_instantiate():
    """mmdeer_chain validates its tables on the host: unsupported widths, panels that do not fit, a residual on a layer that changes
    the geometry all fail with a message instead of launching."""
    import ctypes as C
    from mmdeer import _lib
    lib = _lib.load()
    x = torch.zeros(32, 256, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(256 * 256, dtype=torch.bfloat16, device="cuda")
    a = _lib.ChainArgs()
    a.X, a.ldx, a.K0, a.rows, a.nseg, a.stream = x.data_ptr(), 256, 256, 32, 1, _lib.current_stream()
    s = a.seg[0]
    s.W, s.N, s.K, s.end_layer, s.nout, s.drop_site = w.data_ptr(), 256, 256, 1, 256, -1
    assert lib.mmdeer_chain(C.byref(a)) == 0
    s.K = 320                                              # not an instantiated depth
    assert lib.mmdeer_chain(C.byref(a)) != 0 and b"chain" in lib.mmdeer_last_error()
    s.K, s.N, s.nout = 256, 96, 96                         # N % 64
    assert lib.mmdeer_chain(C.byref(a)) != 0
    s.N, s.nout, s.res_add = 256, 256, 1                   # a bypass copy nobody wrote
    assert lib.mmdeer_chain(C.byref(a)) != 0
    s.res_add, a.samples_per_workgroup = 0, 24
    assert lib.mmdeer_chain(C.byref(a)) != 0
    torch.cuda.synchronize()
