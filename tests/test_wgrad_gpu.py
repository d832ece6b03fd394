"""tsg_wgrad_f32s (csrc/wgrad_split.hip): dW = dY^T [X | shifted H] in split precision, operands converted on load --
vs a float64 product of the same operands (fp32-GEMM-level error), vs the library split-GEMM path it replaces, the shifted
second segment in both sequence layouts, strided operands, determinism and the argument checks."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("M,N,K", [
    (32, 256, 128),            # one chunk, one tile, no split
    (2560, 1024, 1024),        # word-side projections: 8 row ranges of 10 chunks
    (16384, 1024, 1024),       # W_a at the north-star shape
    (8192, 512, 1024),         # boundary head W1 (video half)
    (4128, 256, 384),          # ragged row ranges (129 chunks)
])
def test_wgrad_matches_float64(M, N, K):
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, N, generator=g).cuda()
    B = torch.randn(M, K, generator=g).cuda()
    C = F.wgrad_f32s(A, B)
    assert C.shape == (1, N, K)
    ref = A.double().t() @ B.double()
    assert _rel(C[0], ref) < 1e-5                       # fp32 rocBLAS on the same product: 1.4e-6 .. 6.4e-6 of max|C|
    assert torch.equal(C, F.wgrad_f32s(A, B))           # fixed summation order


def test_wgrad_wide_dynamic_range_and_zero_rows():
    """lo planes matter: operands whose hi*hi product alone is off by 2^-9; zero rows contribute nothing."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(3)
    M, N, K = 1024, 256, 256
    A = (torch.randn(M, N, generator=g) * torch.logspace(-3, 3, M)[:, None]).cuda()
    B = (1.0 + 1e-3 * torch.randn(M, K, generator=g)).cuda()
    A[100:164] = 0
    ref = A.double().t() @ B.double()
    assert _rel(F.wgrad_f32s(A, B)[0], ref) < 1e-5
    hi_only = A.bfloat16().float().double().t() @ B.bfloat16().float().double()
    assert _rel(hi_only, ref) > 1e-4                    # a plain bf16 product is NOT good enough here


@pytest.mark.parametrize("Bn,T", [(8, 16), (4, 64), (3, 96), (4, 40), (2, 256)])
@pytest.mark.parametrize("bm", [True, False])
def test_wgrad_lstm_operands(bm, Bn, T):
    """Two groups (directions), B = [x | h_{t-1}] / [x | h_{t+1}] read from `out` with a row shift: equals the explicitly
    shifted construction, in the batch-major (period T, shift 1) and time-major (shift B) layouts.  Sequence lengths below a
    chunk (16), whole chunks (64, 96, 256: the scalar-position path) and not a multiple of a chunk (40: chunks straddle sequences)."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(5)
    I, h = 128, 128
    TB = Bn * T
    dG = torch.randn(TB, 8 * h, generator=g).cuda()
    x = torch.randn(TB, I, generator=g).cuda()
    out = torch.randn(TB, 2 * h, generator=g).cuda()
    shift, period = (1, T) if bm else (Bn, 0)
    D = F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=shift, period=period)
    assert D.shape == (2, 4 * h, I + h)
    o3 = out.view(Bn, T, 2 * h) if bm else out.view(T, Bn, 2 * h)
    prev, nxt = torch.zeros_like(o3), torch.zeros_like(o3)
    if bm:
        prev[:, 1:] = o3[:, :-1]; nxt[:, :-1] = o3[:, 1:]
    else:
        prev[1:] = o3[:-1]; nxt[:-1] = o3[1:]
    for d, hs in ((0, prev.reshape(TB, 2 * h)[:, :h]), (1, nxt.reshape(TB, 2 * h)[:, h:])):
        ref = dG[:, d * 4 * h:(d + 1) * 4 * h].double().t() @ torch.cat([x, hs], 1).double()
        assert _rel(D[d], ref) < 1e-5, d


def test_wgrad_strided_operands_and_library_path():
    """Column slices of wider matrices go in without copies; the result agrees with split planes + library bf16 GEMM (the
    same arithmetic in another summation order) to fp32 rounding."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(7)
    M, N, K = 2048, 256, 256
    Aw = torch.randn(M, 3 * N, generator=g).cuda()
    Bw = torch.randn(M, 2 * K, generator=g).cuda()
    A, B = Aw[:, N:2 * N], Bw[:, K:]
    C = F.wgrad_f32s(A, B)[0]
    At = torch.empty(N, 3 * M, device="cuda", dtype=torch.bfloat16)
    Bt = torch.empty(K, 3 * M, device="cuda", dtype=torch.bfloat16)
    F.split_bf16x3_t(Aw, N, N, 0, False, At)
    F.split_bf16x3_t(Bw, K, K, 0, True, Bt)
    lib = torch.mm(At, Bt.t(), out_dtype=torch.float32)
    ref = A.double().t() @ B.double()
    assert _rel(C, ref) < 1e-5 and _rel(lib, ref) < 1e-5
    torch.testing.assert_close(C, lib, atol=2e-5 * float(ref.abs().max()), rtol=0)


def test_wgrad_argument_checks():
    from shufflingvideosfortsg_amd import functional as F
    A, B = torch.zeros(64, 256, device="cuda"), torch.zeros(64, 128, device="cuda")
    assert F.wgrad_f32s_ok(64, 256, 128) and not F.wgrad_f32s_ok(48, 256, 128) and not F.wgrad_f32s_ok(64, 128, 128)
    with pytest.raises(ValueError):
        F.wgrad_f32s(A[:48], B[:48])
    with pytest.raises(ValueError):
        F.wgrad_f32s(A[:, :128], B)
    with pytest.raises(RuntimeError):
        F.wgrad_f32s(A.cpu(), B.cpu())


def test_wgrad_alternate_tile_variant():
    """TSG_WGRAD_CFG=1 (128 x 128 tiles, two workgroups per CU) is read once per process: checked in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from shufflingvideosfortsg_amd import functional as F\n"
        "g = torch.Generator().manual_seed(2)\n"
        "for (M, N, K) in ((2560, 1024, 1024), (4128, 256, 384)):\n"
        "    A = torch.randn(M, N, generator=g).cuda(); B = torch.randn(M, K, generator=g).cuda()\n"
        "    C = F.wgrad_f32s(A, B)[0]; ref = A.double().t() @ B.double()\n"
        "    assert float((C.double() - ref).abs().max() / ref.abs().max()) < 1e-5\n"
        "    assert torch.equal(C, F.wgrad_f32s(A, B)[0])\n"
        "print('variant ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TSG_WGRAD_CFG="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("M,N,K", [(32, 256, 128), (2560, 1024, 1024), (16384, 1024, 1024), (8192, 512, 1024), (4128, 256, 384)])
def test_wgrad_bf16_operands(M, N, K):
    """tsg_wgrad_bf16: the same product on bf16 operands (the bf16 storage mode's weight gradients): fp32 result equal to the
    float64 product of the bf16 VALUES to fp32-accumulation level, deterministic."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(M + N + 1)
    A = torch.randn(M, N, generator=g).bfloat16().cuda()
    B = torch.randn(M, K, generator=g).bfloat16().cuda()
    C = F.wgrad_bf16(A, B)
    assert C.shape == (1, N, K) and C.dtype == torch.float32
    ref = A.double().t() @ B.double()
    assert _rel(C[0], ref) < 2e-6
    assert torch.equal(C, F.wgrad_bf16(A, B))


def test_wgrad_bf16_lstm_operands():
    """Two groups, shifted second segment read from a bf16 `out` (batch-major: shift 1, period T), strided column-slice operands:
    the one-launch LSTM weight gradient of the bf16 storage mode vs the explicitly shifted construction."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(6)
    Bn, T, I, h = 8, 16, 128, 128
    TB = Bn * T
    dG = torch.randn(TB, 8 * h, generator=g).bfloat16().cuda()
    x = torch.randn(TB, I, generator=g).bfloat16().cuda()
    out = torch.randn(Bn, T, 2 * h, generator=g).bfloat16().cuda()
    D = F.wgrad_bf16(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out.view(TB, 2 * h), K1=h, b1_group_stride=h, shift=1, period=T)
    assert D.shape == (2, 4 * h, I + h)
    hp = torch.zeros_like(out)
    hp[:, 1:, :h] = out[:, :-1, :h]
    hp[:, :-1, h:] = out[:, 1:, h:]
    hp = hp.view(TB, 2 * h).double()
    for d in range(2):
        gd = dG[:, d * 4 * h:(d + 1) * 4 * h].double()
        ref = torch.cat([gd.t() @ x.double(), gd.t() @ hp[:, d * h:(d + 1) * h]], 1)
        assert _rel(D[d], ref) < 2e-6, d


@pytest.mark.parametrize("Bn,T", [(8, 16), (4, 64), (3, 96), (4, 40), (2, 256), (16, 128)])
@pytest.mark.parametrize("bm", [True, False])
def test_wgrad_bf16_lstm_operands_layouts(bm, Bn, T):
    """The bf16 weight gradient of an LSTM layer in both sequence layouts: sequence lengths that are whole 32-row chunks (64, 96, 128,
    256 and every time-major case: the LDS-DMA + transposing-read kernel, shifted rows that do not exist served from the zero page)
    and lengths that are not (16, 40 batch-major: the register-staged kernel) -- vs the explicitly shifted float64 product of the
    bf16 values, and the two-output entry point bit-equal to slicing the single output."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(11)
    I, h = 128, 128
    TB = Bn * T
    dG = torch.randn(TB, 8 * h, generator=g).bfloat16().cuda()
    x = torch.randn(TB, I, generator=g).bfloat16().cuda()
    out = torch.randn(TB, 2 * h, generator=g).bfloat16().cuda()
    shift, period = (1, T) if bm else (Bn, 0)
    D = F.wgrad_bf16(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=shift, period=period)
    assert D.shape == (2, 4 * h, I + h) and D.dtype == torch.float32
    o3 = out.view(Bn, T, 2 * h) if bm else out.view(T, Bn, 2 * h)
    prev, nxt = torch.zeros_like(o3), torch.zeros_like(o3)
    if bm:
        prev[:, 1:] = o3[:, :-1]; nxt[:, :-1] = o3[:, 1:]
    else:
        prev[1:] = o3[:-1]; nxt[:-1] = o3[1:]
    for d, hs in ((0, prev.reshape(TB, 2 * h)[:, :h]), (1, nxt.reshape(TB, 2 * h)[:, h:])):
        ref = dG[:, d * 4 * h:(d + 1) * 4 * h].double().t() @ torch.cat([x, hs], 1).double()
        assert _rel(D[d], ref) < 2e-6, d
    C0, C1 = F.wgrad_bf16_out2(dG, x, out, N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h, shift=shift, period=period)
    assert torch.equal(C0, D[:, :, :I]) and torch.equal(C1, D[:, :, I:])
    assert torch.equal(D, F.wgrad_bf16(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=shift, period=period))


@pytest.mark.parametrize("env", [{"TSG_WGRAD_TR_CFG": "1"}, {"TSG_WGRAD_TR_CFG": "2"}, {"TSG_WGRAD_TR_CFG": "3"}, {"TSG_WGRAD_BF16_TR": "0"}])
def test_wgrad_bf16_kernel_variants(env):
    """The other ring shapes of the LDS-DMA kernel (sub-chunks x depth: <1,4>, <2,3>, <2,2>) and the register-staged kernel it replaced
    (TSG_WGRAD_BF16_TR=0) are read once per process: each checked in a child process on a plain product (ragged chunk count) and on an
    LSTM layer's shifted operands."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from shufflingvideosfortsg_amd import functional as F\n"
        "g = torch.Generator().manual_seed(3)\n"
        "rel = lambda c, r: float((c.double() - r).abs().max() / r.abs().max())\n"
        "for (M, N, K) in ((2560, 1024, 1024), (4128, 256, 384), (32, 256, 128)):\n"
        "    A = torch.randn(M, N, generator=g).bfloat16().cuda(); B = torch.randn(M, K, generator=g).bfloat16().cuda()\n"
        "    C = F.wgrad_bf16(A, B)[0]\n"
        "    assert rel(C, A.double().t() @ B.double()) < 2e-6 and torch.equal(C, F.wgrad_bf16(A, B)[0])\n"
        "Bn, T, I, h = 4, 64, 128, 128; TB = Bn * T\n"
        "dG = torch.randn(TB, 8 * h, generator=g).bfloat16().cuda(); x = torch.randn(TB, I, generator=g).bfloat16().cuda()\n"
        "out = torch.randn(Bn, T, 2 * h, generator=g).bfloat16().cuda()\n"
        "D = F.wgrad_bf16(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out.view(TB, 2 * h), K1=h, b1_group_stride=h, shift=1, period=T)\n"
        "hp = torch.zeros_like(out); hp[:, 1:, :h] = out[:, :-1, :h]; hp[:, :-1, h:] = out[:, 1:, h:]; hp = hp.view(TB, 2 * h).double()\n"
        "for d in range(2):\n"
        "    gd = dG[:, d * 4 * h:(d + 1) * 4 * h].double()\n"
        "    assert rel(D[d], torch.cat([gd.t() @ x.double(), gd.t() @ hp[:, d * h:(d + 1) * h]], 1)) < 2e-6\n"
        "print('variant ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("bm", [True, False])
def test_wgrad_two_outputs_equal_the_sliced_single_output(bm):
    """tsg_wgrad_f32s_out2 (round 4): the B0 / B1 column segments to two parameter-shaped outputs -- bit-equal to slicing the
    single-output result (same kernel, same summation order), with row ranges (partial tiles + reduce) and without."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(9)
    for Bn, T, I, h in ((8, 16, 128, 128), (64, 64, 256, 128)):        # 4 chunks: one range; 128 chunks: several ranges
        TB = Bn * T
        dG = torch.randn(TB, 8 * h, generator=g).cuda(); x = torch.randn(TB, I, generator=g).cuda(); out = torch.randn(TB, 2 * h, generator=g).cuda()
        shift, period = (1, T) if bm else (Bn, 0)
        D = F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=shift, period=period)
        C0, C1 = F.wgrad_f32s_out2(dG, x, out, N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h, shift=shift, period=period)
        assert C0.shape == (2, 4 * h, I) and C1.shape == (2, 4 * h, h)
        assert torch.equal(C0, D[:, :, :I]) and torch.equal(C1, D[:, :, I:])



# ---- stream-K (round 5): tile counts that do not fill the chip evenly are cut into equal (tile, chunk) pieces, one per CU; a tile with
# several contributors is finished by its last arriver in slot order (csrc/wgrad_split.hip: finish_tile) -------------------------------------
@pytest.mark.parametrize("M,N,K", [
    (2048, 2048, 2304),        # 144 tiles x 64 chunks on 256 workgroups: 36 chunks each, two or three contributors per tile
    (1056, 1536, 3072),        # 144 tiles x 33 chunks: ragged pieces (19 chunks, the last workgroups shorter / empty)
    (512, 2048, 2048),         # 128 tiles x 16 chunks: exactly two contributors per tile
    (16384, 2048, 1280),       # a long contraction: 80 tiles -> NOT stream-K (fewer than half the CUs): the split scheme still runs
])
def test_wgrad_stream_k_matches_float64_and_is_deterministic(M, N, K, request):
    from shufflingvideosfortsg_amd import _lib, functional as F
    _lib.load().tsg_wgrad_set_stream_k(1)               # also for fp32 operands (default: bf16 operands only)
    request.addfinalizer(lambda: _lib.load().tsg_wgrad_set_stream_k(-1))
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, N, generator=g).cuda()
    B = torch.randn(M, K, generator=g).cuda()
    C = F.wgrad_f32s(A, B)
    ref = A.double().t() @ B.double()
    assert _rel(C[0], ref) < 1e-5
    for _ in range(3):                                  # who arrives last differs from launch to launch; the sum order does not
        assert torch.equal(C, F.wgrad_f32s(A, B))
    Ab, Bb = A.bfloat16(), B.bfloat16()
    Cb = F.wgrad_bf16(Ab, Bb)
    assert _rel(Cb[0], Ab.double().t() @ Bb.double()) < 2e-6
    assert torch.equal(Cb, F.wgrad_bf16(Ab, Bb))


@pytest.mark.parametrize("bm", [True, False])
@pytest.mark.parametrize("Bn,T,I,h", [(8, 64, 512, 512), (128, 128, 1024, 512), (4, 40, 512, 512)])
def test_wgrad_stream_k_lstm_operands(bm, Bn, T, I, h, request):
    """The LSTM layer's [dW_ih | dW_hh] at stream-K tile counts (h = 512: 2 x 8 x (I + h) / 128 = 128 / 192 tiles), shifted second segment,
    both layouts, incl. the headline shape [2][2048 x 16384] x [16384 x 1536] and a period that is not a multiple of a chunk: vs the
    explicitly shifted float64 construction; the two-output form equals the sliced single output bit for bit; run-to-run identical."""
    from shufflingvideosfortsg_amd import _lib, functional as F
    _lib.load().tsg_wgrad_set_stream_k(1)
    request.addfinalizer(lambda: _lib.load().tsg_wgrad_set_stream_k(-1))
    g = torch.Generator().manual_seed(Bn + T)
    TB = Bn * T
    dG = torch.randn(TB, 8 * h, generator=g).cuda(); x = torch.randn(TB, I, generator=g).cuda(); out = torch.randn(TB, 2 * h, generator=g).cuda()
    shift, period = (1, T) if bm else (Bn, 0)
    D = F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=shift, period=period)
    o3 = out.view(Bn, T, 2 * h) if bm else out.view(T, Bn, 2 * h)
    prev, nxt = torch.zeros_like(o3), torch.zeros_like(o3)
    if bm:
        prev[:, 1:] = o3[:, :-1]; nxt[:, :-1] = o3[:, 1:]
    else:
        prev[1:] = o3[:-1]; nxt[:-1] = o3[1:]
    for d, hs in ((0, prev.reshape(TB, 2 * h)[:, :h]), (1, nxt.reshape(TB, 2 * h)[:, h:])):
        ref = dG[:, d * 4 * h:(d + 1) * 4 * h].double().t() @ torch.cat([x, hs], 1).double()
        assert _rel(D[d], ref) < 1e-5, d
    C0, C1 = F.wgrad_f32s_out2(dG, x, out, N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h, shift=shift, period=period)
    assert torch.equal(C0, D[:, :, :I]) and torch.equal(C1, D[:, :, I:])
    assert torch.equal(D, F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=shift, period=period))
    if T % 32 == 0:                                     # the bf16 LDS-DMA kernel wants whole chunks per sequence; same decomposition
        dGb, xb, ob = dG.bfloat16(), x.bfloat16(), out.bfloat16()
        Db = F.wgrad_bf16(dGb, xb, N=4 * h, groups=2, a_group_stride=4 * h, B1=ob, K1=h, b1_group_stride=h, shift=shift, period=period)
        o3b = ob.view(Bn, T, 2 * h) if bm else ob.view(T, Bn, 2 * h)
        pb, nb_ = torch.zeros_like(o3b), torch.zeros_like(o3b)
        if bm:
            pb[:, 1:] = o3b[:, :-1]; nb_[:, :-1] = o3b[:, 1:]
        else:
            pb[1:] = o3b[:-1]; nb_[:-1] = o3b[1:]
        for d, hs in ((0, pb.reshape(TB, 2 * h)[:, :h]), (1, nb_.reshape(TB, 2 * h)[:, h:])):
            ref = dGb[:, d * 4 * h:(d + 1) * 4 * h].double().t() @ torch.cat([xb, hs], 1).double()
            assert _rel(Db[d], ref) < 2e-6, d
        assert torch.equal(Db, F.wgrad_bf16(dGb, xb, N=4 * h, groups=2, a_group_stride=4 * h, B1=ob, K1=h, b1_group_stride=h, shift=shift, period=period))


def test_wgrad_stream_k_equals_the_split_scheme_to_rounding(tmp_path):
    """TSG_WGRAD_SK=1 / 0 (read once per process: stream-K always / never; the default is stream-K for bf16 operands only) give the same
    products in a different but equally fixed sum order: both within the float64 bound, and within rounding of each other."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from shufflingvideosfortsg_amd import functional as F\n"
        "g = torch.Generator().manual_seed(3)\n"
        "A = torch.randn(2048, 2048, generator=g).cuda(); B = torch.randn(2048, 2304, generator=g).cuda()\n"
        "C = F.wgrad_f32s(A, B)[0]; Cb = F.wgrad_bf16(A.bfloat16(), B.bfloat16())[0]\n"
        "assert torch.equal(C, F.wgrad_f32s(A, B)[0])\n"
        "torch.save((C.cpu(), Cb.cpu()), sys.argv[1])\n") % root
    outs = []
    for sk in ("1", "0"):
        f = str(tmp_path / f"c{sk}.pt")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, TSG_WGRAD_SK=sk), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(torch.load(f))
    g = torch.Generator().manual_seed(3)
    A = torch.randn(2048, 2048, generator=g); B = torch.randn(2048, 2304, generator=g)
    ref = A.double().t() @ B.double()
    for (C, Cb) in outs:
        assert _rel(C, ref) < 1e-5
    assert not torch.equal(outs[0][0], outs[1][0])          # (different decompositions really ran)
    assert _rel(outs[0][0], outs[1][0].double()) < 2e-6 and _rel(outs[0][1], outs[1][1].double()) < 2e-6
