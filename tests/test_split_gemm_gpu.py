"""GPU tests of the split-precision GEMM operand kernel (tsg_split_bf16x3) through the C ABI."""
import pytest
import torch

from shufflingvideosfortsg_amd import functional as TF

pytestmark = pytest.mark.gpu


def test_split_bf16x3_and_split_gemm():
    """tsg_split_bf16x3: planes are exactly (hi,hi,lo)/(hi,lo,hi) with hi = rne_bf16(x), lo = rne_bf16(x-hi), in both
    layouts; the split GEMM's error is at the fp32 GEMM's level (vs float64)."""
    torch.manual_seed(3)
    x = torch.randn(64, 40, device="cuda") * torch.logspace(-6, 6, 40, device="cuda")
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    L = TF.split_bf16x3(x, 1, False); R = TF.split_bf16x3(x, 1, True)
    assert torch.equal(L, torch.cat([hi, hi, lo], 1)) and torch.equal(R, torch.cat([hi, lo, hi], 1))
    L0 = TF.split_bf16x3(x, 0, False); R0 = TF.split_bf16x3(x, 0, True)
    assert torch.equal(L0, torch.cat([hi, hi, lo], 0)) and torch.equal(R0, torch.cat([hi, lo, hi], 0))
    a = torch.randn(512, 1024, device="cuda"); b = torch.randn(1024, 256, device="cuda") * 0.05
    ref = a.double() @ b.double()
    TF.set_gemm_dtype("f32s")
    try:
        for A, B_ in [(a, b), (a.t().contiguous().t(), b), (a, b.t().contiguous().t())]:
            got = TF._mm(A, B_)
            assert got.dtype == torch.float32
            err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 1e-5, err
    finally:
        TF.set_gemm_dtype(None)


def test_split_bf16x3_shifted_rows():
    """tsg_split_bf16x3_shift: a column slice of a wider matrix, rows shifted by +-s with zero rows shifted in, equals the
    split of the materialised shifted copy (the h_{t-1} operand of the LSTM weight-gradient GEMM)."""
    torch.manual_seed(5)
    R, C, s = 48, 64, 8
    x = torch.randn(R, C, device="cuda")
    for col0, cols, shift in [(0, 32, s), (32, 32, -s), (16, 24, 0), (0, 64, R)]:
        sl = x[:, col0:col0 + cols]
        if shift > 0:
            ref = torch.cat([torch.zeros(min(shift, R), cols, device="cuda"), sl[:max(R - shift, 0)]], 0)
        elif shift < 0:
            ref = torch.cat([sl[-shift:], torch.zeros(-shift, cols, device="cuda")], 0)
        else:
            ref = sl.contiguous()
        got = TF.split_bf16x3_rows_shifted(x, col0, cols, shift, True)
        assert torch.equal(got, TF.split_bf16x3(ref, 0, True)), (col0, cols, shift)


def test_lstm_fwd_bias_entry():
    """tsg_lstm_fwd_bias(Gx, bias) == tsg_lstm_fwd(Gx + bias), persistent and launch-per-step kernels."""
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    for (B, T, h), use_ws in [((20, 12, 128), True), ((5, 6, 36), False)]:
        g = torch.Generator().manual_seed(h)
        Gx = (torch.randn(T, B, 2, 4 * h, generator=g) * 0.5).cuda(); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
        bias = torch.randn(2, 4 * h, generator=g).cuda()
        res = []
        for fused in (False, True):
            ws = torch.zeros(512, dtype=torch.int32, device="cuda") if use_ws else None
            out = torch.empty(T, B, 2 * h, device="cuda"); R = torch.empty(T, 2, B, h, 4, device="cuda"); Cs = torch.empty(T, 2, B, h, device="cuda")
            gin = Gx if fused else Gx + bias.view(1, 1, 2, 4 * h)
            rc = lib.tsg_lstm_fwd_bias(ptr(gin), ptr(bias) if fused else None, ptr(W), ptr(out), ptr(R), ptr(Cs),
                                       ptr(ws) if ws is not None else None, B, T, h, TSG_F32, 0, st)
            torch.cuda.synchronize()
            assert rc == 0 and (ws is None or int(ws[0]) == 0)
            res.append((out, R, Cs))
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_split_bf16x3_transposed():
    """tsg_split_bf16x3_t writes the same planes as the row-stacked split, transposed (contraction index contiguous), for
    column slices, row shifts, ragged tiles and side-by-side operands in one buffer."""
    torch.manual_seed(6)
    for R, C in [(48, 64), (160, 200), (64, 8)]:
        x = torch.randn(R, C, device="cuda") * 3
        W = C + 8
        buf = torch.full((W, 3 * R), 7.0, device="cuda", dtype=torch.bfloat16)
        for col0, cols, shift, right, row0 in [(0, C, 0, False, 0), (0, C // 2, 16, True, 4), (C // 2, C // 2, -16, True, 8)]:
            TF.split_bf16x3_t(x, col0, cols, shift, right, buf, row0)
            ref = TF.split_bf16x3_rows_shifted(x, col0, cols, shift, right)            # [3R, cols]
            assert torch.equal(buf[row0:row0 + cols], ref.t()), (R, C, col0, cols, shift)
            assert (buf[row0 + cols:] == 7.0).all() and (buf[:row0] == 7.0).all()
            buf.fill_(7.0)
        TF.split_bf16x3_t(x, 0, C // 2, 0, True, buf, 0, dup_row0=C // 2 + 4)          # duplicate destination
        ref = TF.split_bf16x3_rows_shifted(x, 0, C // 2, 0, True).t()
        assert torch.equal(buf[:C // 2], ref) and torch.equal(buf[C // 2 + 4:C + 4], ref) and (buf[C // 2:C // 2 + 4] == 7.0).all()


def test_split_shift_within_sequences():
    """period > 0: rows are consecutive sequences of `period` steps (batch-major [B*T, C]); a shift never crosses into the
    neighbouring sequence -- both split kernels, against the shifted copy built per sequence."""
    torch.manual_seed(8)
    Bn, T, C = 6, 8, 32
    x = torch.randn(Bn * T, C, device="cuda")
    x3 = x.view(Bn, T, C)
    z = torch.zeros(Bn, 1, C, device="cuda")
    for shift, ref3 in [(1, torch.cat([z, x3[:, :-1]], 1)), (-1, torch.cat([x3[:, 1:], z], 1))]:
        ref = TF.split_bf16x3(ref3.reshape(Bn * T, C).contiguous(), 0, True)
        got = TF.split_bf16x3_rows_shifted(x, 0, C, shift, True, period=T)
        assert torch.equal(got, ref)
        buf = torch.empty(C, 3 * Bn * T, device="cuda", dtype=torch.bfloat16)
        TF.split_bf16x3_t(x, 0, C, shift, True, buf, 0, period=T)
        assert torch.equal(buf, ref.t())


def test_linear_split_precision_matches_fp32():
    """functional.linear in the f32s mode (forward and input gradient as split-precision GEMMs, weight gradient fp32) vs
    torch.nn.functional.linear in float64: error at the fp32 GEMM's level; small inputs fall through to F.linear."""
    import torch.nn.functional as F
    torch.manual_seed(11)
    x = torch.randn(64, 64, 256, device="cuda", requires_grad=True)          # 4096 rows
    w = (torch.randn(128, 256, device="cuda") * 0.06).requires_grad_(True)
    b = torch.randn(128, device="cuda", requires_grad=True)
    g = torch.randn(64, 64, 128, device="cuda")
    ref = F.linear(x.double(), w.double(), b.double())
    gx, gw, gb = torch.autograd.grad(ref, (x, w, b), g.double())
    TF.set_gemm_dtype("f32s")
    try:
        y = TF.linear(x, w, b)
        assert y.grad_fn is not None and "LinearSplit" in type(y.grad_fn).__name__
        dx, dw, db = torch.autograd.grad(y, (x, w, b), g)
        small = TF.linear(x[:1, :8], w, b)                                  # 8 rows: plain F.linear
        assert "LinearSplit" not in type(small.grad_fn).__name__
    finally:
        TF.set_gemm_dtype(None)
    for got, want in [(y, ref), (dx, gx), (dw, gw), (db, gb)]:
        err = (got.double() - want).abs().max().item() / want.abs().max().item()
        assert err < 2e-5, err


def test_linear_split_large_weight_gradient_vs_float64():
    """functional.linear in the split-precision mode at the self-attention head's projection shape ([8192, 2048] x [2048, 2048]):
    forward, input gradient AND the weight gradient (a split-precision GEMM over the 8192-row contraction from the transposing
    split) stay at the fp32 GEMM's error level vs float64."""
    torch.manual_seed(9)
    M, K, N = 8192, 2048, 2048
    x = (torch.randn(M, K, device="cuda") * 0.5).requires_grad_(True); w = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_(True)
    gy = torch.randn(M, N, device="cuda") * 0.1
    TF.set_gemm_dtype("f32s")
    try:
        y = TF.linear(x, w)
        y.backward(gy)
    finally:
        TF.set_gemm_dtype(None)
    xd, wd, gd = x.detach().double(), w.detach().double(), gy.double()
    for got, ref, name in ((y.detach(), xd @ wd.t(), "y"), (x.grad, gd @ wd, "dx"), (w.grad, gd.t() @ xd, "dw")):
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, (name, err)
