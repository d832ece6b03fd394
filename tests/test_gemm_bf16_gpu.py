"""tsg_gemm_bf16 (csrc/gemm_bf16.hip; round 5): the dense projections of the bf16 STORAGE mode on the hand-written kernel -- operands global -> LDS
by DMA, read as MFMA fragments, fp32 accumulation.  Reference sites: W_s / W_a (networks/attention.py:104-113), sent_linear
(components/VideoEncoder.py:59), the heads' first Linear (components/SpanPredictor.py:71-85, components/DistributionAlign.py:83-118), nn.LSTM's
input projection / input gradient (networks/RNN.py:31,42).  Checked against a float64 product of the same bf16-valued operands: the fp32 output
to fp32-accumulation error, the bf16 output to one rounding (2^-9 relative); strided operands; both row tiles; `functional.linear` and the bf16
BiLSTM layer in the storage mode against the oracle's formulas with the library path (TSG_OWN_GEMM_BF16=0) as the A/B."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops(M, N, K, seed, ldx=None, ldw=None):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(M, ldx or K, generator=g).to(BF).cuda()
    W = (torch.randn(N, ldw or K, generator=g) / K ** 0.5).to(BF).cuda()
    b = torch.randn(N, generator=g).cuda()
    return X, W, b


@pytest.mark.parametrize("M,N,K", [
    (128, 256, 32),             # one 128-row tile, one chunk (ring shorter than its depth)
    (256, 256, 64),             # one 256-row tile, two chunks
    (384, 512, 96),             # 128-row tiles (M % 256 != 0), three chunks = the ring's look-ahead exactly
    (2560, 1024, 1024),         # word-side projections
    (16384, 1024, 1024),        # W_a / sent_linear at the north-star shape
    (16384, 4096, 1024),        # an LSTM layer's input projection
    (16384, 1024, 4096),        # ... and its input gradient (long contraction)
    (1280, 2048, 320),          # sentence encoder l0 (K = 300 padded is not taken: see the predicate)
])
def test_gemm_bf16_matches_float64(M, N, K):
    from shufflingvideosfortsg_amd import functional as F
    X, W, b = _ops(M, N, K, M + N + K)
    ref = X.double() @ W.double().t()
    y32 = F.gemm_bf16(X, W, out_dtype=torch.float32)
    scale = float(ref.abs().max())
    assert float((y32.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, K / 1024) ** 0.5 + 1e-6
    yb = F.gemm_bf16(X, W, b)
    assert yb.dtype == BF
    refb = ref + b.double()
    # one bf16 rounding of the fp32 result: 2^-9 relative per element (+ the fp32 accumulation error)
    torch.testing.assert_close(yb.double(), refb, atol=4e-3 * float(refb.abs().max()) * 2 ** -1 + 1e-3, rtol=2 ** -8)
    assert torch.equal(yb, F.gemm_bf16(X, W, b))
    assert torch.equal(yb, (F.gemm_bf16(X, W, b, out_dtype=torch.float32)).to(BF))      # the bf16 output IS the rounded fp32 output


def test_gemm_bf16_strided_operands():
    """Column slices of wider row-major matrices go in without copies (the video half W1[:, :Dv] of a head's first Linear; a slice of x)."""
    from shufflingvideosfortsg_amd import functional as F
    X, W, b = _ops(512, 512, 256, 7, ldx=384, ldw=2048 + 256)
    xs, ws = X[:, 64:64 + 256], W[:, 2048:2048 + 256]
    y = F.gemm_bf16(xs, ws, b, out_dtype=torch.float32)
    ref = xs.double() @ ws.double().t() + b.double()
    assert float((y.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())


def test_gemm_bf16_argument_checks():
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr, TSG_BF16
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    X, W, _ = _ops(256, 256, 64, 1)
    Y = torch.empty(256, 256, device="cuda", dtype=BF)
    assert lib.tsg_gemm_bf16(ptr(X), 64, ptr(W), 64, None, ptr(Y), 256, 256, 256, 64, TSG_BF16, st) == 0
    assert lib.tsg_gemm_bf16(ptr(X), 64, ptr(W), 64, None, ptr(Y), 256, 200, 256, 64, TSG_BF16, st) != 0      # M % 128
    assert lib.tsg_gemm_bf16(ptr(X), 64, ptr(W), 64, None, ptr(Y), 256, 256, 128, 64, TSG_BF16, st) != 0      # N % 256
    assert lib.tsg_gemm_bf16(ptr(X), 64, ptr(W), 64, None, ptr(Y), 256, 256, 256, 48, TSG_BF16, st) != 0      # K % 32
    assert lib.tsg_gemm_bf16(ptr(X), 60, ptr(W), 64, None, ptr(Y), 256, 256, 256, 32, TSG_BF16, st) != 0      # ldx % 8
    assert lib.tsg_gemm_bf16(ptr(X), 64, ptr(W), 64, None, ptr(Y), 256, 256, 256, 64, 2, st) != 0             # out dtype
    assert lib.tsg_gemm_bf16(None, 64, ptr(W), 64, None, ptr(Y), 256, 256, 256, 64, TSG_BF16, st) != 0
    torch.cuda.synchronize()


def test_linear_bf16_storage_runs_on_the_own_gemm_and_matches_the_formula(request):
    """functional.linear in the bf16 storage mode (the path's nn.Linear projections): forward / input gradient on tsg_gemm_bf16, weight gradient on
    tsg_wgrad_bf16 -- y = x W^T + b, dx = dy W, dW = dy^T x, db = sum dy on the bf16-valued operands in float64."""
    from shufflingvideosfortsg_amd import engine, functional as F
    engine.set_precision("bf16")
    request.addfinalizer(lambda: engine.set_precision(None))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 128, 1024, generator=g).to(BF).cuda().requires_grad_(True)
    w = (torch.randn(1024, 1024, generator=g) / 32).cuda().requires_grad_(True)
    b = torch.randn(1024, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(64, 128, 1024, generator=g).to(BF).cuda()
    F.kernel_timer.enable(only=None)
    y = F.linear(x, w, b)
    y.backward(dy)
    torch.cuda.synchronize()
    names = [r[0] for r in F.kernel_timer.records]
    F.kernel_timer.disable(); F.kernel_timer.records.clear()
    assert names.count("tsg_gemm_bf16") == 2 and "tsg_wgrad_bf16" in names, names
    wb = w.detach().to(BF).double(); xd = x.detach().double().view(-1, 1024); dyd = dy.double().view(-1, 1024)
    ref = xd @ wb.t() + b.detach().double()
    torch.testing.assert_close(y.detach().double().view(-1, 1024), ref, atol=2 ** -8 * float(ref.abs().max()), rtol=2 ** -8)
    refdx = dyd @ wb
    torch.testing.assert_close(x.grad.double().view(-1, 1024), refdx, atol=2 ** -8 * float(refdx.abs().max()), rtol=2 ** -8)
    refdw = dyd.t() @ xd
    assert w.grad.dtype == torch.float32
    assert float((w.grad.double() - refdw).abs().max()) < 2e-6 * float(refdw.abs().max())
    torch.testing.assert_close(b.grad.double(), dyd.sum(0), atol=1e-3 * float(dyd.sum(0).abs().max()), rtol=1e-3)
