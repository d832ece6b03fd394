"""tsg_gemm_f32s (split-precision GEMM, operands converted on load) on the GPU: against the fp64 product and against the
operand-planes + library path it replaces; the Linear that uses it, forward and backward, against torch's fp32 Linear."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,bias", [(256, 256, 32, False), (512, 768, 96, True), (2048, 1024, 1024, True), (16384, 512, 2048, False),
                                        (64, 256, 64, True), (2560, 1024, 1024, True), (1280, 4096, 1024, False), (8192, 1024, 1024, False), (192, 512, 96, True)])
def test_gemm_f32s_matches_fp64_and_the_planes_path(M, N, K, bias):
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    y = F.gemm_f32s(x, w, b)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-5 * scale                  # hi*hi + hi*lo + lo*hi: 2^-16 relative per term
    F.set_gemm_dtype("f32s")
    try:
        planes = torch.mm(F.split_bf16x3(x, 1, False), F.split_bf16x3(w, 1, True).t(), out_dtype=torch.float32) + (b if bias else 0.0)
    finally:
        F.set_gemm_dtype(None)
    torch.testing.assert_close(y, planes, atol=2e-5 * scale, rtol=0)               # same arithmetic, different summation order
    assert torch.equal(y, F.gemm_f32s(x, w, b))                                    # run-to-run identical


def test_gemm_f32s_rejects_ragged_shapes():
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(256, 64, device="cuda"); w = torch.zeros(256, 64, device="cuda"); y = torch.zeros(256, 256, device="cuda")
    assert lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), 255, 256, 64, st) == -2
    assert lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), 96, 256, 64, st) == -2             # rows: whole 64-row tiles
    assert lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), 256, 128, 64, st) == -2
    assert lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), 256, 256, 48, st) == -2
    assert lib.tsg_gemm_f32s(None, ptr(w), None, ptr(y), 256, 256, 64, st) == -1


def test_linear_f32s_uses_the_kernel_and_matches_fp32(request):
    """functional.linear in the "f32s" mode at a shape the kernel takes (25 600 rows: 100 tiles): forward and all three gradients
    against torch.nn.functional.linear in fp32 at the fp32 tolerance of the mode."""
    from shufflingvideosfortsg_amd import engine, functional as F
    engine.set_precision("f32s")
    request.addfinalizer(lambda: engine.set_precision(None))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100, 256, 512, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(256, 512, generator=g) / 512 ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(256, generator=g).cuda().requires_grad_(True)
    assert F.gemm_f32s_ok(100 * 256, 256, 512)
    gy = torch.randn(100, 256, 256, generator=g).cuda()
    y = F.linear(x, w, b); y.backward(gy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    y0 = torch.nn.functional.linear(x, w, b); y0.backward(gy)
    for a, r, name in zip(got, (y0.detach(), x.grad, w.grad, b.grad), ("y", "dx", "dw", "db")):
        torch.testing.assert_close(a, r, atol=3e-4 * max(1.0, float(r.abs().max())), rtol=2e-3, msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("M,N,K,seg", [(512, 256, 96, None), (256, 512, 1024, None), (512, 256, 512, 256), (256, 1024, 4096, None)])
def test_gemm_f32s_nn(M, N, K, seg):
    """tsg_gemm_f32s_nn (round 4): y = x @ w with w [K,N] row-major (the contraction-major right operand: dX = dY W with the weight as
    the parameter stores it), also as a column slice of a wider matrix and as two row segments -- bit-equal to tsg_gemm_f32s on the
    explicitly transposed weight (the same products in the same order), and close to float64."""
    from shufflingvideosfortsg_amd import functional as TF
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    Ww = (torch.randn(K, N + 64, generator=g) / K ** 0.5).cuda()
    w = Ww[:, 32:32 + N]                                   # a column slice: row stride N + 64 (offset 32 floats = 128 B: 16-byte aligned)
    want = TF.gemm_f32s(x, w.t().contiguous())
    if seg is None:
        got = TF.gemm_f32s_nn(x, w)
    else:
        W2 = Ww.clone()                                    # two row segments with the same row stride
        got = TF.gemm_f32s_nn(x, Ww[:seg, 32:32 + N], W2[seg:, 32:32 + N])
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = x.double() @ w.double()
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1e-5
