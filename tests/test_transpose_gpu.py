"""tsg_transpose_f32 (the weight operand of the input-gradient GEMMs, W_hh^T of the LSTM backward): bit-equal to
``.transpose(-1, -2).contiguous()`` -- whole matrices, ragged tile edges, a batch, a column slice read in place."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1024, 4096), (4096, 1024), (2, 512, 2048), (2, 2048, 300), (68, 4), (4, 132), (3, 100, 260)])
def test_transposed_equals_torch(shape):
    from shufflingvideosfortsg_amd import functional as TF
    w = torch.randn(*shape, device="cuda")
    out = TF.transposed(w)
    assert out.is_contiguous() and torch.equal(out, w.transpose(-1, -2).contiguous())


def test_transposed_column_slice_in_place():
    from shufflingvideosfortsg_amd import functional as TF
    W = torch.randn(256, 2048, device="cuda")
    for sl in (W[:, :1024], W[:, 1024:], W[:, 512:516]):
        out = TF.transposed(sl)
        assert torch.equal(out, sl.t().contiguous())
    # shapes the kernel does not take fall back to torch
    odd = torch.randn(10, 7, device="cuda")
    assert torch.equal(TF.transposed(odd), odd.t().contiguous())


def test_transpose_rejects_bad_arguments():
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr
    lib = _lib.load()
    a = torch.randn(8, 8, device="cuda"); b = torch.empty(8, 8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tsg_transpose_f32(ptr(a), 8, ptr(b), 1, 8, 6, st) != 0
    assert lib.tsg_transpose_f32(ptr(a), 4, ptr(b), 1, 8, 8, st) != 0          # ld < cols
    assert lib.tsg_transpose_f32(None, 8, ptr(b), 1, 8, 8, st) != 0
