"""tsg_dropout (csrc/dropout.hip): the dropout between the BiLSTM layers (reference networks/RNN.py:27-31, nn.LSTM(dropout=...))
without a stored mask -- keep fraction and scale, the backward regenerating the forward's mask, reproducibility under
torch.manual_seed, fresh masks per call and per graph replay, bf16 storage, ragged lengths, argument checks."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,p", [(1 << 20, 0.5), (1 << 20, 0.1), (4099, 0.3), (7, 0.5), (128 * 128 * 1024, 0.5)])
def test_keep_fraction_scale_and_backward_mask(dtype, n, p):
    from shufflingvideosfortsg_amd import functional as F
    torch.manual_seed(3)
    x = (torch.rand(n, device="cuda") + 0.5).to(dtype).requires_grad_(True)
    y = F.dropout(x, p, True)
    assert y.dtype == dtype and y.shape == x.shape
    kept = y != 0
    if n >= 4096:
        frac = float(kept.float().mean())
        assert abs(frac - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-3, frac
    ref = (x.detach().float() / (1 - p)).to(dtype)                      # kept elements: x / (1 - p), rounded once to the storage type
    torch.testing.assert_close(y.detach()[kept], ref[kept], rtol=1e-6 if dtype == torch.float32 else 8e-3, atol=0)
    g = (torch.rand(n, device="cuda") + 0.5).to(dtype)
    y.backward(g)
    # the backward regenerates the SAME mask: gradient non-zero exactly where the output is, = g / (1 - p) there
    assert torch.equal(x.grad != 0, kept)
    gref = (g.float() / (1 - p)).to(dtype)
    torch.testing.assert_close(x.grad[kept], gref[kept], rtol=1e-6 if dtype == torch.float32 else 8e-3, atol=0)


def test_reproducible_and_fresh_per_call():
    from shufflingvideosfortsg_amd import functional as F
    x = torch.ones(1 << 16, device="cuda")
    torch.manual_seed(11)
    a1, a2 = F.dropout(x, 0.5), F.dropout(x, 0.5)
    torch.manual_seed(11)
    b1, b2 = F.dropout(x, 0.5), F.dropout(x, 0.5)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)                  # same seed, same call sequence -> same masks
    assert not torch.equal(a1, a2)                                      # every call draws a fresh mask
    assert 0.2 < float(((a1 != 0) & (a2 != 0)).float().mean()) < 0.3    # two independent p = 0.5 masks overlap on a quarter
    assert F.dropout(x, 0.5, training=False) is x and F.dropout(x, 0.0) is x
    # neighbouring elements are not correlated: the keep bits of even and odd positions agree half of the time
    k = (a1 != 0).view(-1, 2)
    assert 0.45 < float((k[:, 0] == k[:, 1]).float().mean()) < 0.55


def test_graph_replays_draw_fresh_masks_and_backward_matches():
    """Under a HIP-graph capture the (seed, offset) pair is device-resident and its increment is part of the graph: every replay draws
    another mask, and the backward inside the same replay uses the keys its forward wrote."""
    from shufflingvideosfortsg_amd import functional as F
    dev = torch.device("cuda", 0)
    F.mha_graph_rng(dev)
    x = torch.ones(1 << 16, device=dev, requires_grad=True)
    gy = torch.ones(1 << 16, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            x.grad = None
            F.dropout(x, 0.5).backward(gy)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    x.grad = None
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = F.dropout(x, 0.5)
        y.backward(gy)
    outs = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(x.grad, y.detach())                          # x = 1, gy = 1: both are keep / (1 - p)
        outs.append(y.detach().clone())
        x.grad.zero_()
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    assert 0.45 < float((outs[2] != 0).float().mean()) < 0.55


def test_bilstm_inter_layer_dropout_uses_it():
    """The BiLSTM module in training mode: two calls differ (fresh masks), the same seed reproduces, p = 0 / eval are deterministic; the
    gradient flows through the dropped activations (finite, non-zero)."""
    from shufflingvideosfortsg_amd.model.networks.RNN import BiLSTM
    torch.manual_seed(0)
    m = BiLSTM(64, 128, 2, dropout=0.5).cuda().train()
    x = torch.randn(4, 16, 64, device="cuda", requires_grad=True)
    torch.manual_seed(5)
    o1 = m(x, states=False)[0]
    o2 = m(x, states=False)[0]
    torch.manual_seed(5)
    o3 = m(x, states=False)[0]
    assert torch.equal(o1, o3) and not torch.equal(o1, o2)
    o1.square().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    m.eval()
    assert torch.equal(m(x, states=False)[0], m(x, states=False)[0])


def test_argument_checks():
    from shufflingvideosfortsg_amd import _lib
    lib = _lib.load()
    x = torch.ones(64, device="cuda"); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tsg_dropout(_lib.ptr(x), _lib.ptr(y), 64, 1.0, 0, 0, None, None, 0, 0, st) != 0        # p must be < 1
    assert lib.tsg_dropout(_lib.ptr(x), _lib.ptr(y), 0, 0.5, 0, 0, None, None, 0, 0, st) != 0
    assert lib.tsg_dropout(_lib.ptr(x), _lib.ptr(y), 64, 0.5, 0, 0, None, None, 1, 0, st) != 0        # key_mode 1 without rng_dev
    assert lib.tsg_dropout(None, _lib.ptr(y), 64, 0.5, 0, 0, None, None, 0, 0, st) != 0
    assert lib.tsg_dropout(_lib.ptr(x), _lib.ptr(y), 64, 0.5, 1, 2, None, None, 0, 0, st) == 0
    torch.cuda.synchronize()
    assert set(y.unique().tolist()) <= {0.0, 2.0}
