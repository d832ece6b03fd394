"""K3 parity on the GPU: tsg_boundary_score_{fwd,bwd} vs the CPU oracle's concat + MLP_predictor."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


def _weights(Dv, Ds, Hm, g):
    p = {}
    for n in ("start", "end"):
        p[f"{n}_mlp_1.weight"] = torch.randn(Hm, Dv + Ds, generator=g) / (Dv + Ds) ** 0.5
        p[f"{n}_mlp_1.bias"] = torch.randn(Hm, generator=g) * 0.1
        p[f"{n}_mlp_2.weight"] = torch.randn(1, Hm, generator=g) / Hm ** 0.5
        p[f"{n}_mlp_2.bias"] = torch.randn(1, generator=g) * 0.1
    return p


def _stack(p, Dv):
    """Reference parameters -> the stacked / split operands of the kernel."""
    W1 = torch.cat([p["start_mlp_1.weight"], p["end_mlp_1.weight"]], 0)        # [2Hm, Dv+Ds]
    return (W1[:, :Dv], W1[:, Dv:], torch.cat([p["start_mlp_1.bias"], p["end_mlp_1.bias"]]),
            torch.cat([p["start_mlp_2.weight"].reshape(-1), p["end_mlp_2.weight"].reshape(-1)]),
            torch.cat([p["start_mlp_2.bias"], p["end_mlp_2.bias"]]))


@pytest.mark.parametrize("B,T,Dv,Ds,Hm,use_mask,use_gate", [
    (3, 11, 24, 16, 16, False, False),
    (3, 11, 24, 16, 16, True, False),
    (2, 32, 512, 512, 256, False, True),      # config 0 shape, GMD gate
    (2, 128, 1024, 1024, 256, True, True),    # north-star shape (B reduced)
    (1, 1, 8, 8, 2, False, False),
    (2, 300, 64, 32, 130, True, True),        # ragged: T not a multiple of anything, 2Hm = 260
    (2, 600, 64, 32, 256, True, True),        # long sequence: several row batches per wave in the backward
])
def test_boundary_parity(B, T, Dv, Ds, Hm, use_mask, use_gate):
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(5)
    p = {k: v.requires_grad_(True) for k, v in _weights(Dv, Ds, Hm, g).items()}
    video = torch.randn(B, T, Dv, generator=g, requires_grad=True)
    sent = torch.randn(B, Ds, generator=g, requires_grad=True)
    gate = (torch.randn(B, T, generator=g)).requires_grad_(True) if use_gate else None
    mask = None
    if use_mask:
        n = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
        mask = (torch.arange(T)[None, :] < n[:, None]).int()
    gs, ge = torch.randn(B, T, generator=g), torch.randn(B, T, generator=g)

    x = O.video_sentence_concat(video, sent)
    if use_gate:
        x = gate.unsqueeze(2) * x
    s0, e0 = O.mlp_predictor(x, p, mask)
    (s0 * gs + e0 * ge).sum().backward()
    leaves = [video, sent] + ([gate] if use_gate else []) + list(p.values())
    ref = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None

    dev = {k: v.detach().cuda().requires_grad_(True) for k, v in p.items()}
    vd, sd = video.detach().cuda().requires_grad_(True), sent.detach().cuda().requires_grad_(True)
    gd = gate.detach().cuda().requires_grad_(True) if use_gate else None
    W1v, W1s, b1, w2, b2 = _stack(dev, Dv)
    y = torch.nn.functional.linear(vd, W1v)
    cs = torch.nn.functional.linear(sd, W1s)
    s1, e1 = F.boundary_score(y, cs, b1, w2, b2, gd, mask.cuda() if mask is not None else None)
    (s1 * gs.cuda() + e1 * ge.cuda()).sum().backward()
    torch.cuda.synchronize()
    torch.testing.assert_close(s1.detach().cpu(), s0.detach(), **TOL)
    torch.testing.assert_close(e1.detach().cpu(), e0.detach(), **TOL)
    got = [vd, sd] + ([gd] if use_gate else []) + list(dev.values())
    names = ["video", "sent"] + (["gate"] if use_gate else []) + list(p.keys())
    for gt, want, name in zip(got, ref, names):
        torch.testing.assert_close(gt.grad.cpu(), want, atol=2e-4, rtol=2e-3, msg=lambda m, n=name: f"d{n}: {m}")


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_boundary_golden(golden, tag):
    from shufflingvideosfortsg_amd import functional as F
    g = golden("mlp_" + tag)
    w = {k: v.cuda() for k, v in g.weights.items()}
    x = g.t("x").cuda()
    Dv = 24                                           # any split works: the kernel only sees y + cs
    W1v, W1s, b1, w2, b2 = _stack(w, Dv)
    y = torch.nn.functional.linear(x[..., :Dv], W1v) + torch.nn.functional.linear(x[..., Dv:], W1s)
    cs = torch.zeros(x.shape[0], y.shape[-1], device="cuda")
    s, e = F.boundary_score(y, cs, b1, w2, b2, None, g.t("mask").cuda() if tag == "mask" else None)
    torch.testing.assert_close(s.cpu(), g.t("start"), **TOL)
    torch.testing.assert_close(e.cpu(), g.t("end"), **TOL)


@pytest.mark.parametrize("B,T,Hm,use_mask,use_gate", [(64, 128, 256, True, True), (5, 77, 34, False, True), (3, 300, 130, True, False)])
def test_boundary_bwd_one_launch_matches_two_kernel_entry(B, T, Hm, use_mask, use_gate):
    """tsg_boundary_score_bwd_ws (one launch, ticket counters in a re-used workspace, fixed-order T-sums) against the two-kernel
    entry point tsg_boundary_score_bwd on the same operands; three calls on ONE workspace zeroed once (the counters must come back
    to zero), bit-identical results across the calls."""
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    J = 2 * Hm
    dev = "cuda"
    y = torch.randn(B, T, J, generator=g).to(dev); cs = torch.randn(B, J, generator=g).to(dev)
    b1 = (torch.randn(J, generator=g) * 0.1).to(dev); w2 = (torch.randn(J, generator=g) / J ** 0.5).to(dev)
    gate = torch.randn(B, T, generator=g).to(dev) if use_gate else None
    mask = None
    if use_mask:
        n = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
        mask = (torch.arange(T)[None, :] < n[:, None]).int().to(dev)
    ps = torch.softmax(torch.randn(B, T, generator=g), -1).to(dev); pe = torch.softmax(torch.randn(B, T, generator=g), -1).to(dev)
    dps = torch.randn(B, T, generator=g).to(dev); dpe = torch.randn(B, T, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream

    def outputs():
        return (torch.empty_like(y), torch.empty_like(cs), torch.empty(B, J, device=dev), torch.empty(B, J, device=dev),
                torch.empty(B, 2, device=dev), torch.empty(B, T, device=dev) if use_gate else None)

    def p_(t):
        return ptr(t) if t is not None else None

    o2 = outputs()
    dl = torch.empty(B, T, 2, device=dev)
    assert lib.tsg_boundary_score_bwd(ptr(y), ptr(cs), ptr(b1), ptr(w2), p_(gate), p_(mask), ptr(ps), ptr(pe), ptr(dps), ptr(dpe),
                                      ptr(o2[0]), ptr(o2[1]), ptr(o2[2]), ptr(o2[3]), ptr(o2[4]), p_(o2[5]), ptr(dl), B, T, Hm, TSG_F32, st) == 0
    nb = int(lib.tsg_boundary_score_bwd_ws_bytes(B, T, Hm))
    ws = torch.zeros(nb, device=dev, dtype=torch.uint8)
    runs = []
    for _ in range(3):
        o1 = outputs()
        assert lib.tsg_boundary_score_bwd_ws(ptr(y), ptr(cs), ptr(b1), ptr(w2), p_(gate), p_(mask), ptr(ps), ptr(pe), ptr(dps), ptr(dpe),
                                             ptr(o1[0]), ptr(o1[1]), ptr(o1[2]), ptr(o1[3]), ptr(o1[4]), p_(o1[5]), ptr(ws), nb,
                                             B, T, Hm, TSG_F32, st) == 0, lib.tsg_last_error()
        runs.append(o1)
    torch.cuda.synchronize()
    assert int(ws[:4 * B].view(torch.int32).abs().sum()) == 0              # ticket counters back to zero
    names = ("dy", "dcs", "db1", "dw2", "db2", "dgate")
    for name, a, b in zip(names, runs[0], o2):
        if a is None:
            continue
        if name in ("db1", "dw2", "db2"):
            a, b = a.sum(0), b.sum(0)
        torch.testing.assert_close(a, b, atol=1e-5, rtol=1e-4, msg=lambda m, n=name: f"{n}: {m}")
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            if a is not None:
                assert torch.equal(a, b)
    assert lib.tsg_boundary_score_bwd_ws(ptr(y), ptr(cs), ptr(b1), ptr(w2), None, None, ptr(ps), ptr(pe), ptr(dps), ptr(dpe),
                                         ptr(o2[0]), ptr(o2[1]), ptr(o2[2]), ptr(o2[3]), ptr(o2[4]), None, ptr(ws), nb - 16,
                                         B, T, Hm, TSG_F32, st) == -2      # TSG_E_SHAPE: workspace too small
