"""functional.shared_grad / GradSink (round 5): the gradient of an activation with several consumers summed inside the consumers' own kernels
(first arrival adopted, later ones added in their GEMM's epilogue -- tsg_gemm_f32s_nn_acc -- or, in the bf16 storage mode, by a beta = 1 GEMM),
against autograd's own accumulation of the same graph.  Every ORDER of arrival is exercised: the full-size consumers first, the row-block consumer
(boundary head on the leading rows: reference SpanGroundMatchDisc.py:84) first -- which zero-fills the sink -- a consumer that knows nothing about
sinks beside them, and a consumer that never runs its backward."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu

B2, T, D, H, HM = 8, 64, 256, 256, 256                    # 2B rows of clip features; matching head width; boundary head width per branch


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    return dict(x=r(B2, T, D), q=r(B2, D), W1=r(H, 2 * D, sc=D ** -0.5), b1=r(H, sc=0.1), w2=r(1, H, sc=H ** -0.5), b2=r(1, sc=0.1),
                sent=r(B2 // 2, D), Ws=r(HM, 2 * D, sc=D ** -0.5), bs=r(HM, sc=0.1), We=r(HM, 2 * D, sc=D ** -0.5), be=r(HM, sc=0.1),
                w2s=r(1, HM, sc=HM ** -0.5), b2s=r(1, sc=0.1), w2e=r(1, HM, sc=HM ** -0.5), b2e=r(1, sc=0.1),
                m=[(torch.rand(B2, T, generator=g) > 0.5).float().cuda() for _ in range(3)])


def _loss(TF, P, x, order, use):
    """The consumers in ``order`` (creation order decides autograd's execution order: last created runs first); ``use``: which of them reach the loss."""
    terms = {}
    with TF.shared_grad(x) as xs:
        for name in order:
            if name == "match":
                terms[name] = TF.match_head_params(xs, P["q"], P["W1"], P["b1"], P["w2"], P["b2"]).square().mean()
            elif name == "boundary":
                ps, pe = TF.boundary_head_params(xs[:B2 // 2], P["sent"], P["Ws"], P["bs"], P["We"], P["be"], P["w2s"], P["b2s"], P["w2e"], P["b2e"])
                terms[name] = (ps * torch.arange(T, device="cuda")).sum() * 1e-2 + pe.square().sum()
            elif name == "pool":
                a, b, c = TF.moment_pool(xs, *P["m"])
                terms[name] = (a * b).mean() + c.square().mean()
            elif name == "plain":                         # knows nothing about sinks: its gradient reaches _ShareGrad.backward the ordinary way
                terms[name] = (xs * xs).mean()
    return sum(terms[n] for n in use)


@pytest.mark.parametrize("order", list(itertools.permutations(("match", "boundary", "pool"))) + [("plain", "boundary", "match"), ("match", "plain", "pool")])
def test_sink_sum_equals_autograd_sum_in_every_order(order):
    from shufflingvideosfortsg_amd import engine, functional as TF
    P = _params(len(order) + sum(map(len, order)))
    res = []
    with engine.precision("f32s"):
        for on in (True, False):
            old, TF._SHARED_GRAD = TF._SHARED_GRAD, on
            try:
                x = P["x"].clone().requires_grad_(True)
                _loss(TF, P, x, order, order).backward()
                res.append(x.grad.clone())
            finally:
                TF._SHARED_GRAD = old
    TF.check_kernel_errors()
    ref = res[1]
    assert float(ref[B2 // 2:].abs().max()) > 0 and float(ref[:B2 // 2].abs().max()) > 0
    torch.testing.assert_close(res[0], ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))


def test_sink_with_a_consumer_that_never_runs_and_with_only_the_row_block():
    from shufflingvideosfortsg_amd import engine, functional as TF
    P = _params(3)
    with engine.precision("f32s"):
        for order, use in ((("match", "boundary", "pool"), ("match", "pool")), (("boundary", "match"), ("boundary",)), (("pool",), ("pool",))):
            res = []
            for on in (True, False):
                old, TF._SHARED_GRAD = TF._SHARED_GRAD, on
                try:
                    x = P["x"].clone().requires_grad_(True)
                    _loss(TF, P, x, order, use).backward()
                    res.append(x.grad.clone())
                finally:
                    TF._SHARED_GRAD = old
            torch.testing.assert_close(res[0], res[1], rtol=1e-5, atol=1e-6 * float(res[1].abs().max()), msg=lambda m, o=order, u=use: f"{o} / {u}: {m}")
            if use == ("boundary",):
                assert float(res[0][B2 // 2:].abs().max()) == 0.0          # the rows nobody touched are exact zeros
    assert TF._ACTIVE_SINK is None                                          # the context restored the previous state


@pytest.mark.parametrize("order", [("lin", "rows", "pool"), ("pool", "lin", "rows"), ("rows", "pool", "lin")])
def test_sinks_in_the_bf16_storage_mode(order):
    """bf16 storage: the consumers are ``linear`` (input-gradient GEMM with beta = 1 onto the sink), a ``linear`` on the leading rows and ``moment_pool``;
    the sums agree with autograd's bf16 additions to bf16 rounding (the sink rounds ONCE per contribution, from the fp32 accumulator)."""
    from shufflingvideosfortsg_amd import engine, functional as TF
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B2, T, D, generator=g).cuda()
    W = (torch.randn(H, D, generator=g) * D ** -0.5).cuda().requires_grad_(True); W2 = (torch.randn(HM, D, generator=g) * D ** -0.5).cuda().requires_grad_(True)
    ms = [(torch.rand(B2, T, generator=g) > 0.5).float().cuda() for _ in range(3)]
    res = []
    with engine.precision("bf16"):
        for on in (True, False):
            old, TF._SHARED_GRAD = TF._SHARED_GRAD, on
            try:
                x = x0.to(torch.bfloat16).requires_grad_(True)
                terms = []
                with TF.shared_grad(x) as xs:
                    assert (xs is not x) == on
                    for name in order:
                        if name == "lin":
                            terms.append(TF.linear(xs, W).float().square().mean())
                        elif name == "rows":
                            terms.append(TF.linear(xs[:B2 // 2], W2).float().sum() * 1e-3)
                        else:
                            a, b, c = TF.moment_pool(xs, *ms)
                            terms.append((a * b).mean() + c.square().mean())
                sum(terms).backward()
                res.append(x.grad.float().clone())
            finally:
                TF._SHARED_GRAD = old
    err = float((res[0] - res[1]).norm() / res[1].norm())
    assert err <= 1e-2, err
