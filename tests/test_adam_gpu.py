"""tsg_adam_step / engine.TsgAdam (csrc/adam.hip; round 5): the reference's optimizer -- torch.optim.Adam(lr=1e-3, weight_decay=1e-4 (L2), eps=1e-6),
grounding/train.py:367-371 -- as ONE launch per 64 tensors, against torch's own Adam on the same gradients: parameters and both moments after
several updates, tensors of every size class (scalars, odd lengths, exactly one chunk, many chunks), more than 64 tensors, the on-device skip
flag (found_inf: parameters, moments AND the update count untouched), the 1 / world gradient scale, and a state_dict round trip."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (7, 5), (1024,), (8192,), (8193,), (2048, 1024), (300, 300), (2, 2048, 512), (4096,), (33, 17)] + [(64, 64)] * 60


def _models(seed):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.randn(*s, generator=g).cuda() for s in SHAPES]
    a = [torch.nn.Parameter(p.clone()) for p in ps]
    b = [torch.nn.Parameter(p.clone()) for p in ps]
    return a, b, g


def test_tsg_adam_matches_torch_adam():
    from shufflingvideosfortsg_amd.engine import TsgAdam
    a, b, g = _models(0)
    assert len(a) > 64                                    # two launches per update
    own = TsgAdam(a, lr=1e-3, weight_decay=1e-4, eps=1e-6)
    ref = torch.optim.Adam(b, lr=1e-3, weight_decay=1e-4, eps=1e-6)
    for it in range(6):
        for pa, pb in zip(a, b):
            gr = (torch.randn(pa.shape, generator=g) * (10.0 ** (it - 3))).cuda()     # gradient scales over six decades
            pa.grad, pb.grad = gr.clone(), gr.clone()
        own.step(); ref.step()
    torch.cuda.synchronize()
    assert float(own.state[a[0]]["step"]) == 6.0
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa.data, pb.data, rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(own.state[pa]["exp_avg"], ref.state[pb]["exp_avg"], rtol=2e-6, atol=1e-10)
        torch.testing.assert_close(own.state[pa]["exp_avg_sq"], ref.state[pb]["exp_avg_sq"], rtol=2e-6, atol=1e-12)


def test_tsg_adam_skip_flag_scale_and_state_dict():
    from shufflingvideosfortsg_amd.engine import TsgAdam
    a, b, g = _models(1)
    own = TsgAdam(a, lr=1e-3, weight_decay=1e-4, eps=1e-6)
    ref = torch.optim.Adam(b, lr=1e-3, weight_decay=1e-4, eps=1e-6)

    def grads(scale_own=1.0):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad, pb.grad = gr * scale_own, gr.clone()
    grads(); own.step(); ref.step()
    before = [p.detach().clone() for p in a]
    m_before = [own.state[p]["exp_avg"].clone() for p in a]
    grads()
    own.found_inf = torch.ones((), device="cuda")          # skip: nothing moves, the count stays at 1
    own.step(); own.found_inf = None
    torch.cuda.synchronize()
    assert float(own.state[a[0]]["step"]) == 1.0
    for p, q, m0 in zip(a, before, m_before):
        assert torch.equal(p.data, q) and torch.equal(own.state[p]["exp_avg"], m0)
    own.found_inf = torch.zeros((), device="cuda")         # flag present but clear: the update runs (same gradients as the skipped call)
    own.step(); own.found_inf = None
    ref.step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa.data, pb.data, rtol=2e-6, atol=1e-7)
    # the 1 / world of a SUM all-reduce folded into the update: gradients 4x too large, grad_scale 1/4
    grads(scale_own=4.0)
    own.grad_scale = 0.25
    own.step(); ref.step()
    own.grad_scale = 1.0
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa.data, pb.data, rtol=2e-6, atol=1e-7)
    # state_dict round trip into a fresh optimizer on fresh parameters: the count and the moments come along
    import copy
    sd = copy.deepcopy(own.state_dict())          # (load_state_dict does not copy tensors that already have the right dtype / device)
    c = [torch.nn.Parameter(p.detach().clone()) for p in a]
    own2 = TsgAdam(c, lr=1e-3, weight_decay=1e-4, eps=1e-6)
    own2.load_state_dict(sd)
    grads()
    for pc, pa in zip(c, a):
        pc.grad = pa.grad.clone()
    own.step(); own2.step(); ref.step()
    torch.cuda.synchronize()
    assert float(own2.state[c[0]]["step"]) == float(own.state[a[0]]["step"]) == 4.0
    for pc, pa, pb in zip(c, a, b):
        assert torch.equal(pc.data, pa.data)
        torch.testing.assert_close(pa.data, pb.data, rtol=3e-6, atol=1e-7)


def test_engine_optimizer_step_uses_the_own_adam_and_skips_on_the_device():
    from shufflingvideosfortsg_amd import engine
    torch.manual_seed(0)
    m = torch.nn.Linear(64, 64).cuda()
    opt = engine.make_optimizer(m, dict(lr=1e-3, weight_decay=1e-4))
    assert isinstance(opt, engine.TsgAdam)
    x = torch.randn(8, 64, device="cuda")
    w0 = m.weight.detach().clone()
    loss = m(x).pow(2).mean(); loss.backward()
    engine.optimizer_step(opt, loss * float("nan"))        # a non-finite loss: skipped on the device, no host sync
    assert torch.equal(m.weight.detach(), w0)
    engine.optimizer_step(opt, loss)
    assert not torch.equal(m.weight.detach(), w0)


def test_tsg_adam_keeps_bf16_shadows_of_the_parameters():
    """Round 5: the bf16 storage mode reads bf16 SHADOWS of the fp32 parameters (functional.weight_bf16); the optimizer's kernel rewrites them with
    every update (tsg_adam_step_shadow), so a training step casts no weight.  After every update the shadow is, bit for bit, the parameter
    rounded to bf16; a skipped update leaves it alone; a change of the parameter behind the optimizer's back (load_state_dict / copy_) is
    noticed through the version counter and the shadow is re-made at its next use; parameters without a shadow are updated as before."""
    from shufflingvideosfortsg_amd import functional as TF
    from shufflingvideosfortsg_amd.engine import TsgAdam
    a, b, g = _models(3)
    own = TsgAdam(a, lr=1e-3, weight_decay=1e-4, eps=1e-6)
    ref = torch.optim.Adam(b, lr=1e-3, weight_decay=1e-4, eps=1e-6)
    shadowed = a[::2]                                      # every second tensor has been used as a bf16 operand (all size classes; both launches)
    for p in shadowed:
        sh = TF.weight_bf16(p)
        assert sh.dtype == torch.bfloat16 and TF.shadow_of(p) is sh and torch.equal(sh, p.detach().to(torch.bfloat16))
    ptrs = [TF.weight_bf16(p).data_ptr() for p in shadowed]
    for it in range(4):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if it == 2:
            own.found_inf = torch.ones((), device="cuda")  # a skipped update: parameters AND shadows stay
            keep = [TF.shadow_of(p).clone() for p in shadowed]
            own.step(); own.found_inf = None
            for p, k in zip(shadowed, keep):
                assert torch.equal(TF.shadow_of(p), k)
            continue
        own.step(); ref.step()
        for p in shadowed:
            assert torch.equal(TF.shadow_of(p), p.detach().to(torch.bfloat16)), f"shadow of a {tuple(p.shape)} tensor after update {it}"
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa.data, pb.data, rtol=2e-6, atol=1e-7)
    assert [TF.weight_bf16(p).data_ptr() for p in shadowed] == ptrs          # the same buffers all along: no cast, no reallocation
    with torch.no_grad():
        shadowed[3].copy_(torch.full_like(shadowed[3], 0.333))               # behind the optimizer's back: the version counter moves
    assert TF.shadow_of(shadowed[3]) is None
    assert torch.equal(TF.weight_bf16(shadowed[3]), shadowed[3].detach().to(torch.bfloat16)) and TF.shadow_of(shadowed[3]) is not None


def test_shadow_goes_stale_through_dot_data_and_invalidate_shadows_drops_it():
    """a write through ``p.data`` does not move ``p._version`` (``.data`` has its own counter), so the shadow check cannot see it --
    that is the documented limit; ``functional.invalidate_shadows`` is the remedy (dp.FlatGradAllReduce calls it after its broadcast)."""
    from shufflingvideosfortsg_amd import functional as TF
    lin = torch.nn.Linear(64, 32).cuda()
    sh = TF.weight_bf16(lin.weight)
    lin.weight.data.mul_(2.0)                              # behind the version counter
    assert TF.shadow_of(lin.weight) is sh and not torch.equal(sh, lin.weight.detach().to(torch.bfloat16))      # stale, undetected
    assert TF.invalidate_shadows(lin) == 1 and TF.shadow_of(lin.weight) is None
    assert torch.equal(TF.weight_bf16(lin.weight), lin.weight.detach().to(torch.bfloat16))
    assert TF.invalidate_shadows([lin.bias]) == 0          # parameters that never had a shadow


def test_tsg_adam_bumps_versions_so_a_late_backward_fails_loudly():
    """The update and the shadow rewrite go through raw pointers; TsgAdam bumps the version counters afterwards, so a backward through a graph
    that SAVED the old weight (or its shadow) raises autograd's in-place error instead of silently using the new values -- torch's own
    optimizers behave the same way -- and the shadow stays current (no re-cast at the next use)."""
    from shufflingvideosfortsg_amd import functional as TF
    from shufflingvideosfortsg_amd.engine import TsgAdam
    w = torch.nn.Parameter(torch.randn(32, 16, device="cuda"))
    opt = TsgAdam([w], lr=1e-2)
    sh = TF.weight_bf16(w)
    x = torch.randn(4, 16, device="cuda", requires_grad=True)
    y = x @ w.t()                                          # saves w for the backward
    v0 = w._version
    w.grad = torch.randn_like(w)
    opt.step()
    assert w._version > v0 and TF.shadow_of(w) is sh and torch.equal(sh, w.detach().to(torch.bfloat16))
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()


def test_tsg_adam_param_groups_keep_their_own_update_count():
    """two param groups on one device used to share (and both advance) one update count; each group's bias correction now sees its own t."""
    from shufflingvideosfortsg_amd.engine import TsgAdam
    g = torch.Generator().manual_seed(5)
    ps = [torch.randn(300, generator=g).cuda(), torch.randn(17, 9, generator=g).cuda(), torch.randn(4096, generator=g).cuda()]
    a = [torch.nn.Parameter(p.clone()) for p in ps]
    b = [torch.nn.Parameter(p.clone()) for p in ps]
    mk = lambda cls, q: cls([dict(params=q[:2], lr=1e-3), dict(params=q[2:], lr=3e-3, betas=(0.8, 0.99))], eps=1e-6, weight_decay=1e-4)
    own, ref = mk(TsgAdam, a), mk(torch.optim.Adam, b)
    for it in range(5):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad, pb.grad = gr.clone(), gr.clone()
        own.step(); ref.step()
    assert float(own.state[a[0]]["step"]) == 5.0 and float(own.state[a[2]]["step"]) == 5.0
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa.data, pb.data, rtol=3e-6, atol=1e-7)


def test_guarded_update_skips_on_non_finite_gradients():
    """the guard looked at the loss and the error words only.  tsg_grads_nonfinite puts the gradients themselves into the skip flag: one
    NaN or infinity anywhere (first, last, odd-sized and unaligned tensors; the second launch's table) and nothing moves; finite gradients pass."""
    import ctypes
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd.engine import TsgAdam
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    a, _, g = _models(7)

    def flag_of(grads):
        n = len(grads); f = torch.zeros((), device="cuda")
        rc = lib.tsg_grads_nonfinite(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads]), (ctypes.c_longlong * n)(*[t.numel() for t in grads]), f.data_ptr(), st)
        assert rc == 0, lib.tsg_last_error()
        return float(f)
    grads = [torch.randn(p.shape, generator=g).cuda() for p in a]
    flat = torch.randn(5000, generator=g).cuda()
    grads.append(flat[1:4098])                             # 4-byte aligned only (a view into a flat buffer)
    assert flag_of(grads) == 0.0
    for which, pos, val in ((0, 0, float("nan")), (4, 8191, float("inf")), (6, -1, float("-inf")), (len(a) - 1, 100, float("nan")), (len(grads) - 1, 4096, float("inf"))):
        keep = grads[which].reshape(-1)[pos].clone()
        grads[which].reshape(-1)[pos] = val
        assert flag_of(grads) == 1.0, (which, pos, val)
        grads[which].reshape(-1)[pos] = keep
    assert flag_of(grads) == 0.0
    own = TsgAdam(a, lr=1e-3)
    for p, gr in zip(a, grads):
        p.grad = gr.clone()
    own.step()
    before = [p.detach().clone() for p in a]
    a[70].grad.reshape(-1)[3] = float("nan")               # a tensor of the second launch
    own.found_inf = torch.zeros((), device="cuda")         # a guarded update whose other inputs (loss, error words) are clean
    own.step()
    torch.cuda.synchronize()
    assert float(own.found_inf) == 1.0 and float(own.state[a[0]]["step"]) == 1.0
    own.found_inf = None
    for p, q in zip(a, before):
        assert torch.equal(p.data, q)
