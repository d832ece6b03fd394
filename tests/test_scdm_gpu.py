"""K1 parity on the GPU: tsg_scdm_attn_{fwd,bwd} (through the C ABI) vs the CPU oracle."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)     # north-star tolerance: 1e-4 fp32


def _run(B, T, N, H, Ds, seed=0, scale=1.0):
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(B, T, H, generator=g) * scale).requires_grad_(True)
    s = (torch.randn(B, N, H, generator=g) * scale).requires_grad_(True)
    w = (torch.randn(H, generator=g) / H ** 0.5).requires_grad_(True)
    sent = torch.randn(B, N, Ds, generator=g).requires_grad_(True)
    gC = torch.randn(B, T, Ds, generator=g)
    C0, P0 = O.scdm_core(a, s, w, sent)
    C0.backward(gC)
    ref = [x.grad.clone() for x in (a, s, w, sent)]
    ad, sd, wd, vd = (x.detach().cuda().requires_grad_(True) for x in (a, s, w, sent))
    C1, P1 = F.scdm_attn(ad, sd, wd, vd, return_p=True)
    C1.backward(gC.cuda())
    torch.cuda.synchronize()
    torch.testing.assert_close(P1.cpu(), P0.detach(), **TOL)
    torch.testing.assert_close(C1.detach().cpu(), C0.detach(), **TOL)
    for got, want, name in zip((ad, sd, wd, vd), ref, "a s w sent".split()):
        # absolute tolerance on the gradient's own scale: dw sums B*T*N terms (|dw| ~ 700 at 130 x 100 x 20, where the
        # fp32 ORACLE is 1.3e-2 from float64 and the kernel 3.4e-4: tools/dw_check.py)
        atol = 2e-4 * max(1.0, float(want.abs().max()))
        torch.testing.assert_close(got.grad.cpu(), want, atol=atol, rtol=1e-3, msg=lambda m, n=name: f"d{n}: {m}")


@pytest.mark.parametrize("shape", [
    (2, 9, 5, 24, 24),       # tiny, H < 256, N not a multiple of 4
    (3, 17, 20, 40, 24),     # H != Ds
    (1, 1, 1, 8, 8),         # degenerate: one clip, one word
    (2, 32, 15, 512, 512),   # BASELINE config 0 shape (Charades N=15, d=512)
    (4, 64, 20, 512, 512),   # config 1 (B reduced for the CPU oracle)
    (2, 128, 20, 1024, 1024),  # north-star tile shape (B reduced)
    (2, 40, 25, 1024, 1024),   # ANet N=25, ragged T
    (1, 33, 32, 260, 516),     # N = 32 max, H and Ds not multiples of 256
])
def test_scdm_parity(shape):
    _run(*shape)


def test_scdm_full_grid_ragged_tiles():
    """Enough pairs that the forward picks its 64-row workgroups (>= 256 tiles; 8 pipelined sub-tiles each) with a
    ragged last tile (T = 100 = 64 + 36), and the 32-row variant (T = 40, B = 140: 280 tiles of 32)."""
    _run(130, 100, 20, 256, 256, seed=5)
    _run(140, 40, 15, 128, 132, seed=6)


def test_scdm_large_activations():
    """|a+s| large: tanh saturates; the exp-product formulation must stay finite and exact."""
    _run(2, 16, 8, 64, 64, seed=3, scale=12.0)


def test_scdm_golden(golden):
    """Same inputs/weights as the reference run captured in tests/golden/scdm_b.npz."""
    from shufflingvideosfortsg_amd import functional as F
    g = golden("scdm_b")
    w = {k: v.cuda() for k, v in g.weights.items()}
    video, sent = g.t("video").cuda(), g.t("sent").cuda().requires_grad_(True)
    a = torch.nn.functional.linear(video, w["W_a.weight"], w["W_a.bias"])
    s = torch.nn.functional.linear(sent, w["W_s.weight"])
    C = F.scdm_attn(a, s, w["w.weight"].reshape(-1), sent)
    torch.testing.assert_close(C.cpu(), g.t("C"), **TOL)


def test_scdm_errors():
    from shufflingvideosfortsg_amd import functional as F
    a = torch.randn(1, 4, 8); s = torch.randn(1, 3, 8); w = torch.randn(8); v = torch.randn(1, 3, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.scdm_attn(a, s, w, v)
    with pytest.raises(RuntimeError, match="N=40"):
        F.scdm_attn(torch.randn(1, 4, 8).cuda(), torch.randn(1, 40, 8).cuda(), w.cuda(), torch.randn(1, 40, 8).cuda())


@pytest.mark.parametrize("shape", [(2, 9, 5, 24), (3, 17, 20, 40), (2, 32, 15, 512), (2, 128, 20, 1024), (2, 40, 25, 1024), (1, 33, 32, 260),
                                   (130, 100, 20, 256)])        # last: 64-row workgroups, ragged last tile
def test_scdm_gate_parity(shape):
    """K1g (attention + sent_linear + sigmoid gate fused, sent_linear reassociated onto the word rows)
    vs the oracle's un-fused tail of rnn_recalibration_layer."""
    from shufflingvideosfortsg_amd import functional as F
    lin = torch.nn.functional.linear
    B, T, N, d = shape
    g = torch.Generator().manual_seed(4)
    r = torch.randn(B, T, d, generator=g, requires_grad=True)          # BiLSTM output
    word = torch.randn(B, N, d, generator=g, requires_grad=True)
    p = {k: (torch.randn(*sh, generator=g) / d ** 0.5).requires_grad_(True) for k, sh in
         dict(Ws=(d, d), Wa=(d, d), ba=(d,), w=(1, d), Wl=(d, d), bl=(d,)).items()}
    gout = torch.randn(B, T, d, generator=g)
    C = O.scdm_attention(r, word, p["Ws"], p["Wa"], p["ba"], p["w"])
    out0 = r * torch.sigmoid(lin(C, p["Wl"], p["bl"]))
    out0.backward(gout)
    leaves = [r, word] + list(p.values())
    ref = [t.grad.clone() for t in leaves]
    rd, wd = r.detach().cuda().requires_grad_(True), word.detach().cuda().requires_grad_(True)
    pd = {k: v.detach().cuda().requires_grad_(True) for k, v in p.items()}
    out1 = F.scdm_gate(lin(rd, pd["Wa"], pd["ba"]), lin(wd, pd["Ws"]), pd["w"], lin(wd, pd["Wl"]), pd["bl"], rd)
    out1.backward(gout.cuda())
    torch.cuda.synchronize()
    torch.testing.assert_close(out1.detach().cpu(), out0.detach(), **TOL)
    for got, want, name in zip([rd, wd] + list(pd.values()), ref, ["r", "word"] + list(p.keys())):
        atol = 3e-4 * max(1.0, float(want.abs().max()))        # on the gradient's own scale (parameter gradients sum B*T terms)
        torch.testing.assert_close(got.grad.cpu(), want, atol=atol, rtol=2e-3, msg=lambda m, n=name: f"d{n}: {m}")


@pytest.mark.parametrize("shape", [(2, 32, 15, 512), (32, 64, 20, 512), (64, 128, 20, 1024), (3, 200, 25, 1024), (5, 100, 32, 256),
                                   (130, 100, 20, 256), (2, 70, 9, 1024), (1, 3, 4, 256), (2, 40, 8, 512), (3, 50, 24, 1024)])
def test_scdm_split_precision_forward(shape, request):
    """The "f32s" mode's forward (dtype TSG_F32S: scdm_fwd_ws_kernel -- producer waves score, consumer waves run phase 2 = P @ VW on the
    bf16 matrix pipe as hi*hi + hi*lo + lo*hi with fp32 accumulation; output columns 256 / 512 / 1024; ragged last tiles, 8- to 64-row
    workgroups; 8 + 8 waves with N <= 16 (one 32x32x16 k step) and 16 < N <= 24 (+ one 32x32x8 step), 4 + 4 waves for N <= 8 and
    N > 24 (two 32x32x16 steps)) vs the oracle at the UNCHANGED fp32 tolerance, gate-fused and plain; the backward is the fp32 kernel."""
    from shufflingvideosfortsg_amd import engine, functional as F
    engine.set_precision("f32s")
    request.addfinalizer(lambda: engine.set_precision(None))
    lin = torch.nn.functional.linear
    B, T, N, d = shape
    g = torch.Generator().manual_seed(14)
    r = torch.randn(B, T, d, generator=g, requires_grad=True)
    word = torch.randn(B, N, d, generator=g, requires_grad=True)
    p = {k: (torch.randn(*sh, generator=g) / d ** 0.5).requires_grad_(True) for k, sh in
         dict(Ws=(d, d), Wa=(d, d), ba=(d,), w=(1, d), Wl=(d, d), bl=(d,)).items()}
    gout = torch.randn(B, T, d, generator=g)
    C0 = O.scdm_attention(r, word, p["Ws"], p["Wa"], p["ba"], p["w"])
    out0 = r * torch.sigmoid(lin(C0, p["Wl"], p["bl"]))
    out0.backward(gout)
    leaves = [r, word] + list(p.values())
    ref = [t.grad.clone() for t in leaves]
    rd, wd = r.detach().cuda().requires_grad_(True), word.detach().cuda().requires_grad_(True)
    pd = {k: v.detach().cuda().requires_grad_(True) for k, v in p.items()}
    a, s = torch.nn.functional.linear(rd, pd["Wa"], pd["ba"]), torch.nn.functional.linear(wd, pd["Ws"])      # exact fp32 projections
    out1 = F.scdm_gate(a, s, pd["w"], lin(wd, pd["Wl"]), pd["bl"], rd)
    C1 = F.scdm_attn(a.detach(), s.detach(), pd["w"].detach(), wd.detach())
    out1.backward(gout.cuda())
    torch.cuda.synchronize()
    torch.testing.assert_close(out1.detach().cpu(), out0.detach(), **TOL)
    torch.testing.assert_close(C1.cpu(), C0.detach(), **TOL)
    for got, want, name in zip([rd, wd] + list(pd.values()), ref, ["r", "word"] + list(p.keys())):
        atol = 3e-4 * max(1.0, float(want.abs().max()))
        torch.testing.assert_close(got.grad.cpu(), want, atol=atol, rtol=2e-3, msg=lambda m, n=name: f"d{n}: {m}")
    # the matrix-pipe forward against the VALU forward on the same operands: the split-precision product is at fp32-GEMM level
    engine.set_precision(None)
    out2 = F.scdm_gate(a.detach(), s.detach(), pd["w"].detach(), lin(wd, pd["Wl"]).detach(), pd["bl"].detach(), rd.detach())
    torch.testing.assert_close(out1.detach(), out2, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("env", [{"TSG_K1_FWD": "mm"}, {"TSG_K1_PW": "4"}])
def test_scdm_split_precision_forward_ab_variants(env):
    """The A/B switches of the f32s forward (TSG_K1_FWD=mm: time-shared roles, scdm_fwd_mm_kernel; TSG_K1_PW=4: 4 + 4 waves) are read
    once per process, so each runs in a child: gate-fused and plain outputs against the fp32 VALU kernel on the same operands."""
    import os, subprocess, sys
    code = r'''
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32, TSG_F32S
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(3)
for (B, T, N, d) in ((5, 100, 20, 1024), (3, 37, 25, 512)):
    A = torch.randn(B, T, d, device="cuda"); S = torch.randn(B, N, d, device="cuda"); w = torch.randn(d, device="cuda") / d ** 0.5
    VW = torch.randn(B, N, d, device="cuda"); gb = torch.randn(d, device="cuda") * 0.1; r = torch.randn(B, T, d, device="cuda")
    res = {}
    for dt in (TSG_F32, TSG_F32S):
        out = torch.empty_like(A); C = torch.empty_like(A); P = torch.empty(B, T, N, device="cuda")
        assert lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, dt, st) == 0
        assert lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(C), ptr(P), B, T, N, d, d, dt, st) == 0
        res[dt] = (out, C, P)
    torch.cuda.synchronize()
    for x, y in zip(res[TSG_F32], res[TSG_F32S]):
        torch.testing.assert_close(x, y, atol=3e-5, rtol=3e-5)
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=root, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_scdm_ws_forward_random_shape_sweep():
    """The role-specialised forward (dtype TSG_F32S, and TSG_BF16 on bf16 copies) against the fp32 VALU kernel over 40 random shapes:
    every N in 1..32 (8 + 8 waves for 8 < N <= 24, 4 + 4 otherwise; one / two 32x32x16 k steps, the 32x32x8 step), T from 1 to 300
    (ragged sub-tiles and tiles, 8- to 64-row workgroups), Ds = H in {256, 512, 1024}; gate-fused and plain; P equal to rounding in f32s."""
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr, TSG_F32, TSG_F32S, TSG_BF16
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(2024)
    ns = list(range(1, 33)) + [20, 25, 15, 24, 9, 8, 17, 28]
    for it, N in enumerate(ns):
        d = (256, 512, 1024)[int(torch.randint(0, 3, (1,), generator=g))]
        T = int(torch.randint(1, 301, (1,), generator=g)) if it % 4 else (1, 7, 8, 9, 63, 64, 65, 128, 200, 257)[it // 4 % 10]
        B = int(torch.randint(1, 9, (1,), generator=g)) if T > 64 else int(torch.randint(1, 70, (1,), generator=g))
        A = torch.randn(B, T, d, generator=g).cuda(); S = torch.randn(B, N, d, generator=g).cuda()
        w = (torch.randn(d, generator=g) / d ** 0.5).cuda(); VW = torch.randn(B, N, d, generator=g).cuda()
        gb = (torch.randn(d, generator=g) * 0.1).cuda(); r = torch.randn(B, T, d, generator=g).cuda()
        res = {}
        for dt, cast in ((TSG_F32, torch.float32), (TSG_F32S, torch.float32), (TSG_BF16, torch.bfloat16)):
            a_, s_, vw_, r_ = (x.to(cast).contiguous() for x in (A, S, VW, r))
            out = torch.empty_like(a_); C = torch.empty_like(a_); P = torch.empty(B, T, N, device="cuda"); P2 = torch.empty_like(P)
            assert lib.tsg_scdm_gate_fwd(ptr(a_), ptr(s_), ptr(w), ptr(vw_), ptr(gb), ptr(r_), ptr(out), ptr(P), B, T, N, d, d, dt, st) == 0, lib.tsg_last_error()
            assert lib.tsg_scdm_attn_fwd(ptr(a_), ptr(s_), ptr(w), ptr(vw_), ptr(C), ptr(P2), B, T, N, d, d, dt, st) == 0, lib.tsg_last_error()
            res[dt] = (out.float(), C.float(), P, P2)
        torch.cuda.synchronize()
        tag = f"shape B={B} T={T} N={N} d={d}"
        o0, c0, p0, q0 = res[TSG_F32]
        o1, c1, p1, q1 = res[TSG_F32S]
        # (the strict-fp32 kernel scores with one reciprocal per element, the role-specialised one with one per four: same value to rounding)
        torch.testing.assert_close(p1, p0, atol=1e-6, rtol=1e-5, msg=lambda m: f"{tag} P gate: {m}")
        torch.testing.assert_close(q1, q0, atol=1e-6, rtol=1e-5, msg=lambda m: f"{tag} P plain: {m}")
        torch.testing.assert_close(o1, o0, atol=3e-5, rtol=3e-5, msg=lambda m: f"{tag} gate: {m}")
        torch.testing.assert_close(c1, c0, atol=3e-5, rtol=3e-5, msg=lambda m: f"{tag} plain: {m}")
        # bf16 storage: against the fp32 kernel on the bf16-rounded operands, within the rounding of the stored output
        ab, sb, vb, rb = (x.to(torch.bfloat16).float() for x in (A, S, VW, r))
        ob = torch.empty_like(ab); cb = torch.empty_like(ab); pb = torch.empty(B, T, N, device="cuda")
        assert lib.tsg_scdm_gate_fwd(ptr(ab), ptr(sb), ptr(w), ptr(vb), ptr(gb), ptr(rb), ptr(ob), ptr(pb), B, T, N, d, d, TSG_F32, st) == 0
        assert lib.tsg_scdm_attn_fwd(ptr(ab), ptr(sb), ptr(w), ptr(vb), ptr(cb), ptr(pb), B, T, N, d, d, TSG_F32, st) == 0
        torch.cuda.synchronize()
        o2, c2, p2, _ = res[TSG_BF16]
        torch.testing.assert_close(o2, ob, atol=2e-2, rtol=1e-2, msg=lambda m: f"{tag} bf16 gate: {m}")
        torch.testing.assert_close(c2, cb, atol=2e-2, rtol=1e-2, msg=lambda m: f"{tag} bf16 plain: {m}")
        torch.testing.assert_close(p2, pb, atol=1e-5, rtol=1e-4, msg=lambda m: f"{tag} bf16 P: {m}")


def _k1_bwd_exchange_run(lib, B, gate, launches, T=128, N=20, d=1024, dirty_every=0):
    """The fused K1 / K1g backward with its partner exchange under the conditions that can expose a publication race (round-3 review):
    the exchange workspace is POISONED with NaN before every launch (a partial row read before its store was visible is then a NaN, not
    the bit-identical stale row of the previous launch), two operand sets ALTERNATE (a stale row of the previous launch belongs to the
    other set), and the reference is the two-kernel path (tsg_scdm_bwd_mode(1): no cross-workgroup exchange at all) on the same
    operands -- the first fused launch of each set must agree with it, every later launch must be bit-identical to the first.
    -> number of launches checked."""
    from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
    st = torch.cuda.current_stream().cuda_stream
    nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, int(gate)))
    sets = []
    for seed in (77, 78):
        g = torch.Generator().manual_seed(seed + B)
        A = torch.randn(B, T, d, generator=g).cuda(); S = torch.randn(B, N, d, generator=g).cuda(); w = (torch.randn(d, generator=g) / d ** 0.5).cuda()
        VW = torch.randn(B, N, d, generator=g).cuda(); gb = (torch.randn(d, generator=g) * 0.1).cuda(); r = torch.randn(B, T, d, generator=g).cuda()
        dout = torch.randn(B, T, d, generator=g).cuda()
        out = torch.empty_like(A); P = torch.empty(B, T, N, device="cuda")
        if gate:
            assert lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, TSG_F32, st) == 0
        else:
            assert lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(out), ptr(P), B, T, N, d, d, TSG_F32, st) == 0
        sets.append(dict(A=A, S=S, w=w, VW=VW, gb=gb, r=r, dout=dout, P=P))

    ws = torch.empty(nb // 4 + 4, device="cuda")

    def launch(o):
        da, ds, dw, dvw = torch.empty_like(o["A"]), torch.empty_like(o["S"]), torch.empty_like(o["w"]), torch.empty_like(o["VW"])
        dgb, dr = torch.empty_like(o["gb"]), torch.empty_like(o["r"])
        ws.fill_(float("nan"))                                       # poison: counters included (the launch zeroes its counters itself)
        if gate:
            rc = lib.tsg_scdm_gate_bwd(ptr(o["A"]), ptr(o["S"]), ptr(o["w"]), ptr(o["VW"]), ptr(o["gb"]), ptr(o["r"]), ptr(o["P"]), ptr(o["dout"]),
                                       ptr(da), ptr(ds), ptr(dw), ptr(dvw), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, d, d, TSG_F32, st)
        else:
            rc = lib.tsg_scdm_attn_bwd(ptr(o["A"]), ptr(o["S"]), ptr(o["w"]), ptr(o["VW"]), ptr(o["P"]), ptr(o["dout"]), ptr(da), ptr(ds), ptr(dw),
                                       ptr(dvw), ptr(ws), nb, B, T, N, d, d, TSG_F32, st)
        assert rc == 0, lib.tsg_last_error()
        return (da, ds, dvw) + ((dr,) if gate else ())

    prev = lib.tsg_scdm_bwd_mode(1)                                  # reference: the two-kernel path (no exchange)
    try:
        refs = [launch(o) for o in sets]
        torch.cuda.synchronize()
    finally:
        lib.tsg_scdm_bwd_mode(prev)
    assert lib.tsg_scdm_bwd_fused_ok(B, T, N, d, d) == 1, "the shape must run the one-launch kernel for this test to mean anything"
    big = torch.empty(256 << 20, device="cuda", dtype=torch.uint8) if dirty_every else None
    first = [None, None]
    for it in range(launches):
        k = it & 1
        if dirty_every and it % dirty_every == 0:
            big.fill_(it & 255)                                      # evict L2 / the Infinity Cache between some launches
        cur = launch(sets[k])
        if first[k] is None:
            torch.cuda.synchronize()
            first[k] = cur
            for x, y in zip(cur, refs[k]):
                assert bool(torch.isfinite(x).all())
                scale = float(y.abs().max())
                torch.testing.assert_close(x, y, atol=2e-5 * max(1.0, scale), rtol=2e-4)
        else:
            for a0, a1 in zip(first[k], cur):
                assert torch.equal(a0, a1), f"launch {it} (operand set {k}) differs from that set's first launch"
    torch.cuda.synchronize()
    return launches


@pytest.mark.parametrize("B,gate", [(4, True), (24, True), (9, False), (128, True)])
def test_scdm_bwd_exchange_is_reproducible(B, gate):
    """The fused backward at small B cuts an item's columns into up to 16 parts that publish partial dP rows (agent-scope stores, an
    explicit vmcnt(0) wait in every thread, the workgroup barrier, one relaxed counter per item -- no release fence) and sum them in part
    order.  80 launches per shape with a NaN-poisoned exchange workspace and two alternating operand sets: first launch of each set
    against the two-kernel path, all later ones bit-identical (see _k1_bwd_exchange_run; tools/k1_bwd_stress.py runs 2 000+ launches
    per batch size, and tests/test_isa_cpu.py gates the vmcnt(0) wait in the shipped ISA)."""
    from shufflingvideosfortsg_amd import _lib
    lib = _lib.load()
    assert _k1_bwd_exchange_run(lib, B, gate, 80, dirty_every=16) == 80


@pytest.mark.parametrize("shape", [(16, 128, 20, 1024), (8, 256, 25, 512)])
def test_scdm_gate_proj_one_node_equals_two_nodes(shape):
    """Round 5: the recalibration block's tail as ONE autograd node (``TF.scdm_gate_proj``: W_a's projection inside the gate's node, the second
    gradient of the clip features added by the input-gradient GEMM's epilogue, tsg_gemm_f32s_nn_acc) against the two-node form (``linear`` +
    ``scdm_gate``, gradients of x summed by autograd): same kernels and the same single fp32 addition per element -- outputs and every gradient
    bit-equal (the two that the K1g backward forms with float atomics: to rounding)."""
    from shufflingvideosfortsg_amd import engine, functional as TF
    B, T, N, D = shape
    g = torch.Generator().manual_seed(B + T + N)
    x0 = torch.randn(B, T, D, generator=g).cuda(); words = torch.randn(B, N, D, generator=g).cuda()
    wa0 = (torch.randn(D, D, generator=g) / D ** 0.5).cuda(); ws0 = (torch.randn(D, D, generator=g) / D ** 0.5).cuda(); ba = torch.randn(D, generator=g).cuda() * 0.1
    w0 = (torch.randn(1, D, generator=g) / D ** 0.5).cuda(); wl0 = (torch.randn(D, D, generator=g) / D ** 0.5).cuda(); bl0 = torch.randn(D, generator=g).cuda() * 0.1
    dout = torch.randn(B, T, D, generator=g).cuda()
    res = []
    with engine.precision("f32s"):
        for one in (True, False):
            x, wa, w, wl, bl, wsn = (t.clone().requires_grad_(True) for t in (x0, wa0, w0, wl0, bl0, ws0))
            VW = TF.linear(words, wl)
            s = TF.linear(words, wsn, ba)
            if one:
                assert TF.scdm_gate_proj_ok(x, wa, VW)
                out = TF.scdm_gate_proj(x, wa, s, w, VW, bl)
            else:
                out = TF.scdm_gate(TF.linear(x, wa, None), s, w, VW, bl, x)
            out.backward(dout)
            res.append((out.detach(), x.grad, wa.grad, w.grad, wl.grad, bl.grad, wsn.grad))
    TF.check_kernel_errors()
    for a, b, name in zip(res[0], res[1], ("out", "dx", "dW_a", "dw", "dW_l", "db_l", "dW_s")):
        if name in ("dw", "db_l"):       # float atomics across workgroups inside the K1g backward: equal to rounding, not to the bit, in either form
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
        else:
            assert torch.equal(a, b), f"{name}: one-node and two-node forms differ (max {float((a - b).abs().max()):.3e})"


def test_timed_launch_brackets_the_kernel():
    """tsg_time_next_launch: the K1g forward's own event pair reports a duration, shorter than a pair recorded around the call."""
    import ctypes
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr, TSG_F32S
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    B, T, N, d = 16, 128, 20, 1024
    A = torch.randn(B, T, d, device="cuda"); S = torch.randn(B, N, d, device="cuda"); w = torch.randn(d, device="cuda") / 32
    VW = torch.randn(B, N, d, device="cuda"); gb = torch.randn(d, device="cuda"); r = torch.randn(B, T, d, device="cuda")
    out = torch.empty_like(A); P = torch.empty(B, T, N, device="cuda")
    run = lambda: lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, TSG_F32S, st)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    ref = out.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    assert lib.tsg_time_next_launch(5) == 0
    e0.record(); assert run() == 0; e1.record()
    us = ctypes.c_float()
    assert lib.tsg_timed_launch_us(5, ctypes.byref(us)) == 0
    torch.cuda.synchronize()
    assert 1.0 < us.value <= e0.elapsed_time(e1) * 1e3 + 1.0
    assert torch.equal(out, ref)                                         # the timed launch is the same launch
    assert run() == 0                                                    # the hook disarmed itself
    assert lib.tsg_timed_launch_us(6, ctypes.byref(us)) == -2
