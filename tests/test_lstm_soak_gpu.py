"""Soak of the persistent LSTM kernels (promoted from tools/lstm_soak.py): repeated forward + backward launches on REUSED
buffers holding the previous launch's results (stale non-sentinel data / stale generation tags at the same addresses must
never be accepted), both exchange modes, error words clean, bitwise determinism of out / R / Cs / dG, and -- random data --
agreement with the launch-per-step kernels; dbias (float atomics across batch slices) to 1e-5 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # B, T, h, dtype (0 = fp32 MFMA, 2 = split-precision), batch_major
    (128, 128, 512, 2, 1), (128, 64, 512, 0, 0), (96, 64, 512, 2, 1), (64, 20, 512, 2, 1), (128, 64, 256, 2, 0),
    (40, 32, 128, 2, 1), (37, 16, 256, 0, 1), (256, 16, 512, 2, 1)]


@pytest.mark.parametrize("l2x", [1, 0])
def test_persistent_lstm_soak(l2x):
    from shufflingvideosfortsg_amd import _lib, functional as TF
    from shufflingvideosfortsg_amd._lib import ptr
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    TF.check_lstm_errors()                                   # registers the sink, runs the start-up self-test
    iters = 200 // len(SHAPES) + 1                           # >= 200 persistent forward + backward launches per mode
    lib.tsg_lstm_set_l2_exchange(l2x)
    try:
        for (B, T, h, dt, bm) in SHAPES:
            g = torch.Generator().manual_seed(B + T + h)
            Gx = (torch.randn(T * B, 2, 4 * h, generator=g) * 0.5).cuda(); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
            dOut = torch.randn(T * B, 2 * h, generator=g).cuda(); WT = W.transpose(1, 2).contiguous()
            nb = lib.tsg_lstm_bwd_ws_bytes(B, T, h)
            sync = torch.zeros(512, dtype=torch.int32, device="cuda")
            out = torch.full((T * B, 2 * h), 9.0, device="cuda"); R = torch.empty(T, 2, B, h, 4, device="cuda"); Cs = torch.empty(T, 2, B, h, device="cuda")
            dG = torch.full((T * B, 2, 4 * h), 5.0, device="cuda"); dC = torch.zeros(2, B, h, device="cuda")
            ws = torch.zeros(nb // 4 + 4, device="cuda"); db = torch.empty(8 * h, device="cuda")

            def launch():
                assert lib.tsg_lstm_fwd_bias(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, dt, bm, st) == 0
                assert lib.tsg_lstm_bwd_ws_layout(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), ptr(ws), nb, ptr(db), B, T, h, dt, bm, st) == 0
            # reference: the launch-per-step kernels on the same data
            lib.tsg_lstm_set_persist(0)
            launch(); torch.cuda.synchronize()
            step = (out.clone(), dG.clone(), dG.view(T * B, 8 * h).sum(0))
            lib.tsg_lstm_set_persist(1)
            assert lib.tsg_lstm_bwd_ws_persistent(B, T, h, nb) == 1
            first = None
            for it in range(iters):                          # buffers are NOT re-initialised: they hold the last launch's results
                launch()
                if it % 8 == 0 or it == iters - 1:
                    torch.cuda.synchronize()
                    assert int(sync[0]) == 0 and int(ws[:1].view(torch.int32)[0]) == 0, (B, T, h, dt, bm, it)
                    cur = (out.clone(), R.clone(), Cs.clone(), dG.clone(), db.clone())
                    if first is None:
                        first = cur
                    for a, b, n in zip(cur[:4], first[:4], ("out", "R", "Cs", "dG")):
                        assert torch.equal(a, b), f"{n} not bitwise reproducible at launch {it} of {(B, T, h, dt, bm)}"
                    torch.testing.assert_close(cur[4], first[4], atol=1e-5 * float(first[4].abs().max()), rtol=1e-5)
            TF.check_lstm_errors()
            tol = dict(atol=3e-5, rtol=1e-4) if dt == 2 else dict(atol=2e-6, rtol=1e-5)
            torch.testing.assert_close(first[0], step[0], **tol)
            torch.testing.assert_close(first[3], step[1], atol=tol["atol"] * 20, rtol=1e-3)
            torch.testing.assert_close(first[4], step[2], atol=1e-4 * max(1.0, float(step[2].abs().max())), rtol=1e-3)
    finally:
        lib.tsg_lstm_set_persist(-1)
        lib.tsg_lstm_set_l2_exchange(1)


POISON_SHAPES = [(128, 48, 512, 2, 1), (128, 48, 512, 1, 1), (96, 32, 512, 2, 1), (32, 64, 512, 1, 1), (40, 24, 256, 0, 0)]


@pytest.mark.parametrize("l2x", [1, 0])
def test_persistent_lstm_poisoned_exchange_buffers(l2x):
    """Round-4 review: the backward's 2-slot generation-tag ring had a soak on reused buffers but no POISONED run.  Before EVERY launch the
    exchange memory -- the forward's ring workspace (tsg_lstm_fwd_ws, round 5) and the backward's partial-dh ring -- is overwritten with one of
    four patterns: quiet NaN (low mantissa bit 0 = "even generation"), NaN | 1 ("odd generation"), the forward's own sentinel pattern, and
    finite garbage with random tags; two operand sets alternate, so a stale slot of the previous launch would carry the OTHER set's values.
    out / R / Cs / dG must equal, bit for bit, what the same kernels give on a clean workspace, every launch; error words stay clean.
    f32s, bf16 storage and strict fp32; full-chip, 12-group and padded grids; both exchange modes."""
    from shufflingvideosfortsg_amd import _lib, functional as TF
    from shufflingvideosfortsg_amd._lib import ptr
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    TF.check_lstm_errors()
    lib.tsg_lstm_set_l2_exchange(l2x)
    lib.tsg_lstm_set_ring(1)                                 # (bf16 storage at >= 96 rows then runs the 64-unit kernel: its default)
    patterns = (0x7fc00000, 0x7fc00001, 0x7fa5c3e1, None)
    try:
        for (B, T, h, dt, bm) in POISON_SHAPES:
            bf = dt == 1
            seq = torch.bfloat16 if bf else torch.float32
            nfw, nb = lib.tsg_lstm_fwd_ws_bytes(B, T, h), lib.tsg_lstm_bwd_ws_bytes(B, T, h)
            assert lib.tsg_lstm_bwd_ws_persistent(B, T, h, nb) == 1
            sets = []
            for k in range(2):
                g = torch.Generator().manual_seed(B + T + h + 97 * k)
                Gx = (torch.randn(T * B, 2, 4 * h, generator=g) * 0.5).cuda().to(seq); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
                dOut = torch.randn(T * B, 2 * h, generator=g).cuda().to(seq)
                sets.append((Gx, W, W.transpose(1, 2).contiguous(), dOut))
            fws = torch.zeros(nfw // 4, dtype=torch.int32, device="cuda")
            ws = torch.zeros(nb // 4 + 4, device="cuda")
            out = torch.empty(T * B, 2 * h, device="cuda", dtype=seq); R = torch.empty(T, 2, B, h, 4, device="cuda", dtype=seq)
            Cs = torch.empty(T, 2, B, h, device="cuda"); dG = torch.empty(T * B, 2, 4 * h, device="cuda", dtype=seq)
            dC = torch.zeros(2, B, h, device="cuda"); db = torch.empty(8 * h, device="cuda")

            def launch(k):
                Gx, W, WT, dOut = sets[k]
                assert lib.tsg_lstm_fwd_ws(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(fws), nfw, B, T, h, dt, bm, st) == 0
                assert lib.tsg_lstm_bwd_ws_layout(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), ptr(ws), nb, ptr(db), B, T, h, dt, bm, st) == 0
            want = []
            for k in range(2):                                # clean workspaces: the reference
                fws.zero_(); ws.zero_()
                launch(k); torch.cuda.synchronize()
                assert int(fws[0]) == 0 and int(ws[:1].view(torch.int32)[0]) == 0
                want.append((out.clone(), R.clone(), Cs.clone(), dG.clone()))
            gp = torch.Generator(device="cuda").manual_seed(5)
            for it in range(40):
                pat = patterns[it % 4]
                if pat is None:
                    fws[512:] = torch.randint(-2 ** 31, 2 ** 31 - 1, (fws.numel() - 512,), device="cuda", generator=gp, dtype=torch.int64).to(torch.int32)
                    ws.view(torch.int32)[512:] = torch.randint(-2 ** 31, 2 ** 31 - 1, (ws.numel() - 512,), device="cuda", generator=gp, dtype=torch.int64).to(torch.int32)
                else:
                    p32 = pat - (1 << 32) if pat >= (1 << 31) else pat
                    fws[512:] = p32; ws.view(torch.int32)[512:] = p32
                k = it & 1
                launch(k)
                torch.cuda.synchronize()
                assert int(fws[0]) == 0 and int(ws[:1].view(torch.int32)[0]) == 0, (B, T, h, dt, bm, it)
                for a, b, n in zip((out, R, Cs, dG), want[k], ("out", "R", "Cs", "dG")):
                    assert torch.equal(a, b), f"{n} differs from the clean-workspace result at poisoned launch {it} (pattern {pat}) of {(B, T, h, dt, bm)}"
            TF.check_lstm_errors()
    finally:
        lib.tsg_lstm_set_ring(-1)
        lib.tsg_lstm_set_l2_exchange(1)
