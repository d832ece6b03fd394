#!/usr/bin/env python3
"""Soak test of the persistent LSTM kernels (developer tool): many launches at several shapes, every launch's error word
checked, forward outputs and gate gradients compared bit for bit with the first launch (the hand-offs carry no ordering
freedom: results must be deterministic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib, functional as TF
from shufflingvideosfortsg_amd._lib import ptr
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for (B, T, h, dt, bm) in [(128, 128, 512, 2, 1), (128, 128, 512, 0, 0), (96, 64, 512, 2, 1), (64, 20, 512, 2, 1), (128, 64, 256, 2, 0), (40, 32, 128, 2, 1),
                          # padded grids (fewer than 8 exchange groups: one group per XCD) and the 16-unit forward of small batches
                          (32, 512, 512, 2, 1), (32, 256, 512, 0, 1), (40, 64, 512, 2, 1), (16, 128, 512, 2, 0), (32, 128, 256, 2, 1), (48, 64, 512, 2, 1)]:
    g = torch.Generator().manual_seed(B + T + h)
    Gx = (torch.randn(T * B, 2, 4 * h, generator=g) * 0.5).cuda(); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
    dOut = torch.randn(T * B, 2 * h, generator=g).cuda(); WT = W.transpose(1, 2).contiguous()
    nb = lib.tsg_lstm_bwd_ws_bytes(B, T, h)
    ref = None
    for it in range(iters):
        sync = torch.zeros(512, dtype=torch.int32, device="cuda")
        out = torch.full((T * B, 2 * h), 9.0, device="cuda"); R = torch.empty(T, 2, B, h, 4, device="cuda"); Cs = torch.empty(T, 2, B, h, device="cuda")
        assert lib.tsg_lstm_fwd_bias(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, dt, bm, st) == 0
        dG = torch.full((T * B, 2, 4 * h), 5.0, device="cuda"); dC = torch.zeros(2, B, h, device="cuda")
        ws = torch.empty(nb // 4 + 4, device="cuda"); db = torch.empty(8 * h, device="cuda")
        assert lib.tsg_lstm_bwd_ws_layout(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), ptr(ws), nb, ptr(db), B, T, h, dt, bm, st) == 0
        if it % 7 == 3:                                   # dirty the caches between launches now and then (another tenant of the L2s / the MALL)
            junk = torch.empty(64 << 20, device="cuda", dtype=torch.uint8).random_(0, 255); del junk
        if it % 25 == 0 or it == iters - 1:
            torch.cuda.synchronize()
            e = int(sync[0]) | int(ws[:1].view(torch.int32)[0])
            if ref is None:
                ref = (out.clone(), dG.clone())
            same = torch.equal(out, ref[0]) and torch.equal(dG, ref[1])
            if e or not same:
                bad += 1
                print(f"[{B},{T},{h}] dtype {dt} bm {bm} iteration {it}: error word {e}, bitwise equal {same}")
    torch.cuda.synchronize()
    TF.check_lstm_errors()
    print(f"[{B},{T},{h}] dtype {dt} batch-major {bm}: {iters} forward + backward launches ok, L2-local workgroups fwd {int(sync[3])} bwd {int(ws[3:4].view(torch.int32)[0])}")
print("soak:", "FAILED" if bad else "passed")
