#!/usr/bin/env python3
"""Developer tool: 400 launches of the gate-fused K1 backward per batch size (1 .. 16 column parts per item), caches dirtied in between;
counts launches whose direct outputs (da, ds, dVW, dr) differ from the first one -- a partner's partial dP row read before it was visible
would show up here.  python tools/k1_bwd_stress.py"""
import sys, torch
sys.path.insert(0, ".")
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
for B in (3, 9, 17, 40, 128):
    T, N, d = 128, 20, 1024
    g = torch.Generator().manual_seed(B)
    A = torch.randn(B, T, d, generator=g).cuda(); S = torch.randn(B, N, d, generator=g).cuda(); w = (torch.randn(d, generator=g) / d ** 0.5).cuda()
    VW = torch.randn(B, N, d, generator=g).cuda(); gb = (torch.randn(d, generator=g) * 0.1).cuda(); r = torch.randn(B, T, d, generator=g).cuda()
    dout = torch.randn(B, T, d, generator=g).cuda()
    out = torch.empty_like(A); P = torch.empty(B, T, N, device="cuda")
    lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, TSG_F32, st)
    nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, 1))
    first = None; bad = 0
    big = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
    for it in range(400):
        da, ds, dw, dvw = torch.empty_like(A), torch.empty_like(S), torch.empty_like(w), torch.empty_like(VW)
        dgb, dr = torch.empty_like(gb), torch.empty_like(r)
        ws = torch.empty(nb // 4 + 4, device="cuda")
        if it % 7 == 0: big.fill_(it & 255)                  # dirty the caches between some launches
        rc = lib.tsg_scdm_gate_bwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(P), ptr(dout), ptr(da), ptr(ds), ptr(dw), ptr(dvw), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, d, d, TSG_F32, st)
        assert rc == 0
        cur = (da, ds, dvw, dr)
        if first is None: first = cur
        else:
            if not all(torch.equal(a, b) for a, b in zip(first, cur)): bad += 1
    torch.cuda.synchronize()
    print("B", B, "mismatching launches:", bad, flush=True)
