#!/usr/bin/env python3
"""K1g / K1 forward + backward, every storage / arithmetic variant, back-to-back launches on resident buffers:
    python tools/k1_variants.py [B] [iters] [N] [T]
Prints microseconds per launch and the fraction of the 8 TB/s HBM roofline (algorithmic bytes with the variant's element size),
and the max deviation of the TSG_F32S (MFMA phase 2) and TSG_BF16 forwards from the TSG_F32 kernel on the same data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32, TSG_F32S, TSG_BF16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
T = int(sys.argv[4]) if len(sys.argv) > 4 else 128
d = int(os.environ.get("D", 1024))
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev)
w = torch.randn(d, device=dev) / d ** 0.5
VW = torch.randn(B, N, d, device=dev); gb = torch.randn(d, device=dev) * 0.1; r = torch.randn(B, T, d, device=dev)
dout = torch.randn(B, T, d, device=dev)


def timeit(fn):
    for _ in range(60):                         # long warm-up: the first launches of a process run at other clocks (10 % bias)
        rc = fn()
        assert rc == 0, lib.tsg_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
for name, dt, cast in (("f32", TSG_F32, torch.float32), ("f32s", TSG_F32S, torch.float32), ("bf16", TSG_BF16, torch.bfloat16),
                       ("f32", TSG_F32, torch.float32), ("f32s", TSG_F32S, torch.float32)):
    a_, s_, vw_, r_, do_ = (x.to(cast).contiguous() for x in (A, S, VW, r, dout))
    out = torch.empty(B, T, d, device=dev, dtype=cast); P = torch.empty(B, T, N, device=dev)
    C = torch.empty(B, T, d, device=dev, dtype=cast)
    e = 2 if cast == torch.bfloat16 else 4
    g = timeit(lambda: lib.tsg_scdm_gate_fwd(ptr(a_), ptr(s_), ptr(w), ptr(vw_), ptr(gb), ptr(r_), ptr(out), ptr(P), B, T, N, d, d, dt, st))
    p = timeit(lambda: lib.tsg_scdm_attn_fwd(ptr(a_), ptr(s_), ptr(w), ptr(vw_), ptr(C), ptr(P), B, T, N, d, d, dt, st))
    res[name] = (out.float().clone(), C.float().clone(), P.clone())
    bg = ((3 * T + 2 * N) * d * e + T * N * 4) * B; bp = ((2 * T + 2 * N) * d * e + T * N * 4) * B
    line = f"{name:5s} B={B} T={T} N={N} d={d}: gate fwd {g:6.1f} us ({bg / g / 1e3 / 8000 * 100:4.1f}%)  plain fwd {p:6.1f} us ({bp / p / 1e3 / 8000 * 100:4.1f}%)"
    if dt != TSG_F32S:
        da, ds, dw = torch.empty_like(a_), torch.empty_like(s_), torch.empty_like(w)
        dvw, dgb, dr = torch.empty_like(vw_), torch.empty_like(gb), torch.empty_like(r_)
        nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, 1)); ws = torch.empty(nb // 4 + 4, device=dev)
        bw = timeit(lambda: lib.tsg_scdm_gate_bwd(ptr(a_), ptr(s_), ptr(w), ptr(vw_), ptr(gb), ptr(r_), ptr(P), ptr(do_), ptr(da), ptr(ds), ptr(dw),
                                                  ptr(dvw), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, d, d, dt, st))
        bb = ((5 * T + 4 * N) * d * e + T * N * 4) * B
        line += f"  gate bwd {bw:6.1f} us ({bb / bw / 1e3 / 8000 * 100:4.1f}%)"
    print(line, flush=True)
ref = res["f32"]
for name in ("f32s", "bf16"):
    o, c, p = res[name]
    print(f"{name} vs f32: max|d out| {float((o - ref[0]).abs().max()):.3e}  max|d C| {float((c - ref[1]).abs().max()):.3e}  max|d P| {float((p - ref[2]).abs().max()):.3e}")
