#!/usr/bin/env python3
"""Event-timed K1 backward (gate-fused and plain) at B pairs per launch: python tools/k1_bwd_time.py [B] [N] [T]
(TSG_K1_BWD=split in the environment times the two-kernel path of revision 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
d, n = 1024, 30
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev); w = torch.randn(d, device=dev) / d ** 0.5
P = torch.softmax(torch.randn(B, T, N, device=dev), -1); VW = torch.randn(B, N, d, device=dev); gb = torch.randn(d, device=dev)
r = torch.randn(B, T, d, device=dev); dC = torch.randn(B, T, d, device=dev)
da = torch.empty_like(A); ds = torch.empty_like(S); dw = torch.empty_like(w); dVW = torch.empty_like(VW); dgb = torch.empty_like(gb)
dr = torch.empty_like(r)
nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, 1)); ws = torch.empty(nb // 4 + 4, device=dev)
def timeit(fn):
    for _ in range(5): assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = timeit(lambda: lib.tsg_scdm_gate_bwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dVW), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, d, d, TSG_F32, st))
p = timeit(lambda: lib.tsg_scdm_attn_bwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dVW), ptr(ws), nb, B, T, N, d, d, TSG_F32, st))
bg = ((5 * T + 4 * N) * d * 4 + T * N * 4) * B; bp = ((3 * T + 4 * N) * d * 4 + T * N * 4) * B
print(f"B={B} T={T} N={N} [{os.environ.get('TSG_K1_BWD', 'fused')}]: gate bwd {g:.1f} us ({bg/g/1e3/8000*100:.1f}% of 8 TB/s)   plain bwd {p:.1f} us ({bp/p/1e3/8000*100:.1f}%)")
