#!/usr/bin/env python3
"""Per-kernel timing at a given shape with HIP events on the launch stream (developer tool)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32


def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64); ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--N", type=int, default=20); ap.add_argument("--d", type=int, default=1024)
    a = ap.parse_args()
    B, T, N, d = a.B, a.T, a.N, a.d
    lib = _lib.load(); dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev)
    w = torch.randn(d, device=dev) / d ** 0.5; V = torch.randn(B, N, d, device=dev)
    C = torch.empty(B, T, d, device=dev); P = torch.empty(B, T, N, device=dev)
    dC = torch.randn(B, T, d, device=dev)
    da = torch.empty_like(A); ds = torch.empty_like(S); dw = torch.empty_like(w); dV = torch.empty_like(V)
    de = torch.empty(B, T, N, device=dev)
    e = 4
    fwd_bytes = B * ((2 * T + 2 * N) * d * e + T * N * e)
    bwd_bytes = B * ((3 * T + 4 * N) * d * e + T * N * e)
    t = timeit(lambda: lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(C), ptr(P), B, T, N, d, d, TSG_F32, st))
    print(f"scdm_fwd  {t:8.1f} us   {fwd_bytes/t/1e6:7.2f} TB/s-alg  ({fwd_bytes/t/1e6/8*100:.1f}% of 8 TB/s)  {B*T*N*d/t/1e6:.1f} Gelem/s")
    t = timeit(lambda: lib.tsg_scdm_attn_bwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dV), ptr(de), B, T, N, d, d, TSG_F32, st))
    print(f"scdm_bwd  {t:8.1f} us   {bwd_bytes/t/1e6:7.2f} TB/s-alg  ({bwd_bytes/t/1e6/8*100:.1f}% of 8 TB/s)")
    # reference points: a plain device copy of the same bytes
    x = torch.empty(fwd_bytes // 8, device=dev, dtype=torch.float32); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy {fwd_bytes/1e6:.1f} MB (rd+wr)  {t:8.1f} us   {fwd_bytes/t/1e6:7.2f} TB/s")


if __name__ == "__main__":
    main()
