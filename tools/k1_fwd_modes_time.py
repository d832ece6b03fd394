#!/usr/bin/env python3
"""Event-timed K1g forward in the three dtypes (TSG_F32 VALU kernel, TSG_F32S role-specialised, TSG_BF16 role-specialised with the packed-f16
score loop) at [B, T, N, 1024]:   python tools/k1_fwd_modes_time.py [B] [T] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32, TSG_F32S, TSG_BF16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d, n = 1024, 100
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
A = torch.randn(B, T, d, generator=g).cuda(); S = torch.randn(B, N, d, generator=g).cuda(); w = (torch.randn(d, generator=g) / d ** 0.5).cuda()
VW = torch.randn(B, N, d, generator=g).cuda(); gb = torch.randn(d, generator=g).cuda() * 0.1; r = torch.randn(B, T, d, generator=g).cuda()
res = {}
for name, dt, cast in (("f32", TSG_F32, torch.float32), ("f32s", TSG_F32S, torch.float32), ("bf16", TSG_BF16, torch.bfloat16)):
    a_, s_, vw_, r_ = (x.to(cast).contiguous() for x in (A, S, VW, r))
    out = torch.empty_like(a_); P = torch.empty(B, T, N, device="cuda")
    fn = lambda: lib.tsg_scdm_gate_fwd(ptr(a_), ptr(s_), ptr(w), ptr(vw_), ptr(gb), ptr(r_), ptr(out), ptr(P), B, T, N, d, d, dt, st)
    for _ in range(20): assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    eb = 2 if cast == torch.bfloat16 else 4
    alg = ((3 * T + 2 * N) * d * eb + T * N * 4) * B
    res[name] = (out.float(), P.clone())
    print(f"[{B},{T},{N},{d}] {name}: {us:.1f} us  alg {alg/1e6:.1f} MB  {alg/us/1e6:.2f} TB/s = {alg/us/1e6/8*100:.1f}% of 8 TB/s", flush=True)
print("max |P_bf16 - P_f32| = %.2e   max |out_bf16 - out_f32| = %.2e (bf16 inputs differ by rounding)" %
      (float((res["bf16"][1] - res["f32"][1]).abs().max()), float((res["bf16"][0] - res["f32"][0]).abs().max())))
