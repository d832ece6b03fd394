#!/usr/bin/env python3
"""Timing-only ablation of the K1g forward kernels (needs tools/_ablate/k1abl.so = a -DTSG_ABLATE build): TSG_ABLATE_MASK bit 1 skips
the score loop, bit 4 phase 2.  python tools/k1_mm_ablate.py [B]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "128"
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr
B, T, N, d = int(sys.argv[1]), 128, 20, 1024
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev); w = torch.randn(d, device=dev) / 32
VW = torch.randn(B, N, d, device=dev); gb = torch.randn(d, device=dev); r = torch.randn(B, T, d, device=dev)
out = torch.empty(B, T, d, device=dev); P = torch.empty(B, T, N, device=dev)
for dt in (0, 2):
    fn = lambda: lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, dt, st)
    for _ in range(100): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print("mask", os.environ.get("TSG_ABLATE_MASK", "0"), "dtype", dt, "%%.1f us" %% (e0.elapsed_time(e1) / 200 * 1e3), flush=True)
''' % root
for mask in ("0", "1", "4", "5", "8", "13", "0"):
    env = dict(os.environ, TSG_HIP_LIB=os.path.join(root, "tools", "_ablate", "k1abl.so"), TSG_ABLATE_MASK=mask)
    subprocess.run([sys.executable, "-c", code, B], env=env)
