#!/usr/bin/env python3
"""tsg_gemm_f32s (split on load) against the operand passes + library bf16 GEMM it would replace, at the step's GEMM shapes:
python tools/gemm_f32s_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib, functional as F
from shufflingvideosfortsg_amd._lib import ptr
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
F.set_gemm_dtype("f32s")


def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in ((16384, 4096, 1024), (16384, 1024, 4096), (16384, 1024, 1024), (16384, 2048, 1024), (2560, 1024, 1024), (16384, 512, 2048)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
    ok = M % 256 == 0 and N % 256 == 0 and K % 32 == 0
    own = t(lambda: lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), M, N, K, st)) if ok else float("nan")
    libt = t(lambda: F._mm(x, w.t()))
    gemm_only = None
    xs, ws = F.split_bf16x3(x, 1, False), F.split_bf16x3(w, 1, True)
    gemm_only = t(lambda: torch.mm(xs, ws.t(), out_dtype=torch.float32))
    ref = x.double() @ w.double().t()
    if ok:
        lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), M, N, K, st); torch.cuda.synchronize()
        err = float((y.double() - ref).abs().max() / ref.abs().max())
    else:
        err = float("nan")
    err_lib = float((F._mm(x, w.t()).double() - ref).abs().max() / ref.abs().max())
    fl = 3 * 2.0 * M * N * K
    print(f"[{M} x {K}] x [{N} x {K}]^T: own {own:7.1f} us ({fl / own / 1e6:6.0f} TF/s bf16-equivalent)  passes + library {libt:7.1f} us  (library GEMM alone {gemm_only:7.1f} us = "
          f"{fl / gemm_only / 1e6:6.0f} TF/s)   rel err own {err:.2e} library path {err_lib:.2e}", flush=True)
