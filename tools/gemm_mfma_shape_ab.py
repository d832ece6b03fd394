#!/usr/bin/env python3
"""tsg_gemm_f32s at the step's GEMM shapes, timed and checked against float64; run once per MFMA shape:
    python tools/gemm_mfma_shape_ab.py            (v_mfma_f32_32x32x16_bf16, the shipped kernel)
    TSG_GEMM_MFMA16=1 python tools/gemm_mfma_shape_ab.py      (the v_mfma_f32_16x16x32_bf16 variant)
LIB=1 adds the library's bf16 GEMM alone on pre-split operand planes (what the LSTM-side products run on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib, functional as F
from shufflingvideosfortsg_amd._lib import ptr
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
tag = "16x16x32" if os.environ.get("TSG_GEMM_MFMA16") == "1" else "32x32x16"


def t(fn, n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rnd in range(2):
    for (M, N, K) in ((16384, 4096, 1024), (16384, 1024, 4096), (16384, 1024, 1024), (16384, 2048, 1024), (8192, 1024, 1024)):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
        own = t(lambda: lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), M, N, K, st))
        fl = 3 * 2.0 * M * N * K
        line = f"[{tag}] [{M} x {K}] x [{N} x {K}]^T: {own:7.1f} us = {fl / own / 1e6:6.0f} TF/s bf16-equivalent"
        if rnd == 0:
            ref = x[:512].double() @ w.double().t()
            err = float((y[:512].double() - ref).abs().max() / ref.abs().max())
            line += f"   rel err {err:.2e}"
            if os.environ.get("LIB") == "1":
                xs, ws = F.split_bf16x3(x, 1, False), F.split_bf16x3(w, 1, True)
                g = t(lambda: torch.mm(xs, ws.t(), out_dtype=torch.float32))
                line += f"   library bf16 GEMM alone on planes {g:7.1f} us = {fl / g / 1e6:6.0f} TF/s"
        print(line, flush=True)
