#!/usr/bin/env python3
"""Developer tool: tsg_gemm_f32s with 256 / 128 / 64-row tiles at the sentence-side shapes against the operand passes + library GEMM.
python tools/gemm_small_time.py   (TSG_GEMM_TM=256|128|64 forces the tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib, functional as F
from shufflingvideosfortsg_amd._lib import ptr
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
F.set_gemm_dtype("f32s")


def t(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in ((2560, 1024, 1024), (1280, 4096, 1024), (1280, 1024, 4096), (2560, 2048, 1024), (8192, 1024, 1024), (1280, 1024, 1024)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
    own = t(lambda: lib.tsg_gemm_f32s(ptr(x), ptr(w), None, ptr(y), M, N, K, st))
    libt = t(lambda: torch.mm(F._split_operand(x, 1, False), F._split_operand(w.t(), 0, True), out_dtype=torch.float32))
    print(f"TM={os.environ.get('TSG_GEMM_TM', 'auto'):4s} [{M} x {K}] x [{N} x {K}]^T: own {own:7.1f} us   passes + library {libt:7.1f} us", flush=True)
