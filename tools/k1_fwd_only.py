#!/usr/bin/env python3
"""Launch the gate-fused K1 forward a few times (for rocprofv3 --pmc): python3 tools/k1_fwd_only.py [B] [n] [dtype: 0 f32, 1 bf16, 2 f32s]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T, N, d = 128, 20, 1024
cast = torch.bfloat16 if dt == 1 else torch.float32
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
A = torch.randn(B, T, d, device=dev).to(cast); S = torch.randn(B, N, d, device=dev).to(cast); w = torch.randn(d, device=dev) / d ** 0.5
P = torch.empty(B, T, N, device=dev); VW = torch.randn(B, N, d, device=dev).to(cast); gb = torch.randn(d, device=dev)
r = torch.randn(B, T, d, device=dev).to(cast); out = torch.empty(B, T, d, device=dev, dtype=cast)
for _ in range(n):
    rc = lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, dt, st)
    assert rc == 0, lib.tsg_last_error()
torch.cuda.synchronize()
