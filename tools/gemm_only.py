#!/usr/bin/env python3
"""Launch tsg_linear_fwd a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
M, N, K = 16384, 1024, 1024
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
for _ in range(4):
    lib.tsg_linear_fwd(ptr(x), ptr(w), None, ptr(y), M, N, K, TSG_F32, st)
    torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
