#!/usr/bin/env python3
"""Profiling target: n launches of tsg_wgrad_f32s at the W_a weight-gradient shape.   python tools/wgrad_only.py [n] [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (16384, 1024, 1024)
A, B = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
for _ in range(n):
    C = F.wgrad_f32s(A, B)
torch.cuda.synchronize()
print(float(C.abs().max()))
