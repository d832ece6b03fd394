#!/usr/bin/env python3
"""Developer tool: time the W_a weight-gradient shape with the library named by TSG_HIP_LIB (ablation builds; results are wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F
M, N, K = 16384, 1024, 1024
A, B = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
F.kernel_timer.enable(only=("tsg_wgrad",))
for _ in range(20):
    F.wgrad_f32s(A, B)
torch.cuda.synchronize()
for k, v in F.kernel_timer.summary().items():
    print(os.environ.get("TSG_HIP_LIB", "default"), k[0], f"median {v[2]:.1f} us")
