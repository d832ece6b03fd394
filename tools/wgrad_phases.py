#!/usr/bin/env python3
"""Developer tool: per-phase s_memtime cycles of the wgrad kernel (needs a -DTSG_WGRAD_TIMING build:
   TSG_VARIANT_SRC=wgrad_split tools/build_variant.sh wgradtime -DTSG_WGRAD_TIMING; TSG_HIP_LIB=tools/_ablate/wgradtime.so python tools/wgrad_phases.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 1024, 1024)
A, B = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
for _ in range(3):
    C = F.wgrad_f32s(A, B)
torch.cuda.synchronize()
v = C[0, 0, :32].cpu().view(8, 4)
print("cycles per chunk: wave | request  compute  stage  barrier | sum")
for w in range(8):
    print(f"  wave {w}: " + "  ".join(f"{float(x):8.0f}" for x in v[w]) + f" | {float(v[w].sum()):8.0f}")
