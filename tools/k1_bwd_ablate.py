#!/usr/bin/env python3
"""Timing-only ablation of the gate-fused K1 backward (needs tools/_ablate/k1abl.so = a -DTSG_ABLATE build of scdm_attn.hip):
TSG_ABLATE_MASK bit 16 skips the row phase, 32 the column phase, 64 the wait on the partner workgroups, 256 the dw / dbias atomics,
512 every workgroup barrier inside the phases + the exchange, 1024 (with 512) waves 4..7 run the column phase before the row phase
(the overlap experiment); TSG_ABL_MASKS=0,512,1536 selects the masks.
python tools/k1_bwd_ablate.py [B]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "128"
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, N, d = int(sys.argv[1]), 128, 20, 1024
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev); w = torch.randn(d, device=dev) / d ** 0.5
P = torch.softmax(torch.randn(B, T, N, device=dev), -1); VW = torch.randn(B, N, d, device=dev); gb = torch.randn(d, device=dev)
r = torch.randn(B, T, d, device=dev); dC = torch.randn(B, T, d, device=dev)
da = torch.empty_like(A); ds = torch.empty_like(S); dw = torch.empty_like(w); dVW = torch.empty_like(VW); dgb = torch.empty_like(gb)
dr = torch.empty_like(r); nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, 1)); ws = torch.empty(nb // 4 + 4, device=dev)
fn = lambda: lib.tsg_scdm_gate_bwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dVW), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, d, d, TSG_F32, st)
for _ in range(60): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): fn()
e1.record(); torch.cuda.synchronize()
print("mask", os.environ.get("TSG_ABLATE_MASK", "0"), "%%.1f us" %% (e0.elapsed_time(e1) / 100 * 1e3), flush=True)
''' % root
for mask in (os.environ.get("TSG_ABL_MASKS", "0,16,32,48,64,112,368,0").split(",")):
    env = dict(os.environ, TSG_HIP_LIB=os.path.join(root, "tools", "_ablate", "k1abl.so"), TSG_ABLATE_MASK=mask)
    subprocess.run([sys.executable, "-c", code, B], env=env)
