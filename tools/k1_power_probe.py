#!/usr/bin/env python3
"""Is the K1g forward power-bound?  Same launch on random operands and on all-zero operands (zero operands toggle fewer bits: a chip that
holds its clock down under load clocks higher on them; a kernel bound by issue slots or bandwidth takes the same time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr
B, T, N, d = 128, 128, 20, 1024
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
def bench(zero, dt=2, n=300):
    mk = (lambda *s: torch.zeros(*s, device="cuda")) if zero else (lambda *s: torch.randn(*s, device="cuda"))
    A, S, VW, r = mk(B, T, d), mk(B, N, d), mk(B, N, d), mk(B, T, d)
    w = mk(d) / 32; gb = mk(d) * 0.1
    out = torch.empty(B, T, d, device="cuda"); P = torch.empty(B, T, N, device="cuda")
    fn = lambda: lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, dt, st)
    for _ in range(200): assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    print(f"random operands {bench(False):6.1f} us   zero operands {bench(True):6.1f} us", flush=True)
