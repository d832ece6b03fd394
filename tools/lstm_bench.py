#!/usr/bin/env python3
"""Per-step time of the LSTM recurrence kernels (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr
TSG_F32 = int(os.environ.get('TSG_REC_DTYPE', '0'))     # 0 = fp32 MFMA recurrence, 2 = split-precision bf16 MFMA (TSG_F32S)
B, T, h = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 128, 512)))
lib = _lib.load(); dev = "cuda"; stream = torch.cuda.current_stream(); st = stream.cuda_stream
Gx = torch.randn(T, B, 2, 4 * h, device=dev) * 0.5; W = torch.randn(2, 4 * h, h, device=dev) / h ** 0.5
out = torch.empty(T, B, 2 * h, device=dev); R = torch.empty(T, 2, B, h, 4, device=dev); Cs = torch.empty(T, 2, B, h, device=dev)
dOut = torch.randn(T, B, 2 * h, device=dev); WT = W.transpose(1, 2).contiguous()
dG = torch.empty(T, B, 2, 4 * h, device=dev); dC = torch.empty(2, B, h, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n): fn()
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) / n
sync = torch.zeros(512, device=dev, dtype=torch.int32)
BM = int(os.environ.get('TSG_BM', '0'))      # 1: batch-major sequence tensors
if BM: Gx = Gx.transpose(0, 1).contiguous(); out = out.transpose(0, 1).contiguous(); dOut = dOut.transpose(0, 1).contiguous(); dG = dG.transpose(0, 1).contiguous()
if TSG_F32 == 1: Gx = Gx.bfloat16(); out = out.bfloat16(); R = R.bfloat16(); dOut = dOut.bfloat16(); dG = dG.bfloat16()
nfw = lib.tsg_lstm_fwd_ws_bytes(B, T, h)
if nfw > 0 and os.environ.get('TSG_RING', '1') != '0':        # the exchange-ring forward (default); TSG_RING=0: tsg_lstm_fwd_bias (polls `out`)
    sync = torch.zeros(nfw // 4, device=dev, dtype=torch.int32)
    f = t(lambda: lib.tsg_lstm_fwd_ws(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), nfw, B, T, h, TSG_F32, BM, st))
else:
    f = t(lambda: lib.tsg_lstm_fwd_bias(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, TSG_F32, BM, st))
sys.stdout.flush(); print('sync word0 (error flag):', int(sync[0]), 'arrived', int(sync[1]), 'workgroups on the L2-local exchange', int(sync[3]), 'phase ticks', sync[8:12].tolist())
b = t(lambda: lib.tsg_lstm_bwd(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), B, T, h, TSG_F32, st)) if TSG_F32 != 1 and os.environ.get('TSG_STEPK', '1') != '0' else 0.0   # (the step kernels are fp32-storage only)
nb = lib.tsg_lstm_bwd_ws_bytes(B, T, h)
if nb > 0:
    ws = torch.empty(nb // 4 + 4, device=dev)
    b2 = t(lambda: lib.tsg_lstm_bwd_ws_layout(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), ptr(ws), nb, None, B, T, h, TSG_F32, BM, st))
    print(f"persistent backward (ring exchange): {b2*1e3/T:.2f} us/step ({b2:.2f} ms), err word {int(ws[:1].view(torch.int32)[0])}, L2-local workgroups {int(ws[3:4].view(torch.int32)[0])}, phase ticks [poll, reduce+cell, gather barrier, stores, Dl barrier, MFMA] {ws[8:14].view(torch.int32).tolist()}")
sys.stdout.flush(); print(f"rec dtype {TSG_F32} batch-major {BM} B={B} T={T} h={h}: fwd {f*1e3/T:.2f} us/step ({f:.2f} ms), bwd {b*1e3/T:.2f} us/step ({b:.2f} ms)")
