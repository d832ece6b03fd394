#!/usr/bin/env python3
"""Persistent vs launch-per-step LSTM backward: run this script twice (TSG_LSTM_PERSIST=0 / 1) -- it writes / compares a dump.
   python tools/lstm_bwd_check.py B T h"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, h = (int(x) for x in sys.argv[1:4])
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
Gx = (torch.randn(T, B, 2, 4 * h, generator=g) * 0.5).cuda(); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
dOut = torch.randn(T, B, 2 * h, generator=g).cuda(); dHn = torch.randn(2, B, h, generator=g).cuda()
out = torch.empty(T, B, 2 * h, device="cuda"); R = torch.empty(T, 2, B, h, 4, device="cuda"); Cs = torch.empty(T, 2, B, h, device="cuda")
assert lib.tsg_lstm_fwd(ptr(Gx), ptr(W), ptr(out), ptr(R), ptr(Cs), None, B, T, h, TSG_F32, st) == 0
WT = W.transpose(1, 2).contiguous()
res = []
for hn in (None, dHn):
    dG = torch.full((T, B, 2, 4 * h), 3.0, device="cuda"); dC = torch.zeros(2, B, h, device="cuda")
    if os.environ.get("TSG_WS") == "1":
        nb = lib.tsg_lstm_bwd_ws_bytes(B, T, h); assert nb > 0, "no persistent backward for this shape"
        ws = torch.empty(nb // 4 + 4, device="cuda")
        rc = lib.tsg_lstm_bwd_ws(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), ptr(hn) if hn is not None else None, ptr(dG), ptr(dC), ptr(ws), nb, None, B, T, h, TSG_F32, st)
        torch.cuda.synchronize(); print("ws err word", int(ws[:1].view(torch.int32)[0]))
    else:
        rc = lib.tsg_lstm_bwd(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), ptr(hn) if hn is not None else None, ptr(dG), ptr(dC), B, T, h, TSG_F32, st)
    torch.cuda.synchronize(); assert rc == 0, lib.tsg_last_error()
    res.append(dG.cpu())
f = f"/tmp/lstm_bwd_{B}_{T}_{h}.pt"
mode = os.environ.get("TSG_LSTM_PERSIST", "auto")
if mode == "0" and os.environ.get("TSG_WS") != "1":
    torch.save(res, f); print("reference written", f)
else:
    ref = torch.load(f)
    for a, b, tag in zip(res, ref, ("no dHn", "with dHn")):
        d = (a - b).abs()
        print(f"persist={mode} B={B} T={T} h={h} {tag}: max diff {d.max().item():.3e}  nan {int(torch.isnan(a).sum())}  max|ref| {b.abs().max().item():.3e}")
