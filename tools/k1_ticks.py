#!/usr/bin/env python3
"""Where the waves of the K1g forward kernel spend their time: needs a -DTSG_K1_TICKS build (tools/build_variant.sh, TSG_HIP_LIB).
    TSG_HIP_LIB=tools/_ablate/k1ticks.so python tools/k1_ticks.py [B] [gate 0/1] [dtype]
Prints per role (producer waves 0..7, consumer waves 8..15) the mean shader-clock ticks per phase over all workgroups of the last launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
gate = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T, N, d = 128, 20, 1024
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH); st = torch.cuda.current_stream().cuda_stream
cast = torch.bfloat16 if dt == 1 else torch.float32
A = torch.randn(B, T, d, device="cuda").to(cast); S = torch.randn(B, N, d, device="cuda").to(cast); w = torch.randn(d, device="cuda") / 32
VW = torch.randn(B, N, d, device="cuda").to(cast); gb = torch.randn(d, device="cuda") * 0.1; r = torch.randn(B, T, d, device="cuda").to(cast)
out = torch.empty_like(A); P = torch.empty(B, T, N, device="cuda")
def run():
    if gate: return lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, dt, st)
    return lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(out), ptr(P), B, T, N, d, d, dt, st)
for _ in range(50): assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print(f"B={B} gate={gate} dtype={dt}: {e0.elapsed_time(e1) * 10:.1f} us per launch (instrumented build)")
buf = np.zeros(256 * 16 * 8, dtype=np.uint64)
assert raw.tsg_debug_k1_ticks(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(256, 16, 8).astype(np.float64)
names = {"producer": ["prologue", "score loop", "wait next row", "reduce+softmax+P+hand-over", "claim next row"],
         "consumer": ["prologue+VW", "wait for the producers", "wait r rows", "phase 2 + stores", "issue r loads"]}
for role, sl in (("producer", slice(0, 8)), ("consumer", slice(8, 16))):
    m = t[:, sl, :].mean(axis=(0, 1)); tot = m.sum()
    print(f"{role}: total {tot:.0f} ticks")
    for i, nm in enumerate(names[role]):
        print(f"   {nm:28s} {m[i]:9.0f}  {100 * m[i] / tot:5.1f}%   (min wave {t[:, sl, i].min():.0f}, max {t[:, sl, i].max():.0f})")

# per-wave view of a few workgroups: HW_ID bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13] ... (gfx9 layout)
raw_t = buf.reshape(256, 16, 8)
for blk in (0, 1, 100):
    print(f"workgroup {blk}: wave: simd  score/phase2 ticks  wait ticks")
    for wv_ in range(16):
        hw = int(raw_t[blk, wv_, 7]); simd = (hw >> 4) & 3; slot = hw & 15; cu = (hw >> 8) & 15
        main = t[blk, wv_, 1] if wv_ < 8 else t[blk, wv_, 3]
        bar = t[blk, wv_, 2] if wv_ < 8 else t[blk, wv_, 1]
        print(f"   wave {wv_:2d}: simd {simd} slot {slot:2d} cu {cu:2d}   {main:8.0f}  {bar:8.0f}")
