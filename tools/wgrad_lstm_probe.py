#!/usr/bin/env python3
"""Developer tool: where the LSTM-layer weight gradient ([2][2048 x 16384] x [16384 x 1536]) loses against the square product per chunk:
the same kernel with the layer's features switched off one at a time.   python tools/wgrad_lstm_probe.py [case]   (case: profiling target)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


TB, I, h, T = 16384, 1024, 512, 128
dG, x, out = torch.randn(TB, 8 * h, device="cuda"), torch.randn(TB, I, device="cuda"), torch.randn(TB, 2 * h, device="cuda")
dG1 = dG[:, :4 * h].contiguous()
xo = torch.randn(TB, I + h, device="cuda")
cases = {
    "layer: 2 groups, lda 4096, B1 shifted (period 128)": lambda: F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=1, period=T),
    "layer, shift by rows only (period 0)": lambda: F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=1, period=0),
    "layer, no shift": lambda: F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=0, period=0),
    "2 groups, one B segment of 1536 columns": lambda: F.wgrad_f32s(dG, xo, N=4 * h, groups=2, a_group_stride=4 * h),
    "1 group of 4096 rows of C, one B segment": lambda: F.wgrad_f32s(dG, xo),
    "1 group, A contiguous [16384 x 2048], one B segment": lambda: F.wgrad_f32s(dG1, xo),
    "1 group, A contiguous [16384 x 2048] x [16384 x 1024]": lambda: F.wgrad_f32s(dG1, x),
    "square [1024 x 16384] x [16384 x 1024]": lambda: F.wgrad_f32s(x, x),
}
if len(sys.argv) > 1:
    fn = list(cases.values())[int(sys.argv[1])]
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    sys.exit(0)
first = next(iter(cases))
cases["(again) " + first] = cases[first]                # the first case is timed while the clock is still settling: repeat it last
for _ in range(40):
    cases[first]()
for name, fn in cases.items():
    C = fn()
    fl = 3 * 2.0 * TB * C.shape[0] * C.shape[1] * C.shape[2]
    t = timeit(fn)
    print(f"{name:60s} {t:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s of bf16 work   C {tuple(C.shape)}")
