#!/usr/bin/env python3
"""Developer tool: which library GEMMs does one GMD train step launch, with which shapes, and what do they cost?
Runs the bench step under torch.profiler (record_shapes) and prints aten::mm / addmm / bmm / matmul grouped by input
shapes + dtype with their device time per step.   python tools/gemm_shapes.py [B T N d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from shufflingvideosfortsg_amd import data, engine

B, T, N, d = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 128, 20, 1024)
params = engine.default_params(video_rnn_hiddendim=d // 2, sent_rnn_hiddendim=d // 2, video_len=T, sent_len=N, dropout=0.0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = engine.build_model("gmd", params).to(dev).train()
opt = engine.make_optimizer(model, params)
batch = data.synthetic_batch(B, T, N, seed=1234, pair=True, device=dev)


def step():
    for p in model.parameters():
        p.grad = None
    with engine.precision(os.environ.get("MODE", "f32s")):
        loss, _, _ = engine.gmd_step(model, batch, params)
    loss.backward()
    engine.optimizer_step(opt, loss)


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::_scaled_mm"):
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = e.cuda_time_total
        rows.append((dt / STEPS, e.count / STEPS, e.key, str(e.input_shapes)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# library GEMM calls per step: {sum(r[1] for r in rows):.0f}, device time {tot:.0f} us/step")
for us, n, k, shp in rows:
    print(f"{us:9.1f} us  x{n:4.1f}  {k:12s} {shp}")
