#!/usr/bin/env python3
"""Replayed gradients vs an eager step and vs each other (dropout off, lr = 0: the parameters never move).  python tools/graph_grad_probe.py [mode] [replays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine, functional as TF
mode = sys.argv[1] if len(sys.argv) > 1 else "f32s"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, video_len=128, sent_len=20); params["dropout"] = 0.0; params["lr"] = 0.0
torch.manual_seed(0)
model = engine.build_model("gmd", params).cuda().train()
for m_ in model.modules():
    if isinstance(m_, torch.nn.Dropout): m_.p = 0.0           # (the discriminator's dropout is not governed by params["dropout"])
batch = data.synthetic_batch(64, 128, 20, seed=1234, pair=True, device="cuda")
engine.set_precision(mode)
step = lambda m, b: engine.gmd_step(m, b, params)[0]
def eager():
    model.zero_grad(set_to_none=True); step(model, batch).backward(); torch.cuda.synchronize()
    return {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
e1, e2 = eager(), eager()
gall = max(float(v.abs().max()) for v in e1.values())
def worst(a, b):
    out = []
    for k in a:
        d = float((a[k] - b[k]).abs().max()); s = float(b[k].abs().max())
        out.append((d / (s + 1e-30), d, s, k))
    return sorted(out, reverse=True)[:4]
print("global max |g| =", gall)
print("eager vs eager:", worst(e1, e2))
model.zero_grad(set_to_none=True)
opt = engine.make_optimizer(model, params, capturable=True)
g = engine.GraphedTrainStep(model, opt, step, batch, warmup=3)
snap = lambda: {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if k in e1}
g(); torch.cuda.synchronize(); r1 = snap()
print("replay 1 vs eager:", worst(r1, e1)[:2])
burst = int(os.environ.get("BURST", 10))
bad = 0
for it in range(n):
    for _ in range(burst):
        g()                                            # back to back: no host synchronisation in between
    torch.cuda.synchronize(); r = snap()
    w = worst(r, e1)
    if w[0][0] > 1e-4: bad += 1; print(f"after burst {it} ({burst} replays):", [(round(x[0], 3), x[3]) for x in w[:3]])
print(f"TSG_GRAPH_PROBE={os.environ.get('TSG_GRAPH_PROBE', '')!r} mode={mode}: {bad} of {n} bursts of {burst} back-to-back replays ended with a gradient off by more than 1e-4 of its max")
