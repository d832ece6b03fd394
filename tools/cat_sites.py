#!/usr/bin/env python3
"""Developer tool: which call sites of the package call torch.cat / torch.stack / dropout / .contiguous() in one GMD train step (forward)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine

acc = collections.Counter()


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        st = traceback.extract_stack(limit=8)[:-1]
        site = "?"
        for fr in reversed(st):
            if "/shufflingvideosfortsg_amd/" in fr.filename:
                site = f"{fr.filename.split('/shufflingvideosfortsg_amd/')[-1]}:{fr.lineno}"
                break
        out = orig(*a, **k)
        shp = tuple(out.shape) if isinstance(out, torch.Tensor) else "-"
        acc[(name, site, str(shp))] += 1
        return out
    setattr(mod, name, f)


for n in ("cat", "stack", "dropout"):
    wrap(torch, n)
wrap(torch.nn.functional, "dropout")
B, T, N, d = 64, 128, 20, 1024
params = engine.default_params(video_rnn_hiddendim=d // 2, sent_rnn_hiddendim=d // 2, video_len=T, sent_len=N)
dev = torch.device("cuda", 0)
model = engine.build_model("gmd", params).to(dev).train()
batch = data.synthetic_batch(B, T, N, seed=1234, pair=True, device=dev)
engine.set_precision("f32s")
loss, _, _ = engine.gmd_step(model, batch, params)
acc.clear()
loss, _, _ = engine.gmd_step(model, batch, params)
loss.backward()
for (name, site, shp), n in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:8s} {site:60s} {shp}")
