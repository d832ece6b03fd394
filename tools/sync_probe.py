#!/usr/bin/env python3
"""Report host<->device synchronisation points inside one train step (torch.cuda.set_sync_debug_mode) -- developer probe."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine
from shufflingvideosfortsg_amd.dp import FlatGradAllReduce
dev = torch.device("cuda", 0)
params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, video_len=128, sent_len=20)
model = engine.build_model("gmd", params).to(dev).train()
dp = FlatGradAllReduce(model); opt = engine.make_optimizer(model, params)
batch = data.synthetic_batch(16, 128, 20, seed=1, pair=True, device=dev)
def step():
    dp.zero_grad()
    with engine.precision("f32s"):
        loss, _, _ = engine.gmd_step(model, batch, params)
    loss.backward(); dp.finish(); opt.step()
step(); torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
print(f"{len(w)} synchronising calls in one step")
import traceback
for x in w[:20]:
    print(x.filename, x.lineno, str(x.message)[:100])
