#!/usr/bin/env python3
"""Which allocations of the capture shared the pool block that ends up as a given parameter's gradient?  (torch.cuda.memory history of the
GraphedTrainStep capture; the innermost package / torch frames of every allocation that overlapped the block.)
    python tools/graph_block_history.py [mode] [parameter name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine
mode = sys.argv[1] if len(sys.argv) > 1 else "f32s"
K = sys.argv[2] if len(sys.argv) > 2 else "sentence_encoder.word_embed.bias"
params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, video_len=128, sent_len=20); params["dropout"] = 0.0; params["lr"] = 0.0
torch.manual_seed(0)
model = engine.build_model("gmd", params).cuda().train()
for m_ in model.modules():
    if isinstance(m_, torch.nn.Dropout): m_.p = 0.0
batch = data.synthetic_batch(64, 128, 20, seed=1234, pair=True, device="cuda")
engine.set_precision(mode)
step = lambda m, b: engine.gmd_step(m, b, params)[0]
opt = engine.make_optimizer(model, params, capturable=True)
torch.cuda.memory._record_memory_history(max_entries=400000, stacks="python")
g = engine.GraphedTrainStep(model, opt, step, batch, warmup=3)
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
pk = dict(model.named_parameters())[K]
lo = pk.grad.data_ptr(); hi = lo + pk.grad.numel() * 4
print(f"{K}.grad at {hex(lo)}..{hex(hi)}")
ev = [e for tr in snap["device_traces"] for e in tr]
hits = [e for e in ev if e.get("action") in ("alloc", "free_completed", "free_requested") and e["addr"] < hi and e["addr"] + e["size"] > lo]
print(len(ev), "events,", len(hits), "touch the block")
def where(e):
    fr = [f for f in e.get("frames", []) if "shufflingvideosfortsg_amd" in f["filename"] or "/torch/nn/" in f["filename"] or "torch/autograd" in f["filename"]]
    return " <- ".join(f"{os.path.basename(f['filename'])}:{f['line']}({f['name']})" for f in fr[:5])
for e in hits[-40:]:
    print(f"{e['action']:15s} {hex(e['addr'])} +{e['size']:7d}  {where(e)}")
