#!/usr/bin/env python3
"""Developer tool: the CPU oracle's speed against torch's intra-op thread count on this host (its recurrences are thousands of small ops):
the 2-layer BiLSTM at [3, 512, 1024] (what the T = 512 parity tests run) and bench.py's cpu_baseline step (GMD, 16 pairs, T=128, d=1024)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import tsg_oracle as O
from shufflingvideosfortsg_amd import data, engine
from test_lstm_gpu import _params
print("cpus", os.cpu_count(), "default threads", torch.get_num_threads())
params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, video_len=128, sent_len=20)
torch.manual_seed(0)
model = engine.build_model("gmd", params)
sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}


def gmd_step(B):
    b = data.synthetic_batch(B, 128, 20, pair=True)
    g, pg = b["gt"], b["pseudo_gt"]
    out = O.gmd_forward(sd, b["query"], b["video"], b["video_mask"], b["pseudo_video"], b["video_mask"], g["temporal_labels"], g["fore_masks"],
                        g["back_masks"], pg["temporal_labels"], pg["fore_masks"], pg["back_masks"])
    loss, _ = O.gmd_losses(out, b["video_mask"], b["video_mask"], g, pg)
    loss.backward()


for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    g = torch.Generator().manual_seed(13)
    p = {k: v.requires_grad_(True) for k, v in _params(1024, 512, 2, g).items()}
    x = torch.randn(3, 512, 1024, generator=g).requires_grad_(True)
    t = time.time(); out0, _, _ = O.bilstm(x, p, 2); out0.sum().backward(); t1 = time.time() - t
    gmd_step(2)
    t = time.time(); gmd_step(16); t2 = time.time() - t
    print(f"threads {nt:3d}: BiLSTM [3,512,1024] fwd+bwd {t1:6.1f} s   GMD step of 16 pairs {t2:6.1f} s = {16 / t2:.2f} pairs/s", flush=True)
