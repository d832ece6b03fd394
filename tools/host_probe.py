#!/usr/bin/env python3
"""Where does the host spend a train step?  Host-side timestamps per phase (no syncs inside), for two batch sizes: phases
whose host time grows with the batch are phases where the host waits for the GPU (developer probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine
from shufflingvideosfortsg_amd.dp import FlatGradAllReduce
dev = torch.device("cuda", 0)
params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, video_len=128, sent_len=20)
torch.manual_seed(0)
model = engine.build_model("gmd", params).to(dev).train()
dp = FlatGradAllReduce(model); opt = engine.make_optimizer(model, params)
for B in (8, 64):
    batch = data.synthetic_batch(B, 128, 20, seed=1, pair=True, device=dev)
    acc = [0.0] * 5
    for it in range(13):
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        dp.zero_grad(); t.append(time.perf_counter())
        with engine.precision("f32s"):
            loss, _, _ = engine.gmd_step(model, batch, params)
        t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        dp.finish(); opt.step(); t.append(time.perf_counter())
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if it >= 3:
            for i in range(5): acc[i] += (t[i + 1] - t[i]) / 10 * 1e3
    print(f"B={B}: host ms  zero_grad {acc[0]:.2f}  forward+loss {acc[1]:.2f}  backward {acc[2]:.2f}  finish+adam {acc[3]:.2f}  final sync wait {acc[4]:.2f}  total {sum(acc):.2f}")
