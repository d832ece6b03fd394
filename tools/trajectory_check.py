#!/usr/bin/env python3
"""Loss trajectory of the GMD train step in the three precision modes from the same initial state and data: strict f32,
split-precision f32s (headline), bf16 operands.  Evidence that f32s trains like f32 (developer tool)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine
from shufflingvideosfortsg_amd.dp import FlatGradAllReduce
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
params = engine.default_params(video_rnn_hiddendim=d // 2, sent_rnn_hiddendim=d // 2, video_len=64, sent_len=20, dropout=0.0)
torch.manual_seed(0)
base = engine.build_model("gmd", params).to(dev).train()
batches = [data.synthetic_batch(16, 64, 20, seed=100 + i, pair=True, device=dev) for i in range(4)]
curves = {}
for name, mode in [("f32", None), ("f32+1ulp", None), ("f32s", "f32s"), ("bf16", torch.bfloat16)]:
    model = copy.deepcopy(base)
    if name == "f32+1ulp":                    # the sensitivity yardstick: strict f32 from parameters perturbed by ~1 ulp (6e-8 relative)
        g = torch.Generator(device=dev).manual_seed(7)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.0 + 6e-8 * torch.randn(p.shape, device=dev, generator=g))
    torch.manual_seed(1)                      # same dropout masks in every mode (the discriminator's p = 0.5 is hard-wired)
    dp = FlatGradAllReduce(model); opt = engine.make_optimizer(model, params)
    losses = []
    for it in range(steps):
        dp.zero_grad()
        with engine.precision(mode):
            loss, _, _ = engine.gmd_step(model, batches[it % 4], params)
        loss.backward(); dp.finish(); opt.step()
        losses.append(float(loss))
    engine.set_precision(None)
    curves[name] = losses
ref = curves["f32"]
for name in ("f32+1ulp", "f32s", "bf16"):
    rel = [abs(a - b) / abs(b) for a, b in zip(curves[name], ref)]
    print(f"{name}: max relative loss deviation from f32 over {steps} steps {max(rel):.2e} (step 1: {rel[0]:.2e}, last: {rel[-1]:.2e})")
print("f32 loss: first %.5f last %.5f" % (ref[0], ref[-1]))
