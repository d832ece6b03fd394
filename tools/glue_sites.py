#!/usr/bin/env python3
"""Developer tool: which Python call sites launch the small torch kernels (copies, adds, cats, fills, reductions) of one GMD train
step?  torch.profiler with stacks; prints device time and launches per step grouped by (op, innermost repo frame)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from shufflingvideosfortsg_amd import data, engine

B, T, N, d = 64, 128, 20, 1024
params = engine.default_params(video_rnn_hiddendim=d // 2, sent_rnn_hiddendim=d // 2, video_len=T, sent_len=N)
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
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True, with_modules=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
OPS = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul", "aten::contiguous",
       "aten::clone", "aten::zeros", "aten::zeros_like", "aten::stack", "aten::native_dropout", "aten::native_dropout_backward", "aten::index_select",
       "aten::embedding", "aten::embedding_dense_backward", "aten::sub", "aten::div", "aten::neg", "aten::sigmoid", "aten::tanh")
acc = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.name not in OPS:
        continue
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if not dt:
        continue
    site = "?"
    for fr in (e.stack or []):
        if "/repo/" in fr and "tools/" not in fr:
            site = fr.split("/repo/")[-1]
            break
    if site == "?":                                       # no Python stack (backward thread): the enclosing autograd node / module
        par, chain = e.cpu_parent, []
        while par is not None and len(chain) < 3:
            if par.name.startswith("autograd::engine::evaluate_function") or par.name.startswith("nn.Module") or "Backward" in par.name:
                chain.append(par.name.replace("autograd::engine::evaluate_function: ", ""))
            par = par.cpu_parent
        site = " < ".join(chain) if chain else "?"
    k = (e.name, site, str(e.input_shapes)[:60])
    acc[k][0] += e.self_device_time_total if hasattr(e, "self_device_time_total") else dt
    acc[k][1] += 1
rows = sorted(((v[0] / STEPS, v[1] / STEPS, k) for k, v in acc.items()), reverse=True)
print(f"# {sum(r[1] for r in rows):.0f} small ops per step, {sum(r[0] for r in rows):.0f} us/step")
for us, n, (op, site, shp) in rows[:110]:
    print(f"{us:8.1f} us x{n:5.1f}  {op:28s} {site:70s} {shp}")
