#!/usr/bin/env python3
"""Probe (developer tool): can the GMD train step be captured in HIP graphs (torch.cuda.CUDAGraph) -- the ctypes-launched
kernels, their hipMemsetAsync calls, autograd and the fused Adam -- and do events recorded DURING capture time the replayed
kernel nodes?  Prints eager vs replay ms/step, host enqueue time, the loss trajectories of both, and event timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import data, engine, functional as TF

B, T, N, d = 64, 128, 20, 1024
DROP = float(os.environ.get("PROBE_DROPOUT", "0.0"))
params = engine.default_params(video_rnn_hiddendim=d // 2, sent_rnn_hiddendim=d // 2, video_len=T, sent_len=N, dropout=DROP)
dev = torch.device("cuda", 0)


def build():
    torch.manual_seed(0)
    model = engine.build_model("gmd", params).to(dev).train()
    model.tod.dropout.p = DROP if DROP > 0 else 0.0
    ps = list(model.parameters())
    opt = torch.optim.Adam(ps, lr=params["lr"], weight_decay=params["weight_decay"], eps=1e-6, fused=True, capturable=True)
    return model, opt


batch = data.synthetic_batch(B, T, N, seed=1234, pair=True, device=dev)


def fwd_bwd(model):
    for p in model.parameters():
        p.grad = None
    with engine.precision("f32s"):
        loss, _, _ = engine.gmd_step(model, batch, params)
    loss.backward()
    return loss


def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, te / n * 1e3, out


# ---- eager
torch.manual_seed(1)
model, opt = build()
def eager():
    loss = fwd_bwd(model); engine.optimizer_step(opt, loss); return loss
for _ in range(3): eager()
ms, enq, _ = timeit(eager)
print(f"eager : {ms:.3f} ms/step, host enqueue {enq:.3f} ms")
torch.manual_seed(1)
model, opt = build()
le = [float(eager()) for _ in range(6)]

# ---- graphs: A = zero-grad + forward + backward, B = Adam
torch.manual_seed(1)
model, opt = build()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        loss = fwd_bwd(model); engine.optimizer_step(opt, loss)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
TF.check_lstm_errors()
torch.manual_seed(1)
model, opt = build()                      # fresh state, same as the eager run above
with torch.cuda.stream(s):                # one eager step to create the optimizer state, then undo nothing: compare from step 2
    pass
gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
TF.kernel_timer.enable(only=("tsg_scdm",))
with torch.cuda.graph(gA):
    loss_static = fwd_bwd(model)
TF.kernel_timer.disable()
with torch.cuda.graph(gB):
    engine.optimizer_step(opt, loss_static)
print("captured: graph A (fwd+bwd) and graph B (Adam)")
def replay():
    gA.replay(); gB.replay(); return loss_static
lg = []
for i in range(6):
    replay(); lg.append(float(loss_static))
    try:
        TF.check_lstm_errors()
    except TF.LstmWaitExpired as e:
        print(f"replay {i}: LSTM wait expired")
print("eager  losses:", [round(x, 5) for x in le])
print("graph  losses:", [round(x, 5) for x in lg])
ms, enq, _ = timeit(replay)
print(f"graphs: {ms:.3f} ms/step, host enqueue {enq:.3f} ms")
TF.check_lstm_errors()
try:
    print("events recorded during capture:", {k[0]: (round(v[0], 2), v[1]) for k, v in TF.kernel_timer.summary().items()})
except Exception as e:                    # noqa: BLE001
    print("event timing of captured nodes failed:", type(e).__name__, e)
