import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from oracle import tsg_oracle as O
from test_lstm_gpu import _params
print("cpus", os.cpu_count(), "default threads", torch.get_num_threads())
B, T, I, h = 3, 512, 1024, 512
for nt in (0, 4, 8, 16, 32):
    if nt: torch.set_num_threads(nt)
    g = torch.Generator().manual_seed(13)
    p = {k: v.requires_grad_(True) for k, v in _params(I, h, 2, g).items()}
    x = torch.randn(B, T, I, generator=g).requires_grad_(True)
    t = time.time()
    out0, hn0, cn0 = O.bilstm(x, p, 2)
    out0.sum().backward()
    print("threads", nt or "default", round(time.time() - t, 1), "s", flush=True)
