#!/usr/bin/env python3
"""Developer probe: inside ONE captured HIP graph, every workgroup reads a small buffer X (kernel 1), one workgroup overwrites X (kernel 2), every
workgroup reads X again (kernel 3).  Does kernel 3 ever see kernel 1's contents when the graph is replayed back to back (no host synchronisation)?
(Round 5: the key words of captured dropout launches showed up inside a gradient that reused their pool blocks -- HISTORY.md, round-5 status (11).)"""
import torch
dev = "cuda"
N = 1 << 22
x = torch.zeros(256, device=dev)                 # the small shared buffer (two 512-byte blocks)
src = torch.arange(256, device=dev, dtype=torch.float32)
acc1 = torch.zeros(N, device=dev); acc3 = torch.zeros(N, device=dev)
big = torch.ones(N, device=dev)
cnt = torch.zeros((), device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        x.copy_(src * cnt); acc1.copy_(big * x[1]); cnt += 1; x.copy_(src * cnt); acc3.copy_(big * x[1])
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    acc1.copy_(big * x[1])                       # kernel 1: every workgroup reads x[1] (old value)
    cnt += 1
    x.copy_(src * cnt)                           # kernel 2: one small launch overwrites x
    acc3.copy_(big * x[1])                       # kernel 3: every workgroup reads x[1] (must be the new value = cnt)
bad = 0
for it in range(2000):
    g.replay()
    if it % 50 == 49:
        torch.cuda.synchronize()
        want = float(cnt)
        nb = int((acc3 != want).sum())
        bad += nb
        if nb: print("replay", it, "stale elements", nb, "want", want, "saw", acc3[acc3 != want][:4].tolist(), flush=True)
print("stale reads in total:", bad)
