#!/bin/bash
# K1g forward: timing-only ablation (tools/_ablate/k1abl.so) + an elementwise 2-read/1-write torch kernel of the same byte count as an HBM reference
O=gpurun_out/$1; mkdir -p $O
python tools/k1_mm_ablate.py 128 > $O/ablate.txt 2>&1
python - >> $O/ablate.txt 2>&1 <<'PY'
import torch
B,T,d=128,128,1024
a=torch.randn(B,T,d,device="cuda"); r=torch.randn(B,T,d,device="cuda"); o=torch.empty_like(a)
for name,fn,nb in (("mul(a,r)->o 201MB", lambda: torch.mul(a,r,out=o), 3*a.numel()*4), ("copy a->o 134MB", lambda: o.copy_(a), 2*a.numel()*4)):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/200*1e3
    print(f"{name}: {us:.1f} us = {nb/us/1e6:.2f} TB/s")
PY
cat $O/ablate.txt
