"""Time K3 (boundary head) forward / backward alone with HIP events: python tools/k3_time.py [B T Hm]"""
import sys
import torch
sys.path.insert(0, ".")
from shufflingvideosfortsg_amd import functional as F

B, T, Hm = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 128, 256)
J = 2 * Hm
g = torch.Generator().manual_seed(0)
y = torch.randn(B, T, J, generator=g).cuda().requires_grad_(True)
cs = torch.randn(B, J, generator=g).cuda().requires_grad_(True)
b1 = torch.zeros(J).cuda().requires_grad_(True); w2 = (torch.randn(J, generator=g) / Hm ** .5).cuda().requires_grad_(True)
b2 = torch.zeros(2).cuda().requires_grad_(True)
gate = torch.randn(B, T, generator=g).cuda().requires_grad_(True)
mask = torch.ones(B, T, dtype=torch.int32).cuda()
gs = torch.randn(B, T, generator=g).cuda()
F.kernel_timer.enable(only=("tsg_boundary",))
big = torch.empty(64 << 20, device="cuda")
for dirty in (False, True):
    F.kernel_timer.enable(only=("tsg_boundary",))
    for it in range(60):
        if dirty:
            big.normal_()                      # leave the L2s full of someone else's dirty lines
        s, e = F.boundary_score(y, cs, b1, w2, b2, gate, mask)
        ((s + e) * gs).sum().backward()
    torch.cuda.synchronize()
    print("dirty L2" if dirty else "clean", {k[0]: (round(v[0], 2), v[1], round(v[2], 2)) for k, v in F.kernel_timer.summary().items()})
