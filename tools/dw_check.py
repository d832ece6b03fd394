import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tsg_oracle as O
from shufflingvideosfortsg_amd import functional as F
B, T, N, H, Ds = 130, 100, 20, 256, 256
g = torch.Generator().manual_seed(5)
a = torch.randn(B, T, H, generator=g); s = torch.randn(B, N, H, generator=g); w = torch.randn(H, generator=g) / H ** 0.5
sent = torch.randn(B, N, Ds, generator=g); gC = torch.randn(B, T, Ds, generator=g)
def run(dt):
    xs = [x.to(dt).detach().requires_grad_(True) for x in (a, s, w, sent)]
    C, P = O.scdm_core(*xs); C.backward(gC.to(dt)); return [x.grad for x in xs]
r64 = run(torch.float64); r32 = run(torch.float32)
xs = [x.cuda().detach().requires_grad_(True) for x in (a, s, w, sent)]
C = F.scdm_attn(*xs); C.backward(gC.cuda())
for name, g64, g32, gk in zip("a s w sent".split(), r64, r32, [x.grad.cpu() for x in xs]):
    print(name, "oracle32 vs 64: %.3e   kernel vs 64: %.3e   max|ref| %.3e" % ((g32.double() - g64).abs().max(), (gk.double() - g64).abs().max(), g64.abs().max()))
