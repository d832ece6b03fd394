import math, os, sys
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import TSG_F32, TSG_F32S, ptr
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
B, T, Tk, d, h = 64, 128, 33, 1024, 8
Q = torch.randn(B, T, d, device="cuda"); K = torch.randn(B, Tk, d, device="cuda"); V = torch.randn(B, Tk, d, device="cuda")
O = torch.empty(B, T, d, device="cuda"); lse = torch.empty(B, h, T, device="cuda"); g = torch.randn(B, T, d, device="cuda")
dQ, dK, dV, dlt = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(lse)
sc = math.sqrt(d)
lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32, st)
for _ in range(20):
    lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32S, st)
torch.cuda.synchronize()
