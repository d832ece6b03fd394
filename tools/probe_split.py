"""Accuracy / speed of a 3-product bf16 split GEMM (hi·hi + hi·lo + lo·hi, fp32 accumulate) vs the fp32 GEMM."""
import torch, time
dev = "cuda"
BF = torch.bfloat16

def split2(x):
    hi = x.to(BF)
    lo = (x - hi).to(BF)
    return hi, lo

def mm_split(a, b):            # a [M,K], b [K,N] row-major
    a1, a2 = split2(a); b1, b2 = split2(b)
    A = torch.cat([a1, a1, a2], 1); Bc = torch.cat([b1, b2, b1], 0)
    return torch.mm(A, Bc, out_dtype=torch.float32)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

torch.manual_seed(0)
for (M, K, N) in [(16384, 1024, 4096), (4096, 16384, 1024), (2048, 16384, 512)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev) * 0.03
    ref = (a.double() @ b.double())
    e32 = ((a @ b).double() - ref).abs().max().item() / ref.abs().max().item()
    es = (mm_split(a, b).double() - ref).abs().max().item() / ref.abs().max().item()
    eb = ((a.to(BF) @ b.to(BF)).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{M}x{K}]x[{K}x{N}]  max-rel-to-max err: fp32 {e32:.2e}  split3 {es:.2e}  bf16 {eb:.2e}   "
          f"time fp32 {bench(lambda: a @ b):.0f} us  split3 {bench(lambda: mm_split(a, b)):.0f} us", flush=True)
