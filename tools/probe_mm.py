import torch, time
a=torch.randn(4096,1024,device='cuda'); b=torch.randn(1024,4096,device='cuda')
ah=a.bfloat16(); bh=b.bfloat16()
try:
    c=torch.mm(ah,bh,out_dtype=torch.float32); print("mm out_dtype ok", c.dtype)
except Exception as e: print("mm out_dtype FAIL", type(e).__name__, str(e)[:200])
try:
    c=torch._scaled_mm; print("has _scaled_mm")
except Exception as e: print("no scaled_mm")
def t(fn,n=20):
    fn(); torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e6
A=torch.randn(16384,1024,device='cuda'); W=torch.randn(4096,1024,device='cuda')
print("fp32 gemm us", t(lambda: A@W.t()))
Ah=A.bfloat16(); Wh=W.bfloat16()
print("bf16 gemm us", t(lambda: Ah@Wh.t()))
A3=torch.cat([Ah,Ah,Ah],1); W3=torch.cat([Wh,Wh,Wh],1)
print("bf16 gemm K*3 us", t(lambda: A3@W3.t()))
try:
    print("bf16->f32 out K*3 us", t(lambda: torch.mm(A3,W3.t(),out_dtype=torch.float32)))
except Exception as e: print("FAIL", str(e)[:100])
