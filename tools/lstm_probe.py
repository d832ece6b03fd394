import time, torch, sys
t0=time.time()
B,T,d=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
m=torch.nn.LSTM(1024,d//2,2,batch_first=True,bidirectional=True,dropout=0.5).cuda().train()
x=torch.randn(B,T,1024,device='cuda',requires_grad=True)
for i in range(4):
    torch.cuda.synchronize(); t=time.time()
    o,_=m(x); torch.cuda.synchronize(); t1=time.time()
    o.sum().backward(); torch.cuda.synchronize(); t2=time.time()
    print(f"iter {i}: fwd {1e3*(t1-t):.1f} ms  bwd {1e3*(t2-t1):.1f} ms", flush=True)
print("total", time.time()-t0)
