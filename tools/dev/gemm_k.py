import torch, time
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for M in (4096, 512*32):
  for K in (103, 104, 112, 128, 256):
    for dt in (torch.bfloat16, torch.float32):
        x=torch.randn(M,K,device='cuda',dtype=dt); w=torch.randn(2048,K,device='cuda',dtype=dt); b=torch.randn(2048,device='cuda',dtype=dt)
        print(M,K,dt, 'linear us', round(t(lambda: torch.nn.functional.linear(x,w,b)),1))
