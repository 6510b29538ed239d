import torch
from myochallenge_amd.rl.vec_normalize import RunningMeanStd
dev="cuda"
torch.manual_seed(0)
r = RunningMeanStd((86,), dev)
x = torch.randn(4096, 86, device=dev) * 0.01 + 1.4
for i in range(300):
    r.update(x + 0.001*i)
print("eager var min", float(r.var.min()), float(r.count))
r2 = RunningMeanStd((86,), dev)
xs = torch.zeros(4096, 86, device=dev)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): r2.update(xs)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r2.update(xs)
r2 = r2
for i in range(300):
    xs.copy_(x + 0.001*i); g.replay()
torch.cuda.synchronize()
print("graph var min", float(r2.var.min()), float(r2.count))
