import torch
dev = torch.device("cuda:0")
R, C = 16384, 2048
x = torch.randn(R, C, device=dev).bfloat16()
ref = x.float().sum(0)
def trial(tag, junk_val, dtype=torch.bfloat16, dim=0):
    xx = x.to(dtype)
    out = torch.zeros(C if dim == 0 else R, device=dev)
    rf = xx.float().sum(dim)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out.copy_(xx.sum(dim).float())
        if junk_val is not None:
            junk = torch.full((1 << 18,), junk_val, dtype=torch.int32, device=dev)
            junk2 = junk + 1
    res = []
    for k in range(3):
        g.replay(); torch.cuda.synchronize()
        err = (out - rf).abs() / (1 + rf.abs())
        res.append((round(float(err.max()), 4), int((err > 0.05).sum())))
    print(tag, res)
trial("bf16 col-reduce, no junk", None)
trial("bf16 col-reduce, junk 0", 0)
trial("bf16 col-reduce, junk 7", 7)
trial("f32 col-reduce, junk 7", 7, torch.float32)
trial("bf16 row-reduce, junk 7", 7, torch.bfloat16, 1)
# full reduction of 16384 and of 1M elements
for n in (16384, 1 << 20):
    v = torch.randn(n, device=dev)
    out = torch.zeros((), device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out.copy_(v.mean())
        junk = torch.full((1 << 18,), 7, dtype=torch.int32, device=dev); junk2 = junk + 1
    r = []
    for k in range(3):
        g.replay(); torch.cuda.synchronize(); r.append(float(out - v.mean()))
    print("mean of", n, r)
