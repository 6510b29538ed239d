"""Does a hipMemsetAsync issued during stream capture become a graph node that runs on every replay?
And: does a multi-block torch column reduction (semaphore-based global reduce) survive replays when the
memory it takes its semaphores from is dirty?"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
for nbytes in (4, 64, 4096, 1 << 20):
    buf = torch.ones(max(nbytes // 4, 1), dtype=torch.int32, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(nbytes), ctypes.c_void_p(st))
    torch.cuda.synchronize()
    after_capture = int(buf.sum())
    buf.fill_(1); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("memset", nbytes, "rc", rc, "sum after capture (expect untouched = n)", after_capture, "sum after replay (expect 0)", int(buf.sum()))

# column reduction [R, C] -> [C] inside a graph, with the private pool dirtied between replays by the graph itself
R, C = 16384, 2048
x = torch.randn(R, C, device=dev).bfloat16()
out = torch.zeros(C, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out.copy_(x.sum(0).float())
    junk = torch.full((1 << 16,), 7, dtype=torch.int32, device=dev)      # reuses the freed semaphore / staging blocks
    junk2 = junk + 1
ref = x.float().sum(0)
for k in range(4):
    g.replay(); torch.cuda.synchronize()
    err = (out - ref).abs() / (1 + ref.abs())
    print("replay", k, "max rel err", float(err.max()), "bad columns", int((err > 0.05).sum()))
