"""Time myo_lstm_step_fwd / _bwd (one launch per step) against bmm + myo_lstm_cell_* at config-E shapes, each as a 32-step
hipGraph.  usage: lstm_step_time.py [m=512] [H=256]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from myochallenge_amd import native
lib = native.load(); L = lib.L
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T, G = 32, 2
dev, bf = torch.device("cuda:0"), torch.bfloat16
mk = lambda *s, sc=1.0: (sc * torch.randn(*s, device=dev)).to(bf)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
gx = mk(T, m, G, 4 * H); gxs = gx.transpose(1, 2).contiguous()
whh = mk(G, 4 * H, H, sc=H ** -0.5); wt = whh.transpose(1, 2).contiguous()
hm, cm = mk(T + 1, G, m, H), mk(T + 1, G, m, H)
cn, ws, lat = mk(T, G, m, H), mk(T, G, m, 4 * H).abs().clamp(max=0.9), mk(G, T, m, H)
out = mk(T, G, m, H)
dG, dcm, dlat = mk(T, G, m, 4 * H, sc=0.1), mk(2, G, m, H), mk(G, T, m, H)
keep = (torch.rand(T, m, device=dev) > 0.1).float()

def fwd_fused():
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for t in range(T):
        lib.check(L.myo_lstm_step_fwd(p(gx[t]), 4 * H, G * 4 * H, p(hm[t]), p(cm[t]), p(whh), p(keep[t]), G, m, H, p(lat[:, t]), T * m * H,
                                      p(hm[t + 1]), p(cm[t + 1]), p(cn[t]), p(ws[t]), None, None, st))
def fwd_pair():
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for t in range(T):
        gh = torch.bmm(hm[t], wt)
        lib.check(L.myo_lstm_cell_fwd(p(gxs[t]), p(gh), p(cm[t]), p(keep[t]), G * m, m, H, 1, p(out[t]), p(hm[t + 1]), p(cm[t + 1]), p(cn[t]), p(ws[t]), None, None, st))
def bwd_fused():
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for t in range(T - 1, -1, -1):
        last = t == T - 1
        lib.check(L.myo_lstm_step_bwd(p(dlat[:, t]), T * m * H, None if last else p(dG[t + 1]), None if last else p(dcm[(t + 1) & 1]), p(wt),
                                      p(keep[t]), p(cm[t]), p(cn[t]), p(ws[t]), G, m, H, p(dG[t]), p(dcm[t & 1]), st))
def bwd_pair():
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    dhm = dcn = None
    for t in range(T - 1, -1, -1):
        lib.check(L.myo_lstm_cell_bwd(p(out[t]), p(dhm), p(dcn), p(keep[t]), p(cm[t]), p(cn[t]), p(ws[t]), G * m, m, H, 1, p(dG[t]), p(dcm[t & 1]), st))
        if t > 0:
            dhm, dcn = torch.bmm(dG[t], whh), dcm[t & 1]

for name, fn in (("fwd fused", fwd_fused), ("fwd gemm+cell", fwd_pair), ("bwd fused", bwd_fused), ("bwd gemm+cell", bwd_pair)):
    keep_state = [t.clone() for t in (hm, cm, dG)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:16s} {(time.time() - t0) / 20 / T * 1e6:7.2f} us per step (m={m}, H={H})")
    for t, k in zip((hm, cm, dG), keep_state):
        t.copy_(k)
