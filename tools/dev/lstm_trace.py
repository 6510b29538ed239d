import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myochallenge_amd.rl.policy import _LstmSeq
torch.manual_seed(0)
T, G, N, H = 4, 2, 512, 256
dt = torch.bfloat16
def run():
    gx = torch.randn(T, G, N, 4 * H, device="cuda", dtype=dt, requires_grad=True)
    wt = torch.randn(G, H, 4 * H, device="cuda", dtype=dt, requires_grad=True)
    h0 = torch.zeros(G, N, H, device="cuda", dtype=dt); c0 = torch.zeros_like(h0)
    keep = torch.ones(T, 1, N, 1, device="cuda", dtype=dt)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out, h, c = _LstmSeq.apply(gx, wt, h0, c0, keep)
    torch.cuda.synchronize(); print("MARK fwd done", flush=True)
    out.float().sum().backward()
    torch.cuda.synchronize()
run(); run()
