import time, torch
dev = "cuda"
B = 16384
x = torch.randn(B, 256, device=dev).bfloat16(); w = torch.randn(256, 256, device=dev).bfloat16(); b = torch.randn(256, device=dev).bfloat16()
x2 = torch.randn(2, B, 256, device=dev).bfloat16(); w2 = torch.randn(2, 256, 256, device=dev).bfloat16()
def bench(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e6
print("addmm            us", bench(lambda: torch.addmm(b, x, w.t())))
print("addmm+relu_      us", bench(lambda: torch.relu_(torch.addmm(b, x, w.t()))))
print("_addmm_activation us", bench(lambda: torch._addmm_activation(b, x, w.t())))
print("bmm (2 nets)     us", bench(lambda: torch.bmm(x2, w2.transpose(1, 2))))
r1 = torch.relu(torch.addmm(b, x, w.t())); r2 = torch._addmm_activation(b, x, w.t())
print("max diff", float((r1.float() - r2.float()).abs().max()))
x86 = torch.randn(B, 86, device=dev).bfloat16(); w86 = torch.randn(256, 86, device=dev).bfloat16()
print("_addmm_activation 86 us", bench(lambda: torch._addmm_activation(b, x86, w86.t())))
