import sys, torch
sys.path.insert(0, ".")
from specforge_amd import ops
torch.manual_seed(0)
def t(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for (M, N, K, n) in [(8192, 8192, 8192, 100), (28672, 4096, 114688, 12), (32000, 4096, 114688, 12)]:
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    ms = t(lambda: ops.gemm_tn(a, b, c), n)
    print(f"{M}x{N}x{K}  TN {ms:.3f} ms {fl/ms/1e9:.0f} TF")
    del a, b, c
