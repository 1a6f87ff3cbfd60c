import sys, torch
sys.path.insert(0, ".")
from specforge_amd import ops
torch.manual_seed(0)
def t(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for (M, N, K, n) in [(8192, 8192, 8192, 100), (28672, 4096, 114688, 12), (32000, 4096, 114688, 12), (4096, 14336, 114688, 12), (6144, 4096, 114688, 12)]:
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    ms = t(lambda: ops.gemm_tn(a, b, c), n)
    at = a.t().contiguous(); bt = b.t().contiguous(); c2 = torch.empty_like(c)
    ms2 = t(lambda: ops.gemm_nt(at, bt, c2), n)
    err = float((c.float() - c2.float()).abs().max() / c2.float().abs().max())
    print(f"{M}x{N}x{K}  TN {ms:.3f} ms {fl/ms/1e9:.0f} TF | NT {ms2:.3f} ms {fl/ms2/1e9:.0f} TF | relerr {err:.2e}")
    del a, b, c, at, bt, c2
