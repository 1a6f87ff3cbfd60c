import sys, torch
sys.path.insert(0, ".")
from specforge_amd import ops
torch.manual_seed(0)
B, S, T, H, QW = 8, 2048, 7, 4096, 6144
N, Spad = B * S, S + T
a = torch.randn(N, H, device="cuda").to(torch.bfloat16); w = torch.randn(QW, 2 * H, device="cuda").to(torch.bfloat16)
add = torch.randn(B * Spad + 24, QW, device="cuda"); out = torch.empty(N, QW, device="cuda", dtype=torch.bfloat16)
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
fl = 2.0 * N * QW * H
ms = t(lambda: ops.gemm_nt(a, w[:, H:], out)); print("plain   ms", round(ms, 4), "TF", round(fl / ms / 1e9))
ms = t(lambda: ops.gemm_nt_rowadd(a, w[:, H:], out, add, S=S, Spad=Spad, off=3)); print("rowadd  ms", round(ms, 4), "TF", round(fl / ms / 1e9))
