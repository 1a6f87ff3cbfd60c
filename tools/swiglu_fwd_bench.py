"""fused gate|up projection + SwiGLU (sf_gemm_nt_swiglu_fwd) vs gemm_nt + swiglu_fwd at the headline shape, one process"""
import sys, json, torch
sys.path.insert(0, ".")
from specforge_amd import ops
torch.manual_seed(0)
M, I, K = 16384, 14336, 4096
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(2 * I, K, device="cuda") / 64).to(torch.bfloat16)
gu = torch.empty(M, 2 * I, device="cuda", dtype=torch.bfloat16); act = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
def two():
    ops.gemm_nt(x, w, gu); ops.swiglu_fwd(gu, act)
res = {}
for rnd in range(3):
    res.setdefault("gemm_only_ms", []).append(round(t(lambda: ops.gemm_nt(x, w, gu)), 4))
    res.setdefault("two_step_ms", []).append(round(t(two), 4))
    res.setdefault("fused_ms", []).append(round(t(lambda: ops.gemm_nt_swiglu_fwd(x, w, gu, act)), 4))
print(json.dumps(res))
