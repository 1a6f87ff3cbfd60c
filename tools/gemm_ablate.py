"""Time one shape under SF_GEMM_VARIANT (set by the caller); ablation variants give wrong numbers by design."""
import sys, os, torch
sys.path.insert(0, ".")
from specforge_amd import ops
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 8192)))
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
if os.environ.get("ZERO"): a.zero_(); b.zero_()
IT = int(os.environ.get("IT", "10"))
if os.environ.get("TORCH"):
    bt = b.t()
    class _O:
        @staticmethod
        def gemm_nt(a, b, c): torch.matmul(a, bt, out=c)
    ops = _O
for _ in range(3): ops.gemm_nt(a, b, c)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(IT): ops.gemm_nt(a, b, c)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / IT
print(f"zero={os.environ.get('ZERO')} w4={os.environ.get('SF_GEMM_W4')} it={IT} variant={os.environ.get('SF_GEMM_VARIANT')} {M}x{N}x{K} ms={ms:.4f} tflops={2.0*M*N*K/ms/1e9:.0f}")
