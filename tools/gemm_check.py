"""GEMM stress: correctness vs torch.matmul at training shapes + run-to-run bitwise determinism + timing."""
import sys, json, torch
sys.path.insert(0, ".")
from specforge_amd import ops
dev = "cuda"
torch.manual_seed(0)
def t(fn, w=2, n=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [(16384, 28672, 4096), (16384, 4096, 14336), (16384, 32000, 4096), (16384, 6144, 8192), (16384, 4096, 4096),
          (32000, 4096, 114688), (28672, 4096, 114688), (4096, 14336, 114688), (6144, 8192, 114688), (4096, 12288, 16384),
          (4096, 128256, 4096), (1000, 777*8//8*8, 640), (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm_nt(a, b, c)
    ref = torch.matmul(a, b.t())
    err = float((c.float() - ref.float()).abs().max()); scale = float(ref.float().abs().max())
    c2 = torch.empty_like(c)
    same = True
    for _ in range(3):
        ops.gemm_nt(a, b, c2)
        same &= bool(torch.equal(c, c2))
    ms = t(lambda: ops.gemm_nt(a, b, c))
    ms_ref = t(lambda: torch.matmul(a, b.t(), out=ref))
    print(json.dumps(dict(M=M, N=N, K=K, relerr=err / scale, deterministic=same, ms=ms, tflops=2.0 * M * N * K / ms / 1e9,
                          hipblaslt_tflops=2.0 * M * N * K / ms_ref / 1e9)), flush=True)
    del a, b, c, c2, ref
