"""Which hipBLASLt/Tensile kernels torch.matmul picks for the step's NT shapes (run under rocprofv3 --kernel-trace)."""
import torch
dev = "cuda"
for (M, N, K) in [(16384, 4096, 4096), (16384, 4096, 14336), (16384, 28672, 4096), (16384, 32000, 4096), (16384, 4096, 32000), (16384, 6144, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
for (M, N, K) in [(4096, 14336, 114688)]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        torch.matmul(a.t(), b, out=c)
    torch.cuda.synchronize()
