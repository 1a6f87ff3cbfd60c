"""Measured peaks of the box beside the datasheet numbers ("record the measured hipBLASLt GEMM peak and a
hipMemcpy / stream-triad HBM peak next to the datasheet numbers").  torch only (hipBLASLt through torch.matmul, ATen copy / add kernels):

    python tools/hw_peaks.py > gpurun_out/hw_peaks.json
"""
import json
import time

import torch

dev = torch.device("cuda", 0)


def timeit(f, n=20, warm=5):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {"device": torch.cuda.get_device_name(0), "datasheet": {"bf16_dense_tflops": 2500.0, "hbm_tbs": 8.0}}
# ---- HBM: copy (1 read + 1 write) and triad a = b + s * c (2 reads + 1 write), 2 GiB fp32 arrays
n = 512 * 1024 * 1024
a, b, c = (torch.empty(n, dtype=torch.float32, device=dev).normal_() for _ in range(3))
t = timeit(lambda: a.copy_(b))
res["hbm_copy_tbs"] = 2 * n * 4 / t / 1e12
t = timeit(lambda: torch.add(b, c, alpha=1.5, out=a))
res["hbm_triad_tbs"] = 3 * n * 4 / t / 1e12
t = timeit(lambda: a.fill_(1.0))
res["hbm_write_tbs"] = n * 4 / t / 1e12
t = timeit(lambda: torch.sum(b))
res["hbm_read_tbs"] = n * 4 / t / 1e12
del a, b, c
# ---- MFMA: hipBLASLt (torch.matmul) bf16 GEMMs, random data (the chip is power-limited: zeros would run faster)
res["gemm_bf16_tflops"] = {}
for M, N, K in [(8192, 8192, 8192), (16384, 4096, 4096), (16384, 4096, 14336), (16384, 28672, 4096), (16384, 32000, 4096)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: torch.matmul(x, w.t(), out=o), n=10, warm=3)
    res["gemm_bf16_tflops"][f"{M}x{N}x{K}"] = 2.0 * M * N * K / t / 1e12
    del x, w, o
print(json.dumps(res, indent=1))
