"""What the weight-gradient (TN) GEMMs lose when a communication kernel holds CUs -- single-GPU evidence for the N-GPU
run (VERDICT r2 "next" 2a).  An RCCL all-reduce runs as 8-32 workgroups of modest register use; a TN GEMM workgroup
owns its CU outright (512 registers per SIMD lane, 128 KiB LDS), so the collective's workgroups take CUs away from the
GEMM grid for as long as they run, and the GEMM's pace groups (32 co-resident workgroups per XCD meeting at a counter)
are then never complete.  `sf_tool_cu_hog` (tools build) is the stand-in: n workgroups parked on a side stream.

    python tools/contention_bench.py > gpurun_out/contention.jsonl        (GPU box, tools build)

Per weight-gradient shape of the headline step: time alone and under 8 / 16 / 32 held CUs, pacing on
(SF_GEMM_TN_SYNC=128) and off (=0)."""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

L = _lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
L.sf_tool_cu_hog.restype = ctypes.c_int
L.sf_tool_cu_hog.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = "cuda"
K = 7 * 16384
SHAPES = [("lm_head", 32000, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("qkv", 6144, 4096), ("o", 4096, 4096)]
side = torch.cuda.Stream()


def run(a, b, c, ws, hog, iters=2):
    """ms per GEMM with `hog` CUs held for the whole timed region"""
    torch.cuda.synchronize()
    if hog:
        L.sf_tool_cu_hog(hog, 100_000, ctypes.c_void_p(side.cuda_stream))   # 0.1 s: longer than the timed region
        torch.cuda._sleep(200_000)                                           # let the hog's workgroups take their CUs first
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm_tn(a, b, c, workspace=ws)
    e.record()
    e.synchronize()
    ms = s.elapsed_time(e) / iters
    torch.cuda.synchronize()
    return ms


for name, M, N in SHAPES:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16)
    b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(2 * M * N + 4096, device=dev)
    row = dict(shape=name, M=M, N=N, K=K)
    for hog in (0, 8, 16, 32):
        for sync in (128, 0):
            os.environ["SF_GEMM_TN_SYNC"] = str(sync)
            run(a, b, c, ws, hog, 1)
            xs = [run(a, b, c, ws, hog) for _ in range(3)]
            row[f"hog{hog}_sync{sync}_ms"] = round(statistics.median(xs), 3)
    print(json.dumps(row), flush=True)
    del a, b, c, ws
