"""One shape, own NT kernel (tools build, knobs from argv[1] like "SF_GEMM_SCHED=0") and torch.matmul, a few launches each:
run under rocprofv3 --pmc ... to compare SQ wait / busy counters per kernel."""
import os
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "base" else []):
    k, v = kv.split("=")
    os.environ[k] = v
M, N, K = 16384, 4096, 14336
zero = os.environ.get("ZERO") == "1"
a = (torch.zeros if zero else torch.randn)(M, K, device="cuda").to(torch.bfloat16)
b = (torch.zeros if zero else torch.randn)(N, K, device="cuda").to(torch.bfloat16)
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(4):
    ops.gemm_nt(a, b, c)
for _ in range(4):
    torch.matmul(a, b.t(), out=c)
torch.cuda.synchronize()
