"""TN GEMM, two big shapes, pace sync off (2 launches) then on (2 launches) each: run under rocprofv3 --pmc FETCH_SIZE and
read the counter per dispatch in launch order (tools/gemm_tn_pmc_rows.py)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
K = 7 * 16384
for (M, N) in [(32000, 4096), (28672, 4096)]:
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(2 * M * N + 4096, device="cuda")
    for sync in ("0", "128"):
        os.environ["SF_GEMM_TN_SYNC"] = sync
        for _ in range(2):
            ops.gemm_tn(a, b, c, workspace=ws)
    torch.cuda.synchronize()
    del a, b, c, ws
