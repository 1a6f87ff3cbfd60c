"""The last partial round of the persistent NT GEMM: lm_head forward (16384 x 32000 x 4096 = 8000 tiles = 31.25 rounds on 256 CUs) as one
launch vs 124 column tiles (31 rounds exactly) + the last 256 columns by another kernel.  Tools build; the knobs are read once per
process, so run it once per tail kernel: default (8-wave 256 kernel), SF_GEMM_TILE=128 (128 x 128 kernel), SF_GEMM_W4=1 (4-wave)."""
import json, os, sys, torch
sys.path.insert(0, ".")
from specforge_amd import _lib, ops
_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so")); _lib._emulated = False
torch.manual_seed(0)
M, N, K = 16384, 32000, 4096
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / 64).to(torch.bfloat16)
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
NM = 124 * 256
res = {"env": {k: v for k, v in os.environ.items() if k.startswith("SF_GEMM")}}
for rnd in range(2):
    res.setdefault("full_ms", []).append(round(t(lambda: ops.gemm_nt(a, w, c)), 4))
    res.setdefault("main_31_rounds_ms", []).append(round(t(lambda: ops.gemm_nt(a, w[:NM], c[:, :NM])), 4))
    res.setdefault("tail_256cols_ms", []).append(round(t(lambda: ops.gemm_nt(a, w[NM:], c[:, NM:])), 4))
print(json.dumps(res))
