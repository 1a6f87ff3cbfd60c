"""NT GEMM epilogue A/B in one process: the product library (current sources) vs the tools library as built from an EARLIER commit (build it
before the change under test: python specforge_amd/build.py ablate), interleaved rounds on the step's bf16 shapes.  (GPU box)"""
import json, os, statistics, sys, torch
sys.path.insert(0, ".")
from specforge_amd import _lib, ops
OLD = os.path.join("tools", "experiments", sys.argv[1] if len(sys.argv) > 1 else "libsfhip_old.so")   # a product build of the earlier commit
SHAPES = [(16384, 28672, 4096), (16384, 4096, 4096), (16384, 32000, 4096), (16384, 4096, 4096), (16384, 14336, 4096), (16384, 4096, 14336), (16384, 6144, 4096), (16384, 4096, 4096), (16384, 4096, 32000)]
def timed(fn, iters=5):
    for _ in range(2): fn()      # untimed: the first launches after a switch / an idle moment run at a lower clock (a null test -- the same
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # library on both sides -- showed up to -4 % for whichever ran first)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
def use(old):
    _lib._inject_library_for_tests(OLD if old else None); _lib._emulated = False
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(N, K, device="cuda") / 64).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {"new": [], "old": []}
    for rnd in range(7):
        for which in (("new", "old") if rnd % 2 else ("old", "new")):     # alternate the order
            use(which == "old")
            ms = timed(lambda: ops.gemm_nt(a, b, c))
            if rnd: res[which].append(ms)
    n, o = statistics.median(res["new"]), statistics.median(res["old"])
    print(json.dumps(dict(shape=[M, N, K], new_ms=round(n, 4), old_ms=round(o, 4), gain_pct=round(100 * (o - n) / o, 2))), flush=True)
# weight-gradient form (TN), K shortened to keep the run short
for (M, N, K) in [(32000, 4096, 32768), (4096, 14336, 32768), (6144, 4096, 32768)]:
    dy = torch.randn(K, M, device="cuda").to(torch.bfloat16); xx = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    g = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); ws = torch.empty(2 * M * N + 4096, device="cuda")
    res = {"new": [], "old": []}
    for rnd in range(5):
        for which in (("new", "old") if rnd % 2 else ("old", "new")):
            use(which == "old")
            ms = timed(lambda: ops.gemm_tn(dy, xx, g, workspace=ws), iters=3)
            if rnd: res[which].append(ms)
    n, o = statistics.median(res["new"]), statistics.median(res["old"])
    print(json.dumps(dict(shape=["tn", M, N, K], new_ms=round(n, 4), old_ms=round(o, 4), gain_pct=round(100 * (o - n) / o, 2))), flush=True)
# fused gate|up + SwiGLU
M, I, K = 16384, 14336, 4096
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(2 * I, K, device="cuda") / 64).to(torch.bfloat16)
gu = torch.empty(M, 2 * I, device="cuda", dtype=torch.bfloat16); act = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
res = {"new": [], "old": []}
for rnd in range(7):
    for which in (("new", "old") if rnd % 2 else ("old", "new")):
        use(which == "old")
        ms = timed(lambda: ops.gemm_nt_swiglu_fwd(x, w, gu, act))
        if rnd: res[which].append(ms)
n, o = statistics.median(res["new"]), statistics.median(res["old"])
print(json.dumps(dict(shape="gate|up + SwiGLU fwd", new_ms=round(n, 4), old_ms=round(o, 4), gain_pct=round(100 * (o - n) / o, 2))))

# fused down-dgrad + d(SwiGLU)
dy = torch.randn(M, K, device="cuda").to(torch.bfloat16); wd = (torch.randn(I, K, device="cuda") / 64).to(torch.bfloat16)
guv = torch.randn(M, 2 * I, device="cuda").to(torch.bfloat16); dgu = torch.empty_like(guv); scr = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
res = {"new": [], "old": []}
for rnd in range(7):
    for which in (("new", "old") if rnd % 2 else ("old", "new")):
        use(which == "old")
        ms = timed(lambda: ops.gemm_nt_swiglu_bwd(dy, wd, guv, dgu, scr))
        if rnd: res[which].append(ms)
n, o = statistics.median(res["new"]), statistics.median(res["old"])
print(json.dumps(dict(shape="down dgrad + d(SwiGLU)", new_ms=round(n, 4), old_ms=round(o, 4), gain_pct=round(100 * (o - n) / o, 2))))
