"""Per-shape GEMM table for the headline step (Llama-3-8B draft, 8 x 2048 tokens, ttt 7): every NT / row-addend / TN
shape the engine launches, own kernel vs torch.matmul (hipBLASLt) on the SAME random bf16 operands, variants interleaved
in one process over several rounds (median reported; CDNA guide section 5.4 rules 24/25).

    python tools/gemm_table.py [--rounds 5] > gpurun_out/gemm_table.jsonl      (GPU box)

Each line: form, name, M, N, K, launches_per_step, own_ms, lib_ms, own_tflops, lib_tflops, ratio.  The last line sums the
step's GEMM time for both."""
import json
import statistics
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import ops  # noqa: E402

dev = "cuda"
ROUNDS = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
def _arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


B, S, T = _arg("--batch", 8), _arg("--seq", 2048), 7       # (--batch 1 --seq 1024: the reference recipes' regime)
N_TOK = B * S
H, I, Vd, Vt, QW, Ht3 = 4096, 14336, 32000, 128256, 6144, 12288
NP = (B * (S + T) + 31) // 32 * 32

NT = [  # name, M, N, K, launches per step
    ("fc fwd", N_TOK, H, Ht3, 1), ("qkv(hidden half, rowadd) fwd", N_TOK, QW, H, 7), ("o fwd", N_TOK, H, H, 7),
    ("gate|up fwd", N_TOK, 2 * I, H, 7), ("down fwd", N_TOK, H, I, 7), ("lm_head fwd", N_TOK, Vd, H, 7),
    ("lm_head dgrad", N_TOK, H, Vd, 7), ("down dgrad", N_TOK, I, H, 7), ("gate|up dgrad", N_TOK, H, 2 * I, 7),
    ("o dgrad", N_TOK, H, H, 7), ("qkv dgrad (hidden half)", N_TOK, H, QW, 7), ("teacher head chunk", min(4096, N_TOK), Vt, H, max(1, N_TOK // 4096)),
    ("qkv embedding half fwd", NP, QW, H, 1), ("qkv embedding half dgrad", NP, H, QW, 1),
]
TN = [  # dW[M, N] = dY[K, M]^T . X[K, N]
    ("lm_head wgrad", Vd, H, T * N_TOK, 1), ("gate|up wgrad", 2 * I, H, T * N_TOK, 1), ("down wgrad", H, I, T * N_TOK, 1),
    ("qkv wgrad (hidden half)", QW, H, T * N_TOK, 1), ("qkv wgrad (embedding half)", QW, H, 2 * NP, 1),
    ("o wgrad", H, H, T * N_TOK, 1), ("fc wgrad", H, Ht3, N_TOK, 1),
]


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def ab(own, lib, iters):
    own(), lib()
    torch.cuda.synchronize()
    o, l = [], []
    for _ in range(ROUNDS):
        o.append(timed(own, iters))
        l.append(timed(lib, iters))
    return statistics.median(o), statistics.median(l)


tot_own = tot_lib = 0.0
WS = torch.empty(4 * 128 * 65536, device=dev)
for name, M, N, K, n in NT:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if "rowadd" in name:
        add = torch.randn(B * (S + T), N, device=dev)
        own = lambda: ops.gemm_nt_rowadd(a, b, c, add, S=S, Spad=S + T, off=3)
    else:
        # (the engine hands the N = H GEMMs a split-K workspace when tokens x H is at most 128 tiles: engine._buffers "nt_ws")
        ws = WS if (N == H and (N_TOK + 255) // 256 * ((H + 255) // 256) <= 128) else None
        own = lambda: ops.gemm_nt(a, b, c, workspace=ws)
    o, l = ab(own, lambda: torch.matmul(a, b.t(), out=c), 3 if N_TOK >= 8192 else 20)
    fl = 2.0 * M * N * K
    tot_own, tot_lib = tot_own + n * o, tot_lib + n * l
    print(json.dumps(dict(form="nt", name=name, M=M, N=N, K=K, launches_per_step=n, own_ms=round(o, 4), lib_ms=round(l, 4),
                          own_tflops=round(fl / o / 1e9, 1), lib_tflops=round(fl / l / 1e9, 1), ratio=round(l / o, 4))), flush=True)
    del a, b, c
for name, M, N, K, n in TN:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16)
    b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(2 * M * N, device=dev)
    o, l = ab(lambda: ops.gemm_tn(a, b, c, workspace=ws), lambda: torch.matmul(a.t(), b, out=c), 2)
    fl = 2.0 * M * N * K
    tot_own, tot_lib = tot_own + n * o, tot_lib + n * l
    print(json.dumps(dict(form="tn", name=name, M=M, N=N, K=K, launches_per_step=n, own_ms=round(o, 4), lib_ms=round(l, 4),
                          own_tflops=round(fl / o / 1e9, 1), lib_tflops=round(fl / l / 1e9, 1), ratio=round(l / o, 4))), flush=True)
    del a, b, c, ws
print(json.dumps(dict(form="sum", name="all GEMM launches of one step", own_ms=round(tot_own, 2), lib_ms=round(tot_lib, 2),
                      ratio=round(tot_lib / tot_own, 4))), flush=True)
