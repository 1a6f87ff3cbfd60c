"""Per-kernel timings at the Llama-3-8B EAGLE3 (cfg 2) shapes on one MI355X.
Usage (GPU box): python tools/microbench.py [--quick] > gpurun_out/microbench.jsonl
Each line: kernel, shape, ms, achieved TFLOP/s or GB/s.  HIP events on the current stream."""
import json
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from specforge_amd import ops  # noqa: E402

dev = "cuda"
QUICK = "--quick" in sys.argv


def timeit(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_gemm(M, N, K, name):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm_nt(a, b, c))
    ref_ms = timeit(lambda: torch.matmul(a, b.t(), out=c))
    fl = 2.0 * M * N * K
    emit(kernel="gemm_nt", name=name, M=M, N=N, K=K, ms=ms, tflops=fl / ms / 1e9, rocblas_ms=ref_ms,
         rocblas_tflops=fl / ref_ms / 1e9)


def bench_attn(B, S, nh, nkv, hd, ndiag):
    N = B * S
    qkv = [torch.randn(N, (nh + 2 * nkv) * hd, device=dev).to(torch.bfloat16) for _ in range(ndiag + 1)]
    q = qkv[-1][:, :nh * hd]
    kv = [t[:, nh * hd:(nh + nkv) * hd] for t in qkv]
    vv = [t[:, (nh + nkv) * hd:] for t in qkv]
    o = torch.empty(N, nh * hd, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, nh, S, device=dev)
    kw = dict(B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=1 / math.sqrt(hd))
    f = lambda: ops.attn_fwd(q, kv[0], vv[0], kv[1:], vv[1:], None, o, lse, **kw)
    ms = timeit(f)
    fl = 4.0 * B * nh * hd * S * S / 2
    emit(kernel="attn_fwd", B=B, S=S, nh=nh, nkv=nkv, hd=hd, ndiag=ndiag, ms=ms, tflops=fl / ms / 1e9)
    do = torch.randn(N, nh * hd, device=dev).to(torch.bfloat16)
    delta = torch.empty(B, nh, S, device=dev)
    dq_init = torch.zeros(N, nh * hd, device=dev)
    dk = [torch.zeros(N, nkv * hd, device=dev) for _ in range(ndiag + 1)]
    dv = [torch.zeros(N, nkv * hd, device=dev) for _ in range(ndiag + 1)]
    dq = torch.empty(N, nh * hd, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.attn_bwd_pre(q, o, do, kv[1:], vv[1:], dk[1:], dv[1:], lse, delta, dq_init, **kw))
    emit(kernel="attn_bwd_pre", ndiag=ndiag, ms=ms)
    ms = timeit(lambda: ops.attn_bwd_dq(q, do, kv[0], vv[0], None, lse, delta, dq_init, dq, **kw))
    emit(kernel="attn_bwd_dq", ms=ms, tflops=6.0 * B * nh * hd * S * S / 2 / ms / 1e9)
    ms = timeit(lambda: ops.attn_bwd_dkv(q, do, kv[0], vv[0], None, lse, delta, dk[0], dv[0], **kw))
    emit(kernel="attn_bwd_dkv", ms=ms, tflops=8.0 * B * nh * hd * S * S / 2 / ms / 1e9)


def bench_ce(B, S, V, T=7):
    N, Spad = B * S, S + T
    logits = torch.randn(N, V, device=dev).to(torch.bfloat16)
    target = torch.softmax(torch.randn(B, Spad, V, device=dev), -1)
    ones = torch.ones(B, Spad, device=dev, dtype=torch.int32)
    pod = torch.rand(B, Spad, device=dev)
    tsum = target.sum(-1)
    ids = torch.randint(0, V, (B, Spad), device=dev)
    d2t = torch.zeros(V, device=dev, dtype=torch.int64)
    rl, rc, ra = (torch.empty(N, device=dev) for _ in range(3))
    f = lambda: ops.ce_fused(logits, target, S=S, Spad=Spad, off=3, pos_mask_pad=ones, loss_mask_pad=ones, tgt_ids_pad=ids,
                             pod_scale_pad=pod, tsum_pad=tsum, d2t=d2t, grad_scale=1e-4, row_loss=rl, row_correct=rc,
                             row_accept=ra)
    ms = timeit(f)
    emit(kernel="ce_fused", B=B, S=S, V=V, ms=ms, gbs_algorithmic=8.0 * N * V / ms / 1e6)


def bench_pointwise(N, H, I):
    x = torch.randn(N, H, device=dev).to(torch.bfloat16)
    w = torch.ones(H, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    rstd = torch.empty(N, device=dev)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-5, y, rstd))
    emit(kernel="rmsnorm_fwd", N=N, H=H, ms=ms, gbs=4.0 * N * H / ms / 1e6)
    dx = torch.empty_like(x)
    dw = torch.zeros(H, device=dev)
    ws = torch.empty(ops.rmsnorm_bwd_workspace(N, H), device=dev)
    ms = timeit(lambda: ops.rmsnorm_bwd(y, x, w, rstd, dx=dx, add=x, dw_acc=dw, workspace=ws))
    emit(kernel="rmsnorm_bwd", N=N, H=H, ms=ms, gbs=8.0 * N * H / ms / 1e6)
    gu = torch.randn(N, 2 * I, device=dev).to(torch.bfloat16)
    act = torch.empty(N, I, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.swiglu_fwd(gu, act))
    emit(kernel="swiglu_fwd", N=N, I=I, ms=ms, gbs=6.0 * N * I / ms / 1e6)
    dgu = torch.empty_like(gu)
    ms = timeit(lambda: ops.swiglu_bwd(act, gu, dgu))
    emit(kernel="swiglu_bwd", N=N, I=I, ms=ms, gbs=10.0 * N * I / ms / 1e6)
    t = torch.empty(2 * I, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.transpose2d(gu, t))
    emit(kernel="transpose2d", R=N, C=2 * I, ms=ms, gbs=8.0 * N * I / ms / 1e6)


if __name__ == "__main__":
    torch.manual_seed(0)
    t0 = time.time()
    N = 16384
    bench_gemm(N, 14336 * 2, 4096, "gate_up fwd")
    bench_gemm(N, 4096, 14336, "down fwd")
    bench_gemm(N, 32000, 4096, "lm_head fwd")
    bench_gemm(N, 6144, 8192, "qkv fwd")
    bench_gemm(N, 4096, 4096, "o fwd")
    if not QUICK:
        bench_gemm(N, 4096, 32000, "lm_head dgrad")
        bench_gemm(32000, 4096, N, "lm_head wgrad (1 step)")
        bench_gemm(2 * 14336, 4096, N, "gate_up wgrad (1 step)")
        bench_gemm(4096, 4096, 4096, "4096^3")
        bench_gemm(8192, 8192, 8192, "8192^3")
    bench_attn(8, 2048, 32, 8, 128, 0)
    bench_attn(8, 2048, 32, 8, 128, 6)
    bench_ce(8, 2048, 32000)
    bench_pointwise(N, 4096, 14336)
    emit(done=True, wall_s=time.time() - t0)
