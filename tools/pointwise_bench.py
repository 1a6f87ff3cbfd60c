"""HBM-bound row kernels of the headline step at their real shapes: time, algorithmic bytes, GB/s.

    python tools/pointwise_bench.py > gpurun_out/pointwise_bench.jsonl      (GPU box)

teacher_reduce   4096 x 128256 bf16 logits -> 32000 fp32 probabilities per row (+ ids, masks, scales)
rmsnorm_bwd      16384 x 4096 bf16: x, dy, add -> dx, dw
Each line: name, ms (median of 5 rounds of 10 launches), bytes (read + written once), GBps."""
import json
import statistics
import sys

import torch

sys.path.insert(0, ".")
from oracle import eagle3_oracle as O  # noqa: E402  (vocab mapping generator only)
from specforge_amd import ops  # noqa: E402

dev = "cuda"
bf = torch.bfloat16


def timed(fn, iters=10, rounds=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / iters)
    return statistics.median(out)


def line(name, ms, nbytes):
    print(json.dumps(dict(name=name, ms=round(ms, 4), bytes=nbytes, GBps=round(nbytes / ms / 1e6, 1))), flush=True)


# ---- teacher_reduce
rows, Vt, Vd, S, T = 4096, 128256, 32000, 2048, 7
B, Spad = rows // S, S + T
z = (torch.randn(rows, Vt, device=dev) * 3).to(bf)
t2d, d2t = O.make_vocab_mapping(Vt, Vd, seed=1)
lm = torch.ones(B, Spad, dtype=torch.int32, device=dev)
tp = torch.empty(B, Spad, Vd, device=dev)
pod, tsum = torch.zeros(B, Spad, device=dev), torch.zeros(B, Spad, device=dev)
ids, pm = torch.zeros(B, Spad, dtype=torch.int64, device=dev), torch.zeros(B, Spad, dtype=torch.int32, device=dev)
d2t_d, t2d_d = d2t.to(dev), t2d.to(torch.uint8).to(dev)
ms = timed(lambda: ops.teacher_reduce(z, Vd=Vd, d2t=d2t_d, t2d_u8=t2d_d, loss_mask_pad=lm, S=S, Spad=Spad, target_p_pad=tp,
                                      pod_scale_pad=pod, tsum_pad=tsum, ids_pad=ids, pos_mask_pad=pm))
line("teacher_reduce 4096x128256 -> 32000", ms, rows * (Vt * 2 + Vd * 4))
del z, tp

# ---- rmsnorm_bwd
R, H = 16384, 4096
x, dy, add = (torch.randn(R, H, device=dev).to(bf) for _ in range(3))
w = torch.randn(H, device=dev).to(bf)
rstd = torch.rand(R, device=dev) + 0.5
dx = torch.empty(R, H, device=dev, dtype=bf)
dw = torch.zeros(H, device=dev)
ws = torch.empty(ops.rmsnorm_bwd_workspace(R, H), device=dev)
ms = timed(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, dx=dx, add=add, dw_acc=dw, dw_accumulate=True, workspace=ws))
line("rmsnorm_bwd 16384x4096 (+add, dw)", ms, R * H * 2 * 4)
ms = timed(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, dx=dx, add=None, dw_acc=dw, dw_accumulate=True, workspace=ws))
line("rmsnorm_bwd 16384x4096 (dw)", ms, R * H * 2 * 3)
del x, dy, add, dx

# ---- fused soft-target CE (forward metrics + in-place gradient), position-mask density 25 % (the synthetic t2d) and 100 %
R, V, Bc = 16384, 32000, 8
Sp = S + T
tp = torch.rand(Bc, Sp, V, device=dev)
tp /= tp.sum(-1, keepdim=True)
pod, tsum = torch.rand(Bc, Sp, device=dev), torch.ones(Bc, Sp, device=dev)
lmask = torch.ones(Bc, Sp, dtype=torch.int32, device=dev)
tids = torch.randint(0, 128256, (Bc, Sp), device=dev)
d2t32 = O.make_vocab_mapping(128256, V, seed=1)[1].to(dev)
rl, rc, ra = (torch.empty(R, device=dev) for _ in range(3))
src = torch.randn(R, V, device=dev).to(bf)
logits = torch.empty_like(src)
for dens in (0.25, 1.0):
    pmask = (torch.rand(Bc, Sp, device=dev) < dens).to(torch.int32)

    def run():
        ops.ce_fused(logits, tp, S=S, Spad=Sp, off=3, pos_mask_pad=pmask, loss_mask_pad=lmask, tgt_ids_pad=tids,
                     pod_scale_pad=pod, tsum_pad=tsum, d2t=d2t32, grad_scale=0.1, write_grad=True, row_loss=rl,
                     row_correct=rc, row_accept=ra)
    logits.copy_(src)
    ms = timed(run)   # in place: later launches read gradients instead of logits -- same bytes, same code path
    line(f"ce_fused 16384x32000 bf16, mask density {dens}", ms, int(R * V * (2 + 2 + 4 * dens)))
