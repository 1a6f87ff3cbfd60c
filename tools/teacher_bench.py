"""teacher reduce at the headline chunk shape (4096 x 128256 logits, Vd 32000): natural layout (gather through d2t) vs permuted
columns (draft sub-vocabulary first), one process"""
import sys, json, torch
sys.path.insert(0, ".")
from specforge_amd import ops
torch.manual_seed(0)
R, Vt, Vd, S, T = 4096, 128256, 32000, 2048, 7
B, Spad = R // S, S + T
dev = "cuda"
z = (torch.randn(R, Vt, device=dev) * 3).to(torch.bfloat16)
ids = torch.randperm(Vt, device=dev)[:Vd].sort().values
t2d = torch.zeros(Vt, dtype=torch.bool, device=dev); t2d[ids] = True
d2t = (ids - torch.arange(Vd, device=dev)).long()
perm = torch.cat([ids, torch.nonzero(~t2d).flatten()])
zp = z[:, perm].contiguous()
lm = torch.ones(B, Spad, dtype=torch.int32, device=dev)
def outs():
    return dict(target_p_pad=torch.empty(B, Spad, Vd, device=dev), pod_scale_pad=torch.empty(B, Spad, device=dev), tsum_pad=torch.empty(B, Spad, device=dev),
                ids_pad=torch.empty(B, Spad, dtype=torch.int64, device=dev), pos_mask_pad=torch.empty(B, Spad, dtype=torch.int32, device=dev))
o1, o2 = outs(), outs()
t8 = t2d.to(torch.uint8); p32 = perm.to(torch.int32)
f1 = lambda: ops.teacher_reduce(z, Vd=Vd, d2t=d2t, t2d_u8=t8, loss_mask_pad=lm, S=S, Spad=Spad, **o1)
f2 = lambda: ops.teacher_reduce_perm(zp, Vt=Vt, Vd=Vd, perm=p32, t2d_u8=t8, loss_mask_pad=lm, S=S, Spad=Spad, **o2)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
res = {"natural_ms": round(t(f1), 4), "permuted_ms": round(t(f2), 4)}
res["ids_equal"] = bool(torch.equal(o1["ids_pad"][:, :S], o2["ids_pad"][:, :S]))
res["pm_equal"] = bool(torch.equal(o1["pos_mask_pad"][:, :S], o2["pos_mask_pad"][:, :S]))
res["tp_maxdiff"] = float((o1["target_p_pad"][:, :S] - o2["target_p_pad"][:, :S]).abs().max())
res["pod_maxrel"] = float(((o1["pod_scale_pad"][:, :S] - o2["pod_scale_pad"][:, :S]).abs() / o1["pod_scale_pad"][:, :S].abs()).max())
gb = (R * Vt * 2 + R * Vd * 4) / 1e9
res["permuted_TBps"] = round(gb / res["permuted_ms"], 2); res["natural_TBps"] = round(gb / res["natural_ms"], 2)
# the whole teacher chunk: head GEMM + reduce, every logit stored vs the reduction epilogue
K = 4096
x = torch.randn(R, K, device=dev).to(torch.bfloat16); w = (torch.randn(Vt, K, device=dev) * (4.0 / 64)).to(torch.bfloat16)
part = torch.empty(R, (Vt - Vd + 127) // 128, 4, device=dev)
def full():
    vz, n = ops.gemm_nt_teacher(x, w, zp, None, Vd=Vd)
    ops.teacher_reduce_perm(zp, Vt=Vt, Vd=Vd, perm=p32, t2d_u8=t8, loss_mask_pad=lm, S=S, Spad=Spad, **o1)
def fused():
    vz, n = ops.gemm_nt_teacher(x, w, zp, part, Vd=Vd)
    ops.teacher_reduce_perm(zp[:, :vz], Vt=Vt, Vd=Vd, perm=p32, t2d_u8=t8, loss_mask_pad=lm, S=S, Spad=Spad, part=part, nparts=n, **o2)
for rnd in range(2):
    res.setdefault("gemm_store_all_ms", []).append(round(t(lambda: ops.gemm_nt_teacher(x, w, zp, None, Vd=Vd)), 4))
    res.setdefault("gemm_reduce_epilogue_ms", []).append(round(t(lambda: ops.gemm_nt_teacher(x, w, zp, part, Vd=Vd)), 4))
    res.setdefault("chunk_store_all_ms", []).append(round(t(full), 4))
    res.setdefault("chunk_fused_ms", []).append(round(t(fused), 4))
res["fused_ids_equal"] = bool(torch.equal(o1["ids_pad"][:, :S], o2["ids_pad"][:, :S]))
print(json.dumps(res))
