import math, sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from specforge_amd import ops
from test_attention import _mk, _oracle

def run(hd, B, S, nh, nkv, lengths, nsteps):
    backend = "cuda"
    q, ks, vs, do = _mk(B, S, nh, nkv, hd, nsteps, seed=hd + S)
    o_ref, dq_ref, dk_ref, dv_ref = _oracle(q, ks, vs, do, B, S, nh, nkv, hd, lengths)
    d = lambda t: t.to(backend)
    scale = 1.0 / math.sqrt(hd)
    N = B * S
    qkv = [torch.zeros(N, (nh + 2 * nkv) * hd, dtype=torch.bfloat16) for _ in range(nsteps)]
    for i in range(nsteps):
        qkv[i][:, nh * hd:(nh + nkv) * hd] = ks[i].view(N, -1)
        qkv[i][:, (nh + nkv) * hd:] = vs[i].view(N, -1)
    qkv[-1][:, :nh * hd] = q.view(N, -1)
    qkv = [d(t) for t in qkv]
    qv = qkv[-1][:, :nh * hd]
    kview = [t[:, nh * hd:(nh + nkv) * hd] for t in qkv]
    vview = [t[:, (nh + nkv) * hd:] for t in qkv]
    kv_len = d(torch.tensor(lengths, dtype=torch.int32))
    o = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend)
    lse = torch.empty(B, nh, S, device=backend)
    ops.attn_fwd(qv, kview[0], vview[0], kview[1:], vview[1:], kv_len, o, lse, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    dout = d(do.view(N, -1))
    delta = torch.empty(B, nh, S, device=backend)
    dq_init = torch.zeros(N, nh * hd, device=backend) if nsteps > 1 else None
    dk_acc = [torch.zeros(N, nkv * hd, device=backend) for _ in range(nsteps)]
    dv_acc = [torch.zeros(N, nkv * hd, device=backend) for _ in range(nsteps)]
    ops.attn_bwd_pre(qv, o, dout, kview[1:], vview[1:], dk_acc[1:], dv_acc[1:], lse, delta, dq_init, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    for rep in range(3):
        dk_acc[0].zero_(); dv_acc[0].zero_()
        ops.attn_bwd_dkv(qv, dout, kview[0], vview[0], kv_len, lse, delta, dk_acc[0], dv_acc[0], B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
        torch.cuda.synchronize()
        for name, got, ref in (("dk0", dk_acc[0], dk_ref[0]), ("dv0", dv_acc[0], dv_ref[0])):
            g = got.float().cpu()
            bad = ~torch.isfinite(g)
            err = (g - ref).abs()
            err[bad] = 0
            msg = f"hd={hd} S={S} nsteps={nsteps} rep={rep} {name}: nonfinite={int(bad.sum())} maxerr={float(err.max()):.4f} refmax={float(ref.abs().max()):.3f}"
            if bad.any():
                idx = bad.nonzero()
                rows = sorted(set(idx[:, 0].tolist()))
                cols = sorted(set(idx[:, 1].tolist()))
                msg += f" rows={rows[:20]}(n={len(rows)}) cols={cols[:20]}(n={len(cols)})"
            big = (err > 3e-2 * float(ref.abs().max()))
            if big.any():
                idx = big.nonzero()
                msg += f" bigerr_rows={sorted(set(idx[:,0].tolist()))[:20]} cols={sorted(set(idx[:,1].tolist()))[:16]}"
            print(msg, flush=True)
    print("finite lse", bool(torch.isfinite(lse).all()), "delta", bool(torch.isfinite(delta).all()))

for hd in (64, 128):
    run(hd, 2, 48, 4, 2, [48, 23], 1)
    run(hd, 1, 136, 2, 2, [130], 7)
    run(hd, 1, 200, 2, 1, [200], 4)
