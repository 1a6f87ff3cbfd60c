"""TTT attention kernels vs the oracle's sdpa-backend restatement
(oracle/eagle3_oracle.py::ttt_attention <- llama3_eagle.py:745-778), forward and all
gradients (dq, dK/dV of block 0, dK_i/dV_i of the diagonal branches), GQA, right padding,
sequence lengths that are not multiples of the tile sizes.  Mirrors the reference's
backend-vs-sdpa parity tests (tests/test_utils/test_flex_attention.py:47-242, atol/rtol 1e-2
forward; tests/test_utils/test_flash_attention.py:35-42 for bf16).
"""
import math

import pytest
import torch

from oracle import eagle3_oracle as O
from specforge_amd import ops


def _mk(B, S, nh, nkv, hd, nsteps, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, S, nh * hd, generator=g).to(torch.bfloat16)
    ks = [torch.randn(B, S, nkv * hd, generator=g).to(torch.bfloat16) for _ in range(nsteps)]
    vs = [torch.randn(B, S, nkv * hd, generator=g).to(torch.bfloat16) for _ in range(nsteps)]
    do = torch.randn(B, S, nh * hd, generator=g).to(torch.bfloat16)
    return q, ks, vs, do


def _oracle(q, ks, vs, do, B, S, nh, nkv, hd, lengths, device="cpu"):
    """fp32 oracle on the bf16-rounded inputs (``device``: the long-sequence cases run the same restatement on the GPU -- [B, nh, S, S + k]
    fp32 scores with autograd are tens of GB and minutes on the host at S 8192); results come back as CPU tensors"""
    lead = lambda t, n: t.to(device).float().view(B, S, n, hd).transpose(1, 2)
    qf = lead(q, nh).requires_grad_(True)
    kf = [lead(k, nkv).requires_grad_(True) for k in ks]
    vf = [lead(v, nkv).requires_grad_(True) for v in vs]
    am = torch.zeros(B, S, dtype=torch.long)
    for b, L in enumerate(lengths):
        am[b, :L] = 1
    add_mask = O.additive_attention_mask(am.bool(), S, torch.float32).to(device)
    rep = nh // nkv
    out = O.ttt_attention(qf, [O.repeat_kv(k, rep) for k in kf], [O.repeat_kv(v, rep) for v in vf], add_mask, hd)
    out.backward(lead(do, nh))
    flat = lambda t: t.transpose(1, 2).reshape(B * S, -1).cpu()
    return flat(out.detach()), flat(qf.grad), [flat(k.grad) for k in kf], [flat(v.grad) for v in vf]


@pytest.mark.parametrize("hd", [64, 128, 256])
@pytest.mark.parametrize("B,S,nh,nkv,lengths,nsteps", [
    (2, 48, 4, 2, [48, 23], 1),
    (2, 48, 4, 2, [48, 23], 3),
    (1, 200, 2, 1, [200], 4),
    (1, 136, 2, 2, [130], 7),
    (2, 13, 2, 1, [13, 7], 3),      # S not a multiple of 8 (the collator pads to the longest sample, whatever it is)
    (1, 75, 4, 2, [75], 2),
])
def test_ttt_attention_fwd_bwd(backend, hd, B, S, nh, nkv, lengths, nsteps):
    q, ks, vs, do = _mk(B, S, nh, nkv, hd, nsteps, seed=hd + S)
    o_ref, dq_ref, dk_ref, dv_ref = _oracle(q, ks, vs, do, B, S, nh, nkv, hd, lengths)
    d = lambda t: t.to(backend)
    scale = 1.0 / math.sqrt(hd)
    N = B * S
    # the kernels see [B*S, *] views of wider (fused qkv) buffers
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
    torch.testing.assert_close(o.float().cpu(), o_ref, rtol=2e-2, atol=2e-2)

    # backward
    dout = d(do.view(N, -1))
    delta = torch.empty(B, nh, S, device=backend)
    dq_init = torch.zeros(N, nh * hd, device=backend) if nsteps > 1 else None
    dk_acc = [torch.zeros(N, nkv * hd, device=backend) for _ in range(nsteps)]
    dv_acc = [torch.zeros(N, nkv * hd, device=backend) for _ in range(nsteps)]
    ops.attn_bwd_pre(qv, o, dout, kview[1:], vview[1:], dk_acc[1:], dv_acc[1:], lse, delta, dq_init, B=B, S=S, nh=nh,
                     nkv=nkv, hd=hd, scale=scale)
    dq = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend)
    ops.attn_bwd_dq(qv, dout, kview[0], vview[0], kv_len, lse, delta, dq_init, dq, B=B, S=S, nh=nh, nkv=nkv, hd=hd,
                    scale=scale)
    ops.attn_bwd_dkv(qv, dout, kview[0], vview[0], kv_len, lse, delta, dk_acc[0], dv_acc[0], B=B, S=S, nh=nh,
                     nkv=nkv, hd=hd, scale=scale)

    def close(got, ref, what):
        tol = 3e-2 * float(ref.abs().max()) + 1e-6
        err = float((got.float().cpu() - ref).abs().max())
        assert err <= tol, (what, err, tol)

    close(dq, dq_ref, "dq")
    for i in range(nsteps):
        close(dk_acc[i], dk_ref[i], f"dk{i}")
        close(dv_acc[i], dv_ref[i], f"dv{i}")
    # padded keys of block 0 receive exactly zero gradient
    for b, L in enumerate(lengths):
        assert float(dk_acc[0].view(B, S, -1)[b, L:].abs().max() if L < S else 0.0) == 0.0
        assert float(dv_acc[0].view(B, S, -1)[b, L:].abs().max() if L < S else 0.0) == 0.0
    # the finished-gradient form: the last branch's sums (old fp32 value + this launch's terms) leave as bf16, rounded once,
    # and its fp32 accumulators are not written -- bit-identical to accumulating in fp32 and casting afterwards
    if nsteps > 1:
        g = torch.Generator().manual_seed(77)
        seed_k = [torch.randn(N, nkv * hd, generator=g).to(backend) for _ in range(nsteps)]
        seed_v = [torch.randn(N, nkv * hd, generator=g).to(backend) for _ in range(nsteps)]
        acc_k, acc_v = [t.clone() for t in seed_k], [t.clone() for t in seed_v]
        ops.attn_bwd_pre(qv, o, dout, kview[1:], vview[1:], acc_k[1:], acc_v[1:], lse, delta, torch.zeros_like(dq_init), B=B, S=S,
                         nh=nh, nkv=nkv, hd=hd, scale=scale)
        fin_k, fin_v = [t.clone() for t in seed_k], [t.clone() for t in seed_v]
        wide = torch.full((N, 2 * nkv * hd + 16), 3.0, dtype=torch.bfloat16, device=backend)     # strided bf16 outputs
        ok, ov = wide[:, 8:8 + nkv * hd], wide[:, 8 + nkv * hd:8 + 2 * nkv * hd]
        ops.attn_bwd_pre(qv, o, dout, kview[1:], vview[1:], fin_k[1:], fin_v[1:], lse, delta, torch.zeros_like(dq_init), B=B, S=S,
                         nh=nh, nkv=nkv, hd=hd, scale=scale, dk_last=ok, dv_last=ov)
        assert torch.equal(ok.cpu(), acc_k[-1].to(torch.bfloat16).cpu()) and torch.equal(ov.cpu(), acc_v[-1].to(torch.bfloat16).cpu())
        assert torch.equal(fin_k[-1].cpu(), seed_k[-1].cpu()) and torch.equal(fin_v[-1].cpu(), seed_v[-1].cpu())   # untouched
        for i in range(1, nsteps - 1):
            assert torch.equal(fin_k[i].cpu(), acc_k[i].cpu()) and torch.equal(fin_v[i].cpu(), acc_v[i].cpu())
        assert float((wide[:, :8].float() - 3.0).abs().max()) == 0.0 and float((wide[:, 8 + 2 * nkv * hd:].float() - 3.0).abs().max()) == 0.0
    # accumulation semantics: a second backward call adds into the fp32 buffers
    before = dk_acc[0].clone()
    ops.attn_bwd_dkv(qv, dout, kview[0], vview[0], kv_len, lse, delta, dk_acc[0], dv_acc[0], B=B, S=S, nh=nh,
                     nkv=nkv, hd=hd, scale=scale)
    torch.testing.assert_close(dk_acc[0].cpu(), 2 * before.cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hd,grow", [(64, "both"), (128, "block1"), (128, "both"), (256, "block1"), (256, "block0"), (256, "both")])
def test_forward_online_softmax_rescale_paths(backend, hd, grow):
    """Scores that keep growing along the keys, so the running row max rises by more than 2^8 again and again: in the SECOND 32-key block of
    a tile (the head_dim-256 kernel takes that block's exponentials speculatively and must redo them after draining the PV products
    of the first block -- sf_attn_w1.hip fixup1), in the first block of a tile (the deferred rescale), or in both; plus the backward
    kernels on the lse that comes out of it."""
    B, S, nh, nkv, nsteps = 1, 330, 2, 1, 2
    lengths = [301]
    q, ks, vs, do = _mk(B, S, nh, nkv, hd, nsteps, seed=17 + hd)
    blk = torch.arange(S) // 32
    on = {"block1": blk % 2 == 1, "block0": blk % 2 == 0, "both": torch.ones(S, dtype=torch.bool)}[grow]
    gain = 1.0 + 60.0 * torch.cumsum(on.float() * (torch.arange(S) % 32 == 0).float(), 0)    # +60 per selected 32-key block: the row max rises by ~100 nats each time -- an exponential taken against the old max overflows
    k0 = (ks[0].float().view(B, S, nkv, hd) * gain.view(1, S, 1, 1)).to(torch.bfloat16).view(B, S, nkv * hd)
    ks = [k0] + ks[1:]
    if grow == "both":      # ... and a diagonal branch whose scores tower over the block-0 maximum for half of the rows (the epilogue's rescale)
        ks[1] = (ks[1].float() * 400.0).to(torch.bfloat16)
    o_ref, dq_ref, dk_ref, dv_ref = _oracle(q, ks, vs, do, B, S, nh, nkv, hd, lengths)
    d = lambda t: t.to(backend)
    N, scale = B * S, 1.0 / math.sqrt(hd)
    qv, dout = d(q.view(N, -1)), d(do.view(N, -1))
    K, V = [d(t.view(N, -1)) for t in ks], [d(t.view(N, -1)) for t in vs]
    kv_len = d(torch.tensor(lengths, dtype=torch.int32))
    o = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend)
    lse = torch.empty(B, nh, S, device=backend)
    ops.attn_fwd(qv, K[0], V[0], K[1:], V[1:], kv_len, o, lse, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    torch.testing.assert_close(o.float().cpu(), o_ref, rtol=2e-2, atol=2e-2)
    delta = torch.empty(B, nh, S, device=backend)
    dq_init = torch.zeros(N, nh * hd, device=backend)
    dk_acc = [torch.zeros(N, nkv * hd, device=backend) for _ in range(nsteps)]
    dv_acc = [torch.zeros(N, nkv * hd, device=backend) for _ in range(nsteps)]
    ops.attn_bwd_pre(qv, o, dout, K[1:], V[1:], dk_acc[1:], dv_acc[1:], lse, delta, dq_init, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    dq = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend)
    ops.attn_bwd_dq(qv, dout, K[0], V[0], kv_len, lse, delta, dq_init, dq, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    ops.attn_bwd_dkv(qv, dout, K[0], V[0], kv_len, lse, delta, dk_acc[0], dv_acc[0], B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    for got, ref, what in ((dq, dq_ref, "dq"), (dk_acc[0], dk_ref[0], "dk0"), (dv_acc[0], dv_ref[0], "dv0")):
        assert float((got.float().cpu() - ref).abs().max()) <= 3e-2 * float(ref.abs().max()) + 1e-6, what


@pytest.mark.parametrize("hd", [64, 128, 256])
def test_dkv_head_split_equals_unsplit(backend, hd):
    """B * nkv * ceil(S / 128) < 512 (a bs 1 recipe): the dK/dV kernel divides the query heads of a kv group over several
    workgroups whose partial sums go through a workspace (sf_attn_bwd_dkv, ABI 5) -- same gradients as the unsplit launch
    (fp32 summation order aside), run-to-run bit-identical, rows at / after kv_len untouched, accumulation (+=) kept."""
    B, S, nh, nkv, nsteps = 2, 150, 8, 2, 1
    lengths = [150, 97]
    q, ks, vs, do = _mk(B, S, nh, nkv, hd, nsteps, seed=hd)
    _, _, dk_ref, dv_ref = _oracle(q, ks, vs, do, B, S, nh, nkv, hd, lengths)
    d = lambda t: t.to(backend)
    N, scale = B * S, 1.0 / math.sqrt(hd)
    qv, k0, v0, dout = d(q.view(N, -1)), d(ks[0].view(N, -1)), d(vs[0].view(N, -1)), d(do.view(N, -1))
    kv_len = d(torch.tensor(lengths, dtype=torch.int32))
    o = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend)
    lse = torch.empty(B, nh, S, device=backend)
    ops.attn_fwd(qv, k0, v0, [], [], kv_len, o, lse, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    delta = torch.empty(B, nh, S, device=backend)
    ops.attn_bwd_pre(qv, o, dout, [], [], [], [], lse, delta, None, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    nws = ops.attn_bwd_dkv_workspace_floats(B, S, nh, nkv, hd)
    assert nws == 2 * 4 * N * nkv * hd                 # 2 key blocks x 2 kv heads x 2 samples = 8 workgroups: all 4 heads of a group split
    assert ops.attn_bwd_dkv_workspace_floats(8, 2048, 32, 8, hd) == 0    # the headline shape fills the chip unsplit
    runs = []
    for ws in (None, torch.full((nws,), float("nan"), device=backend), torch.full((nws,), 7.0, device=backend)):
        dk = torch.full((N, nkv * hd), 0.5, device=backend)      # += semantics: the seed must survive
        dv = torch.full((N, nkv * hd), -0.25, device=backend)
        ops.attn_bwd_dkv(qv, dout, k0, v0, kv_len, lse, delta, dk, dv, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale, workspace=ws)
        runs.append((dk.cpu(), dv.cpu()))
    (dk0, dv0), (dk1, dv1), (dk2, dv2) = runs
    assert torch.equal(dk1, dk2) and torch.equal(dv1, dv2)       # whatever the workspace held; run-to-run identical
    torch.testing.assert_close(dk1, dk0, rtol=1e-5, atol=1e-5 * float(dk0.abs().max()))
    torch.testing.assert_close(dv1, dv0, rtol=1e-5, atol=1e-5 * float(dv0.abs().max()))
    tol = lambda r: 2e-2 * float(r.abs().max())
    assert float((dk1 - 0.5 - dk_ref[0]).abs().max()) <= tol(dk_ref[0]) and float((dv1 + 0.25 - dv_ref[0]).abs().max()) <= tol(dv_ref[0])
    assert float((dk1.view(B, S, -1)[1, 97:] - 0.5).abs().max()) == 0.0     # padded keys of sample 1: nothing added


@pytest.mark.parametrize("hd,T", [(64, 7), (128, 10), (128, 4), (256, 6)])
def test_blocked_diagonal_backward_follows_the_plan(backend, hd, T):
    """sf_attn_bwd_diag driven by engine.diag_plan over a whole backward sweep of T TTT steps (step k: its own q_k / dO_k, branches
    1..k): every dq_k and every finished dK_i / dV_i (bf16, written at sweep step i) against the sum of the per-step oracle
    gradients.  T = 10: two full blocks + one partial, steps with more than 6 branches (dq in two launches); T = 7 is the
    headline's plan (block {1..4} gathered at step 4 with steps 5, 6 streamed)."""
    from specforge_amd.engine import diag_plan

    B, S, nh, nkv = 2, 40, 4, 2
    lengths = [40, 27]
    N, scale = B * S, 1.0 / math.sqrt(hd)
    g = torch.Generator().manual_seed(100 * hd + T)
    ks = [torch.randn(B, S, nkv * hd, generator=g).to(torch.bfloat16) for _ in range(T)]
    vs = [torch.randn(B, S, nkv * hd, generator=g).to(torch.bfloat16) for _ in range(T)]
    qs = [torch.randn(B, S, nh * hd, generator=g).to(torch.bfloat16) for _ in range(T)]
    dos = [torch.randn(B, S, nh * hd, generator=g).to(torch.bfloat16) for _ in range(T)]
    dq_ref, dk_ref, dv_ref = [], [torch.zeros(N, nkv * hd) for _ in range(T)], [torch.zeros(N, nkv * hd) for _ in range(T)]
    for k in range(T):
        _, dq, dk, dv = _oracle(qs[k], ks[:k + 1], vs[:k + 1], dos[k], B, S, nh, nkv, hd, lengths)
        dq_ref.append(dq)
        for i in range(k + 1):
            dk_ref[i] += dk[i]
            dv_ref[i] += dv[i]
    d = lambda t: t.to(backend)
    kv_len = d(torch.tensor(lengths, dtype=torch.int32))
    K = [d(t.view(N, -1)) for t in ks]
    V = [d(t.view(N, -1)) for t in vs]
    Q = [d(t.view(N, -1)) for t in qs]
    DO = [d(t.view(N, -1)) for t in dos]
    O = [torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend) for _ in range(T)]
    LSE = [torch.empty(B, nh, S, device=backend) for _ in range(T)]
    for k in range(T):
        ops.attn_fwd(Q[k], K[0], V[0], K[1:k + 1], V[1:k + 1], kv_len, O[k], LSE[k], B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    DELTA = [torch.full((B, nh, S), float("nan"), device=backend) for _ in range(T)]
    dq_init = torch.full((N, nh * hd), float("nan"), device=backend)
    acc_k = [torch.full((N, nkv * hd), float("nan"), device=backend) for _ in range(T)]   # never zeroed: first touch must not read them
    acc_v = [torch.full((N, nkv * hd), float("nan"), device=backend) for _ in range(T)]
    out_k = [torch.full((N, nkv * hd), float("nan"), dtype=torch.bfloat16, device=backend) for _ in range(T)]
    out_v = [torch.full((N, nkv * hd), float("nan"), dtype=torch.bfloat16, device=backend) for _ in range(T)]
    plan = diag_plan(T)
    assert all(len(plan[s]) == 1 for s in range(min(T, 7)))          # up to 6 branches and 8 streamed steps: one launch per step
    nlaunch = 0
    for s in range(T - 1, -1, -1):
        for L in plan[s]:
            rd, na, fin = L["read"], L["nacc"], L["final"]
            own = L["own"]
            ops.attn_bwd_diag(q=Q[s] if own else None, o=O[s] if own else None, dout=DO[s] if own else None, lse=LSE[s] if own else None,
                              delta=DELTA[s] if own else None, dq_init=dq_init if (own and rd) else None, dq_accumulate=L["dq_accumulate"],
                              kd=[K[i] for i in rd], vd=[V[i] for i in rd],
                              dkd=[None if (L["first"][j] and fin[j]) else acc_k[rd[j]] for j in range(na)],
                              dvd=[None if (L["first"][j] and fin[j]) else acc_v[rd[j]] for j in range(na)], first=L["first"],
                              dk_out=[out_k[rd[j]] if fin[j] else None for j in range(na)],
                              dv_out=[out_v[rd[j]] if fin[j] else None for j in range(na)],
                              xq=[Q[x] for x in L["stream"]], xdo=[DO[x] for x in L["stream"]], xlse=[LSE[x] for x in L["stream"]],
                              xdelta=[DELTA[x] for x in L["stream"]], B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
            nlaunch += 1
        dq = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=backend)
        ops.attn_bwd_dq(Q[s], DO[s], K[0], V[0], kv_len, LSE[s], DELTA[s], dq_init if s > 0 else None, dq, B=B, S=S, nh=nh, nkv=nkv,
                        hd=hd, scale=scale)
        tol = 3e-2 * float(dq_ref[s].abs().max())
        assert float((dq.float().cpu() - dq_ref[s]).abs().max()) <= tol, ("dq", s)
        if s >= 1:       # branch s is final at sweep step s: its bf16 gradient has been written
            for got, ref, what in ((out_k[s], dk_ref[s], "dk"), (out_v[s], dv_ref[s], "dv")):
                err = float((got.float().cpu() - ref).abs().max())
                assert err <= 3e-2 * float(ref.abs().max()) + 1e-6, (what, s, err)
    if T == 7:
        assert nlaunch == 7 and [len(L["stream"]) for s in (6, 5, 4, 3) for L in plan[s]] == [0, 0, 2, 0]


def test_diag_plan_covers_every_pair_once_within_its_window():
    """engine.diag_plan for every ttt_length the engine accepts (1 .. 33): each pair (step k, branch i <= k) adds to the branch's
    sums exactly once, at a sweep step in [i, k] (dK_i / dV_i are final at sweep step i, q_k / dO_k exist from sweep step k on);
    the first launch touching a branch is first-touch and no later one is; the branch turns final exactly once, at sweep step i, in
    the LAST launch that touches it; dq of every step reads each of its branches exactly once (first launch writes, later ones
    accumulate); kernel limits (branches read / accumulating / steps streamed per launch) hold."""
    from specforge_amd.engine import diag_plan

    for T in range(1, ops.MAX_DIAG + 2):
        plan = diag_plan(T)
        assert sorted(plan) == list(range(T))
        pairs, touched, final_at = {}, set(), {}
        for s in range(T - 1, -1, -1):                      # the sweep order
            seen_reads = []
            for li, L in enumerate(plan[s]):
                rd, na = L["read"], L["nacc"]
                assert len(rd) <= ops.DIAG_READ and na <= min(ops.DIAG_ACC, len(rd)) and len(L["stream"]) <= ops.DIAG_X
                assert len(L["first"]) == len(L["final"]) == na and all(1 <= i <= s for i in rd)
                if L["own"]:
                    seen_reads += rd
                    assert L["dq_accumulate"] == (li > 0 and any(M["own"] for M in plan[s][:li]))
                else:
                    assert L["stream"] and not L["dq_accumulate"]
                steps = ([s] if L["own"] else []) + list(L["stream"])
                assert all(s < x < T for x in L["stream"])
                for j in range(na):
                    i = rd[j]
                    assert i not in final_at, (T, s, i, "touched after it went final")
                    assert L["first"][j] == (i not in touched), (T, s, i)
                    touched.add(i)
                    for k in steps:
                        assert i <= s <= k and (k, i) not in pairs, (T, k, i, s)
                        pairs[(k, i)] = s
                    if L["final"][j]:
                        assert s == i
                        final_at[i] = (s, li)
            assert sorted(seen_reads) == list(range(1, s + 1)), (T, s, seen_reads)      # dq of step s: every branch once
        assert set(pairs) == {(k, i) for k in range(1, T) for i in range(1, k + 1)}, T
        assert sorted(final_at) == list(range(1, T))
        for i, (s, li) in final_at.items():                  # ... and final in the last launch of its step that accumulates it
            assert not any(i in M["read"][:M["nacc"]] for M in plan[s][li + 1:])
