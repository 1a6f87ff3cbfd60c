"""The kernels the headline step actually dispatches, at the headline shapes (Llama-3-8B draft, 8 x 2048 tokens,
ttt 7), each against a plain fp32 reference computed with torch on the same GPU:

* ``gemm_nt_256w4_kernel`` (taken when ceil(M/256)*ceil(N/256) >= 256 and K >= 512): lm_head (16384, 32000, 4096),
  down-proj (16384, 4096, 14336), the row-addend form of the QKV projection (16384, 6144, 4096);
* ``gemm_tn_256w4_kernel`` at K = T*N = 114 688 for the lm_head and down-proj weight gradients, incl. the AUTOMATIC
  deterministic 2-way split-K (down: 16 x 56 = 896 tiles = 3.5 rounds of 256 CUs);
* bitwise run-to-run determinism of those kernels over 26 launches (they carry hand-managed hazards);
* TTT attention at S in {1024, 2048} with right padding and 6 diagonal branches (the reference's own backend-parity
  test goes to 2048: tests/test_utils/test_flex_attention.py:44-242) -- the heaviest-first 1-D dispatch and the
  multi-block causal bookkeeping only exist at these lengths;
* the fused CE at (8*2048, 32000) bf16 with a DENSE position mask (every row reads its soft target).
GPU only (the SIMT interpreter would take hours at these sizes).
"""
import math

import pytest
import torch

from specforge_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _randn(shape, seed, dtype=torch.bfloat16, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, device=DEV, generator=g) * scale).to(dtype)


def _check(out, ref, K, out_dtype, what):
    tol = 1e-3 if out_dtype == torch.float32 else 2e-2
    err = (out.float() - ref).abs()
    rel_max = float(err.max() / ref.abs().max())
    print(f"\n[{what}] max|err|/max|ref| = {rel_max:.3e}")
    torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol * math.sqrt(K))
    assert rel_max < (2e-4 if out_dtype == torch.float32 else 8e-3), rel_max


@pytest.mark.parametrize("M,N,K", [(16384, 32000, 4096), (16384, 4096, 14336), (16384, 28672, 4096),
                                   (4096, 128256, 4096),   # one teacher chunk: target hidden x lm_head (Vt = 128 256; 16 x 501 tiles)
                                   (16448, 6144, 4096),    # the embedding half: 65 x 24 tiles, a row of EDGE tiles inside the persistent walk
                                   (16390, 8456, 1024)])   # ragged M and N (N % 8 == 0), A-first plan, 65 x 34 tiles
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt_headline_shapes(M, N, K, out_dtype):
    a, b = _randn((M, K), 1), _randn((N, K), 2)
    ref = a.float() @ b.float().t()
    out = torch.full((M, N), 7.0, dtype=out_dtype, device=DEV)
    ops.gemm_nt(a, b, out)
    _check(out, ref, K, out_dtype, f"nt {M}x{N}x{K}")
    if out_dtype == torch.bfloat16 and N == 4096:
        res = _randn((M, N), 3)
        ops.gemm_nt(a, b, out, residual=res)      # the down-proj / o-proj epilogue: + residual after the rounding
        torch.testing.assert_close(out.float(), (ref.to(torch.bfloat16) + res).float(), rtol=2e-2, atol=2e-2 * math.sqrt(K))


def test_gemm_nt_swiglu_bwd_headline_shape():
    """the down-projection input gradient with d(SwiGLU) fused (16384 x 14336 x 4096: 64 x 56 whole tiles, persistent walk)
    against gemm_nt + swiglu_bwd on the same operands"""
    M, I, K = 16384, 14336, 4096
    dy, w = _randn((M, K), 1), _randn((I, K), 2)
    gu = _randn((M, 2 * I), 3, scale=2.0)
    dact = torch.empty((M, I), dtype=torch.bfloat16, device=DEV)
    ref = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(dy, w, dact)
    ops.swiglu_bwd(dact, gu, ref)
    out = torch.full((M, 2 * I), 7.0, dtype=torch.bfloat16, device=DEV)
    scratch = torch.full((M, I), 5.0, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_swiglu_bwd(dy, w, gu, out, scratch)
    assert float((scratch == 5.0).float().mean()) == 1.0          # fused: d(act) is never written
    same = float((out == ref).float().mean())
    print(f"\n[swiglu-fused dgrad] identical elements {same:.6f}")
    assert same >= 0.999, same
    torch.testing.assert_close(out.float(), ref.float(), rtol=2 ** -6, atol=1e-30)
    # ... and directly against fp32 torch (not through the repo's own two kernels): d(act) = dy . W^T in fp32 (the kernels
    # round it to bf16 once, as the unfused path does), d(SwiGLU) by autograd of  act = bf16(silu(g)) * u
    blk = 2048
    worst = 0.0
    for m0 in range(0, M, blk):
        da = (dy[m0:m0 + blk].float() @ w.float().t()).to(torch.bfloat16).float()
        g = gu[m0:m0 + blk, :I].float().requires_grad_(True)
        u = gu[m0:m0 + blk, I:].float().requires_grad_(True)
        silu = torch.nn.functional.silu(g)
        act = (silu.detach().to(torch.bfloat16).float() - silu.detach() + silu) * u      # straight-through bf16 rounding of silu(g)
        dg, du = torch.autograd.grad(act, (g, u), da)
        want = torch.cat([dg, du], dim=1)
        got = out[m0:m0 + blk].float()
        worst = max(worst, float((got - want).abs().max() / want.abs().max()))
        torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2 * float(want.abs().max()) / 8)
    print(f"[swiglu-fused dgrad vs fp32 torch] max|err|/max|ref| = {worst:.3e}")
    assert worst < 8e-3, worst


def test_gemm_nt_swiglu_fwd_headline_shape():
    """the fused gate|up projection with SwiGLU in its epilogue (16384 x 2*14336 x 4096: 64 x 112 whole tiles, persistent walk)
    against gemm_nt + swiglu_fwd on the same operands, and act directly against fp32 torch"""
    M, I, K = 16384, 14336, 4096
    x, w = _randn((M, K), 1), _randn((2 * I, K), 2, scale=1.0 / math.sqrt(K))
    gu_ref = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=DEV)
    act_ref = torch.empty((M, I), dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(x, w, gu_ref)
    ops.swiglu_fwd(gu_ref, act_ref)
    gu = torch.full((M, 2 * I), 7.0, dtype=torch.bfloat16, device=DEV)
    act = torch.full((M, I), 5.0, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_swiglu_fwd(x, w, gu, act)
    assert torch.equal(gu, gu_ref)
    same = float((act == act_ref).float().mean())
    print(f"\n[swiglu-fused gate|up] identical act elements {same:.6f}")
    assert same >= 0.999, same
    torch.testing.assert_close(act.float(), act_ref.float(), rtol=2 ** -6, atol=1e-30)
    blk, worst = 2048, 0.0
    for m0 in range(0, M, blk):
        z = x[m0:m0 + blk].float() @ w.float().t()
        want = torch.nn.functional.silu(z[:, :I]) * z[:, I:]
        got = act[m0:m0 + blk].float()
        worst = max(worst, float((got - want).abs().max() / want.abs().max()))
    print(f"[swiglu-fused gate|up vs fp32 torch] max|err|/max|ref| = {worst:.3e}")
    assert worst < 1e-2, worst
    runs = [torch.empty_like(act) for _ in range(3)]
    for o in runs:
        ops.gemm_nt_swiglu_fwd(x, w, gu, o)
    assert all(torch.equal(o, act) for o in runs)          # run-to-run bit-identical


def test_gemm_nt_teacher_headline_chunk():
    """the teacher chunk of the headline step (4096 x 128256 x 4096, Vd 32000): the head GEMM with the reduction epilogue (draft
    columns stored, 752 column blocks reduced) + teacher_reduce_perm vs the same GEMM storing every logit"""
    from oracle import eagle3_oracle as O
    M, Vt, Vd, K, S, T = 4096, 128256, 32000, 4096, 2048, 7
    B, Spad = M // S, S + T
    x, w = _randn((M, K), 1), _randn((Vt, K), 2, scale=4.0 / math.sqrt(K))
    t2d, d2t = O.make_vocab_mapping(Vt, Vd, seed=1)
    t2d, d2t = t2d.to(DEV), d2t.to(DEV)
    perm = torch.cat([torch.arange(Vd, device=DEV) + d2t, torch.nonzero(~t2d).flatten()])
    wp = w[perm].contiguous()
    wp[Vd + 5000] = wp[Vd + 77]                      # identical columns in different reduced blocks
    lm = torch.ones(B, Spad, dtype=torch.int32, device=DEV)
    res = []
    for fused in (False, True):
        z = torch.full((M, Vt), 9.0, dtype=torch.bfloat16, device=DEV)
        part = torch.empty((M, (Vt - Vd + 127) // 128, 4), device=DEV) if fused else None
        vz, nparts = ops.gemm_nt_teacher(x, wp, z, part, Vd=Vd)
        assert (vz, nparts) == ((32000, 752) if fused else (Vt, 0))
        if fused:
            assert float((z[:, vz:].float() - 9.0).abs().max()) == 0.0
        o = dict(target_p_pad=torch.zeros(B, Spad, Vd, device=DEV), pod_scale_pad=torch.zeros(B, Spad, device=DEV),
                 tsum_pad=torch.zeros(B, Spad, device=DEV), ids_pad=torch.zeros(B, Spad, dtype=torch.int64, device=DEV),
                 pos_mask_pad=torch.zeros(B, Spad, dtype=torch.int32, device=DEV))
        ops.teacher_reduce_perm(z[:, :vz], Vt=Vt, Vd=Vd, perm=perm.to(torch.int32), t2d_u8=t2d.to(torch.uint8), loss_mask_pad=lm, S=S,
                                Spad=Spad, part=part, nparts=nparts, **o)
        res.append(o)
        if fused:         # the STORED (draft-range) logits of the reduced form: the plain form's, bit for bit
            assert torch.equal(z[:, :vz], z_plain[:, :vz])
        if not fused:     # the stored logits against fp32 torch, and the ids against torch.argmax on the natural layout
            z_plain = z
            for m0 in range(0, M, 1024):
                zr = x[m0:m0 + 1024].float() @ wp.float().t()
                err = float((z[m0:m0 + 1024].float() - zr).abs().max())
                assert err <= 3e-3 * float(zr.abs().max()) + 1e-3, (m0, err)      # one bf16 rounding of an fp32-accumulated sum
            inv = torch.empty_like(perm); inv[perm] = torch.arange(Vt, device=DEV)
            want = torch.empty(M, dtype=torch.int64, device=DEV)
            for m0 in range(0, M, 1024):
                want[m0:m0 + 1024] = z[m0:m0 + 1024][:, inv].float().argmax(dim=-1)
            assert torch.equal(o["ids_pad"][:, :S].reshape(-1), want)
    a, f = res
    assert torch.equal(a["ids_pad"], f["ids_pad"]) and torch.equal(a["pos_mask_pad"], f["pos_mask_pad"])
    torch.testing.assert_close(f["target_p_pad"], a["target_p_pad"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(f["pod_scale_pad"], a["pod_scale_pad"], rtol=2e-5, atol=1e-9)
    torch.testing.assert_close(f["tsum_pad"], a["tsum_pad"], rtol=1e-6, atol=0)


def test_gemm_nt_rowadd_headline_shape():
    M, N, K, S, T = 16384, 6144, 4096, 2048, 7
    B, Spad = M // S, S + T
    a, b = _randn((M, K), 1), _randn((N, K), 2)
    add = _randn((B * Spad, N), 3, dtype=torch.float32, scale=30.0)
    base = a.float() @ b.float().t()
    for off in (0, 3, T):
        rows = (torch.arange(M, device=DEV) // S) * Spad + torch.arange(M, device=DEV) % S + off
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt_rowadd(a, b, out, add, S=S, Spad=Spad, off=off)
        _check(out, base + add[rows], K, torch.bfloat16, f"rowadd off={off}")


@pytest.mark.parametrize("M,N,name", [(32000, 4096, "lm_head"), (4096, 14336, "down"), (6144, 4096, "qkv-hidden-half")])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_k114688(M, N, name, out_dtype):
    """dW = dY^T . X over all T*N = 114 688 token rows, workspace given -> the kernel picks split-K by itself"""
    K = 7 * 16384
    a, b = _randn((K, M), 1), _randn((K, N), 2)
    ref = torch.empty(M, N, device=DEV)
    for m0 in range(0, M, 8192):                      # fp32 reference in row blocks (bounded temporaries)
        ref[m0:m0 + 8192] = a[:, m0:m0 + 8192].float().t() @ b.float()
    # 2*M*N floats for the split-K partials + the 4096-float tail the launcher needs to switch PACE-KEEPING on: this is
    # the workspace the engine passes (engine.py: tn_ws), i.e. split-K + pacing is what the bench runs and what is tested
    ws = torch.empty(2 * M * N + 4096, dtype=torch.float32, device=DEV)
    out = torch.full((M, N), 7.0, dtype=out_dtype, device=DEV)
    ops.gemm_tn(a, b, out, workspace=ws)
    _check(out, ref, K, out_dtype, f"tn {name} {M}x{N}x{K}")
    ws_np = torch.empty(2 * M * N, dtype=torch.float32, device=DEV)   # too small for the counters: unpaced, same numbers
    out_np = torch.full((M, N), 7.0, dtype=out_dtype, device=DEV)
    ops.gemm_tn(a, b, out_np, workspace=ws_np)
    assert torch.equal(out, out_np), "pace-keeping passes no data: paced and unpaced results must be bit-identical"
    out2 = torch.full((M, N), 1.0, dtype=out_dtype, device=DEV)
    ops.gemm_tn(a, b, out2, alpha=0.5, beta=2.0, workspace=ws)   # accumulation-window form (beta = 1 in the engine)
    _check(out2, 0.5 * ref + 2.0, K, out_dtype, f"tn {name} alpha/beta")
    out3 = torch.full((M, N), 7.0, dtype=out_dtype, device=DEV)
    ops.gemm_tn(a, b, out3)                                      # no workspace -> unsplit
    _check(out3, ref, K, out_dtype, f"tn {name} unsplit")


@pytest.mark.parametrize("form,M,N,K", [("nt", 16384, 4096, 14336), ("nt", 16384, 32000, 4096), ("rowadd", 16384, 6144, 4096),
                                        ("tn", 4096, 14336, 114688), ("tn", 32000, 4096, 114688)])
def test_gemm_bitwise_determinism(form, M, N, K):
    if form == "tn":
        a, b = _randn((K, M), 1), _randn((K, N), 2)
        ws = torch.empty(2 * M * N + 4096, dtype=torch.float32, device=DEV)   # split-K + pace-keeping, as in the step
        f = lambda o: ops.gemm_tn(a, b, o, workspace=ws)
    elif form == "nt":
        a, b = _randn((M, K), 1), _randn((N, K), 2)
        f = lambda o: ops.gemm_nt(a, b, o)
    else:
        a, b = _randn((M, K), 1), _randn((N, K), 2)
        add = _randn((8 * 2055, N), 3, dtype=torch.float32)
        f = lambda o: ops.gemm_nt_rowadd(a, b, o, add, S=2048, Spad=2055, off=5)
    first = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    f(first)
    o = torch.empty_like(first)
    for i in range(25):
        o.fill_(0)
        f(o)
        assert torch.equal(o, first), f"run {i + 1} differs from run 0"


@pytest.mark.parametrize("S,lengths", [(1024, [1024, 651]), (2048, [2048, 1675]), (4096, [4096, 2817]), (8192, [8192, 5003]), (16384, [16384, 9001])])
@pytest.mark.parametrize("nsteps", [1, 7])
def test_ttt_attention_long(S, lengths, nsteps):
    # S 4096 (cfg 4's recipe: bs 1 x 4096): 128 query blocks per (batch, kv head) -- twice what one XCD holds of the
    # pair-major forward / dQ work order, 64 key tiles per query row.  S 8192: the longest `max_length` among the reference's
    # recipes (examples/configs/qwen3.5-35b-a3b-eagle3-online.yaml:11), 256 query blocks per pair, a ragged second sample.  S 16384: twice
    # that (the oracle's [B, nh, S, S + k] fp32 scores with autograd peak at 40 GB on the GPU; S 32768 -- 150 GB -- was run once by hand:
    # profiles/r5_attn_long_s16384_s32768.log)
    _attn_long(S, lengths, nsteps, 128)


@pytest.mark.parametrize("S,lengths", [(1024, [1024, 651]), (4096, [4096, 2817]), (8192, [8192, 5003]), (16384, [16384, 9001])])
@pytest.mark.parametrize("nsteps", [1, 7])
def test_ttt_attention_long_head_dim_256(S, lengths, nsteps):
    """head_dim 256 (gemma3-1b / qwen3-next-80b-a3b / qwen3.5-35b-a3b recipes): one workgroup per CU for forward / dQ, the
    role-split dK/dV kernel, 32 lanes per token row in the diagonal-branch kernel -- at the sequence lengths those recipes train at
    (qwen3.5-35b-a3b-eagle3-{offline,online}.yaml: max_length 4096 / 8192; qwen3-next-80b-a3b-eagle3-online.yaml: 4096)"""
    _attn_long(S, lengths, nsteps, 256)


@pytest.mark.parametrize("hd", [128, 256])
def test_attention_kernels_are_bitwise_reproducible(hd):
    """forward (with diagonal branches), dQ and dK/dV six times over: identical bits (fixed work assignment, no atomics; the head_dim-256
    dK/dV pair hands P over through LDS behind the workgroup barrier -- a race there would show up here)"""
    from tests.test_attention import _mk

    B, S, nh, nkv, nsteps, lengths = 2, 1024, 4, 2, 3, [1024, 777]
    q, ks, vs, do = _mk(B, S, nh, nkv, hd, nsteps, seed=hd)
    N, scale = B * S, 1.0 / math.sqrt(hd)
    d = lambda t: t.to(DEV)
    qv, dout = d(q.view(N, -1)), d(do.view(N, -1))
    K, V = [d(t.view(N, -1)) for t in ks], [d(t.view(N, -1)) for t in vs]
    kv_len = d(torch.tensor(lengths, dtype=torch.int32))
    first = None
    for run in range(6):
        o = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B, nh, S, device=DEV)
        ops.attn_fwd(qv, K[0], V[0], K[1:], V[1:], kv_len, o, lse, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
        delta = torch.empty(B, nh, S, device=DEV)
        dq_init = torch.zeros(N, nh * hd, device=DEV)
        dk = [torch.zeros(N, nkv * hd, device=DEV) for _ in range(nsteps)]
        dv = [torch.zeros(N, nkv * hd, device=DEV) for _ in range(nsteps)]
        ops.attn_bwd_pre(qv, o, dout, K[1:], V[1:], dk[1:], dv[1:], lse, delta, dq_init, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
        dq = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=DEV)
        ops.attn_bwd_dq(qv, dout, K[0], V[0], kv_len, lse, delta, dq_init, dq, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
        ops.attn_bwd_dkv(qv, dout, K[0], V[0], kv_len, lse, delta, dk[0], dv[0], B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
        got = [o, lse, dq, dk[0], dv[0]]
        if first is None:
            first = [t.clone() for t in got]
        else:
            for a, b_, what in zip(got, first, ("o", "lse", "dq", "dk", "dv")):
                assert torch.equal(a, b_), (what, run)


def _attn_long(S, lengths, nsteps, hd):
    from tests.test_attention import _mk, _oracle

    B, nh, nkv = 2, 4, 2
    q, ks, vs, do = _mk(B, S, nh, nkv, hd, nsteps, seed=S + nsteps)
    o_ref, dq_ref, dk_ref, dv_ref = _oracle(q, ks, vs, do, B, S, nh, nkv, hd, lengths, device=DEV if S >= 4096 else "cpu")
    torch.cuda.empty_cache()
    d = lambda t: t.to(DEV)
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
    o = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, nh, S, device=DEV)
    ops.attn_fwd(qv, kview[0], vview[0], kview[1:], vview[1:], kv_len, o, lse, B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
    torch.testing.assert_close(o.float().cpu(), o_ref, rtol=1e-2, atol=1e-2)   # the reference's bar: test_flex_attention.py:124-132
    dout = d(do.view(N, -1))
    delta = torch.empty(B, nh, S, device=DEV)
    dq_init = torch.zeros(N, nh * hd, device=DEV) if nsteps > 1 else None
    dk_acc = [torch.zeros(N, nkv * hd, device=DEV) for _ in range(nsteps)]
    dv_acc = [torch.zeros(N, nkv * hd, device=DEV) for _ in range(nsteps)]
    ops.attn_bwd_pre(qv, o, dout, kview[1:], vview[1:], dk_acc[1:], dv_acc[1:], lse, delta, dq_init, B=B, S=S, nh=nh,
                     nkv=nkv, hd=hd, scale=scale)
    dq = torch.empty(N, nh * hd, dtype=torch.bfloat16, device=DEV)
    ops.attn_bwd_dq(qv, dout, kview[0], vview[0], kv_len, lse, delta, dq_init, dq, B=B, S=S, nh=nh, nkv=nkv, hd=hd,
                    scale=scale)
    ops.attn_bwd_dkv(qv, dout, kview[0], vview[0], kv_len, lse, delta, dk_acc[0], dv_acc[0], B=B, S=S, nh=nh,
                     nkv=nkv, hd=hd, scale=scale)

    def close(got, ref, what):
        tol = 2e-2 * float(ref.abs().max()) + 1e-6
        err = float((got.float().cpu() - ref).abs().max())
        print(f"[attn S={S} k={nsteps}] {what}: err/max = {err / float(ref.abs().max() + 1e-12):.3e}")
        assert err <= tol, (what, err, tol)

    close(dq, dq_ref, "dq")
    for i in range(nsteps):
        close(dk_acc[i], dk_ref[i], f"dk{i}")
        close(dv_acc[i], dv_ref[i], f"dv{i}")
    for bi, L in enumerate(lengths):
        if L < S:
            assert float(dk_acc[0].view(B, S, -1)[bi, L:].abs().max()) == 0.0
            assert float(dv_acc[0].view(B, S, -1)[bi, L:].abs().max()) == 0.0


@pytest.mark.parametrize("density", [1.0, 0.25])
def test_ce_fused_headline_shape(density):
    """(8*2048, 32000) bf16 logits; density 1.0 = every row carries a position mask (reads its soft target)"""
    B, S, V, T, off = 8, 2048, 32000, 7, 3
    Spad, N = S + T, B * S
    g = torch.Generator(device=DEV).manual_seed(5)
    logits = (torch.randn(N, V, device=DEV, generator=g) * 2).to(torch.bfloat16)
    target_pad = torch.softmax(torch.randn(B, Spad, V, device=DEV, generator=g) * 3, -1)
    pos_pad = (torch.rand(B, Spad, device=DEV, generator=g) < density).int()
    lm_pad = torch.ones(B, Spad, dtype=torch.int32, device=DEV)
    pod_pad = torch.rand(B, Spad, device=DEV, generator=g)
    tsum_pad = target_pad.sum(-1)
    d2t = torch.randint(0, 50, (V,), device=DEV, generator=g).sort().values
    ids_pad = torch.randint(0, V + 50, (B, Spad), device=DEV, generator=g)
    sl = lambda t: t[:, off:off + S].reshape(N, *t.shape[2:])
    gs = 0.512 / N
    # fp32 reference in row chunks
    rl, acc, cor, pred = (torch.empty(N, device=DEV) for _ in range(4))
    grad = torch.empty(N, V, device=DEV)
    tgt, pm, pod = sl(target_pad), sl(pos_pad).float(), sl(pod_pad)
    for r0 in range(0, N, 2048):
        r = slice(r0, r0 + 2048)
        x = logits[r].float()
        lp = torch.log_softmax(x, -1)
        sm = lp.exp()
        rl[r] = -(tgt[r] * lp).sum(-1) * pm[r]
        grad[r] = (sm * tgt[r].sum(-1, keepdim=True) - tgt[r]) * pm[r, None] * gs
        acc[r] = torch.minimum(tgt[r] * pod[r, None], sm).sum(-1) * pm[r]
        p = x.argmax(-1)
        pred[r] = p.float()
        cor[r] = ((p + d2t[p]) == sl(ids_pad)[r]).float()
    x = logits.clone()
    row_loss, row_cor, row_acc = (torch.empty(N, device=DEV) for _ in range(3))
    row_pred = torch.empty(N, dtype=torch.int32, device=DEV)
    ops.ce_fused(x, target_pad, S=S, Spad=Spad, off=off, pos_mask_pad=pos_pad, loss_mask_pad=lm_pad, tgt_ids_pad=ids_pad,
                 pod_scale_pad=pod_pad, tsum_pad=tsum_pad, d2t=d2t, grad_scale=gs, row_loss=row_loss, row_correct=row_cor,
                 row_accept=row_acc, row_pred=row_pred)
    assert torch.equal(row_pred.float(), pred)
    assert torch.equal(row_cor, cor)
    tol = 2e-2
    torch.testing.assert_close(row_loss, rl, rtol=tol, atol=tol)
    torch.testing.assert_close(row_acc, acc, rtol=tol, atol=tol)
    torch.testing.assert_close(x.float(), grad, rtol=tol, atol=tol * float(grad.abs().max()))
    assert float(x.float()[pm == 0].abs().max() if density < 1 else 0.0) == 0.0


def test_operands_larger_than_4_gib():
    """operands whose byte size passes 2^32.  The NT GEMM's LDS-DMA descriptors are rebuilt per 256-row tile from a
    64-bit base (only offsets inside a tile are 32-bit; sf_gemm_nt refuses a tile span >= 2 GiB), the fused CE addresses rows with
    64-bit arithmetic: an A operand of 4.7 GB (163 840 x 14 336 bf16) and a logits matrix of 4.6 GB (72 000 x 32 000 bf16) give the
    same results in their LAST rows -- past the 4 GiB mark -- as torch computes there."""
    M, K, N = 163840, 14336, 512
    a = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    assert a.numel() * 2 > (1 << 32)
    g = torch.Generator(device=DEV).manual_seed(3)
    for m0 in range(0, M, 16384):
        a[m0:m0 + 16384] = torch.randn(16384, K, device=DEV, generator=g).to(torch.bfloat16)
    b = _randn((N, K), 4)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, c)
    for m0 in (0, 81920, M - 4096):                    # first rows, the rows around the 2 GiB mark, the last rows (> 4 GiB)
        ref = a[m0:m0 + 4096].float() @ b.float().t()
        _check(c[m0:m0 + 4096], ref, K, torch.bfloat16, f"nt A > 4 GiB rows {m0}")
    del a, c
    torch.cuda.empty_cache()
    # fused CE: rows x V x 2 bytes > 4 GiB; the last rows against an fp32 reference
    B, S, V, T = 36, 2000, 32000, 7
    Spad, Nr = S + T, B * S
    logits = torch.empty(Nr, V, dtype=torch.bfloat16, device=DEV)
    assert logits.numel() * 2 > (1 << 32)
    for r0 in range(0, Nr, 8000):
        logits[r0:r0 + 8000] = (torch.randn(8000, V, device=DEV, generator=g) * 2).to(torch.bfloat16)
    keep = logits[-S:].clone()                         # the last sample's rows (the kernel writes the gradient in place)
    target_pad = torch.zeros(B, Spad, V, device=DEV)
    target_pad[-1, :S] = torch.softmax(torch.randn(S, V, device=DEV, generator=g) * 3, -1)
    pos_pad = torch.zeros(B, Spad, dtype=torch.int32, device=DEV)
    pos_pad[-1, :S] = 1
    lm_pad = torch.ones(B, Spad, dtype=torch.int32, device=DEV)
    rl, acc, cor = (torch.zeros(Nr, device=DEV) for _ in range(3))
    gs = 1.0 / Nr
    tsum_pad = target_pad.sum(-1)
    pod_pad = torch.ones(B, Spad, device=DEV)
    ids_pad = torch.zeros(B, Spad, dtype=torch.int64, device=DEV)
    d2t = torch.zeros(V, dtype=torch.int64, device=DEV)
    ops.ce_fused(logits, target_pad, S=S, Spad=Spad, off=0, pos_mask_pad=pos_pad, loss_mask_pad=lm_pad, tgt_ids_pad=ids_pad,
                 pod_scale_pad=pod_pad, tsum_pad=tsum_pad, d2t=d2t, grad_scale=gs, row_loss=rl, row_correct=cor, row_accept=acc)
    x = keep.float().requires_grad_(True)
    want = -(target_pad[-1, :S] * torch.log_softmax(x, -1)).sum(-1)
    (want.sum() * gs).backward()
    torch.testing.assert_close(rl[-S:], want.detach(), rtol=1e-4, atol=1e-4)
    err = float((logits[-S:].float() - x.grad).abs().max() / x.grad.abs().max())
    assert err < 1e-2, err
    assert float(rl[:-S].abs().max()) == 0.0           # rows without a position mask carry no loss
