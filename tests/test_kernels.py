"""Per-kernel parity tests of the C-ABI ops against the CPU oracle / plain torch fp32.

Each test runs twice: ``[emu]`` = the kernel sources under the SIMT interpreter on CPU
tensors (runs everywhere, checks index logic + host glue), ``[gpu]`` = libsfhip.so on a
real MI355X (marked ``gpu``; the parity tests proper, incl. the reference's own CE test
grid shapes, tests/test_utils/test_loss.py:13-39).

Tolerances: integer artefacts bit-exact; fp32 kernels 1e-4 (the reference's own
rtol/atol for its Triton CE, tests/test_utils/test_loss.py:24-30) ... 1e-3 (north_star
fp32 tolerance); bf16 2e-2 (north_star bf16 tolerance).
"""
import math

import pytest
import torch

from oracle import eagle3_oracle as O
from specforge_amd import ops


def _dev(backend, t):
    return t.to(backend)


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (64, 48, 288), (257, 384, 512), (256, 256, 64),
                                   (520, 264, 192), (300, 200, 128), (300, 260, 640), (256, 512, 64 * 11),
                                   (200, 8456, 512),   # N > 8192: the A-first plan of the 4-wave kernel
                                   (1100, 608, 512),   # 5 x 3 tiles: workgroups of the persistent kernel take 2 tiles (interior + edge)
                                   (512, 1280, 512),   # interpreter (8 "CUs"): 2 x 5 tiles = 1 round + 2 -> the last column tile is peeled
                                   (512, 1200, 512)])  # ... and a ragged peeled tile (176 columns)
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt(backend, M, N, K, out_dtype):
    a = _rand((M, K), torch.bfloat16, 1)
    b = _rand((N, K), torch.bfloat16, 2)
    # asymmetric operands catch transposed outputs
    ref = a.float() @ b.float().t()
    out = torch.full((M, N), 7.0, dtype=out_dtype, device=backend)
    ops.gemm_nt(_dev(backend, a), _dev(backend, b), out)
    tol = 1e-3 if out_dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float().cpu(), ref, rtol=tol, atol=tol * math.sqrt(K))


@pytest.mark.parametrize("M,N,K,S,T", [(64, 48, 64, 16, 3), (512, 264, 128, 64, 2), (256, 200, 576, 32, 2),
                                       (768, 520, 512, 64, 2)])   # 3 x 3 tiles of the 4-wave kernel, interior + edge
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt_rowadd(backend, M, N, K, S, T, out_dtype):
    """fp32 row-mapped addend joins the accumulator before the single rounding (embedding half of the TTT QKV)"""
    a = _rand((M, K), torch.bfloat16, 1)
    b = _rand((N, K), torch.bfloat16, 2)
    B, Spad = M // S, S + T
    add = _rand((B * Spad, N), torch.float32, 3, scale=3.0)
    for off in (0, T):
        rows = (torch.arange(M) // S) * Spad + torch.arange(M) % S + off
        ref = a.float() @ b.float().t() + add[rows]
        out = torch.full((M, N), 7.0, dtype=out_dtype, device=backend)
        ops.gemm_nt_rowadd(_dev(backend, a), _dev(backend, b), out, _dev(backend, add), S=S, Spad=Spad, off=off)
        tol = 1e-3 if out_dtype == torch.float32 else 2e-2
        torch.testing.assert_close(out.float().cpu(), ref, rtol=tol, atol=tol * math.sqrt(K))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rows_expand(backend, dtype):
    """sf_rows_expand: the inverse of a row compaction -- compact rows back at their token rows, exact zeros elsewhere (strided views too)"""
    N, C, Nc = 37, 48, 11
    g = torch.Generator().manual_seed(3)
    rows = torch.randperm(N, generator=g)[:Nc].sort().values
    src = _rand((Nc, C + 8), dtype, 4)[:, :C]
    inv = torch.full((N,), -1, dtype=torch.int32)
    inv[rows] = torch.arange(Nc, dtype=torch.int32)
    wide = torch.full((N, C + 16), 7.0, dtype=dtype, device=backend)
    ops.rows_expand(_dev(backend, src.contiguous()), _dev(backend, inv), wide[:, 8:8 + C])
    want = torch.zeros(N, C, dtype=dtype)
    want[rows] = src
    assert torch.equal(wide[:, 8:8 + C].cpu(), want)
    assert float(wide[:, :8].float().min()) == 7.0 and float(wide[:, 8 + C:].float().min()) == 7.0


@pytest.mark.parametrize("M,N,K,S,T,ks", [(300, 512, 512, 100, 3, 2), (256, 512, 1024, 64, 2, 4), (512, 768, 256, 128, 2, 0)])
def test_gemm_nt_rowadd_split_k(backend, M, N, K, S, T, ks):
    """the row-addend form of an under-filled grid (batch-1 recipes): split-K through the workspace, the fp32 addend joins in the
    fixed-order reduce before the single rounding -- the fused epilogue's result up to fp32 summation order; ragged M included"""
    a, b = _rand((M, K), torch.bfloat16, 1), _rand((N, K), torch.bfloat16, 2)
    B, Spad = M // S, S + T
    add = _rand((B * Spad, N), torch.float32, 3, scale=3.0)
    Mpad = (M + 255) // 256 * 256
    for off in (0, T):
        rows = (torch.arange(M) // S) * Spad + torch.arange(M) % S + off
        ref = a.float() @ b.float().t() + add[rows]
        ws = torch.full((4 * Mpad * N,), float("nan"), device=backend)
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=backend)
        ops.gemm_nt_rowadd(_dev(backend, a), _dev(backend, b), out, _dev(backend, add), S=S, Spad=Spad, off=off, workspace=ws)
        fused = torch.empty_like(out)
        ops.gemm_nt_rowadd(_dev(backend, a), _dev(backend, b), fused, _dev(backend, add), S=S, Spad=Spad, off=off)
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-2, atol=2e-2 * math.sqrt(K))
        torch.testing.assert_close(out.float().cpu(), fused.float().cpu(), rtol=1e-2, atol=0.13)      # (one bf16 ulp at |x| ~ 16-32)
        if backend == "cpu":
            assert int(torch.isfinite(ws).sum()) == ks * Mpad * N


@pytest.mark.parametrize("M,N,K", [(64, 48, 64), (200, 136, 128), (256, 256, 64), (520, 264, 192), (136, 1000, 320)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn(backend, M, N, K, out_dtype):
    """weight-gradient form: both operands stored with the contraction index outermost"""
    a = _rand((K, M), torch.bfloat16, 1)
    b = _rand((K, N), torch.bfloat16, 2)
    ref = a.float().t() @ b.float()
    out = torch.full((M, N), 7.0, dtype=out_dtype, device=backend)
    ops.gemm_tn(_dev(backend, a), _dev(backend, b), out)
    tol = 1e-3 if out_dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float().cpu(), ref, rtol=tol, atol=tol * math.sqrt(K))
    # accumulate into a strided view (the fused q|k|v gradient block) with alpha/beta
    wide = torch.ones(M, N + 16, dtype=out_dtype, device=backend)
    ops.gemm_tn(_dev(backend, a), _dev(backend, b), wide[:, 8:8 + N], alpha=0.5, beta=2.0)
    torch.testing.assert_close(wide[:, 8:8 + N].float().cpu(), 0.5 * ref + 2.0, rtol=tol, atol=tol * math.sqrt(K))
    assert float(wide[:, :8].float().min()) == 1.0 and float(wide[:, 8 + N:].float().max()) == 1.0


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_split_k(backend, out_dtype):
    """2-way split of the contraction through the fp32 workspace == the unsplit product (fixed-order reduction)"""
    M, N, K = 136, 200, 4096
    a = _rand((K, M), torch.bfloat16, 1)
    b = _rand((K, N), torch.bfloat16, 2)
    ref = a.float().t() @ b.float()
    ws = torch.empty(2 * M * N, dtype=torch.float32, device=backend)
    out = torch.ones(M, N, dtype=out_dtype, device=backend)
    ops.gemm_tn(_dev(backend, a), _dev(backend, b), out, alpha=0.5, beta=2.0, workspace=ws, ksplit=2)
    with pytest.raises(Exception, match="ksplit = 2 needs"):
        ops.gemm_tn(_dev(backend, a), _dev(backend, b), out, ksplit=2)          # no workspace
    tol = 1e-3 if out_dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float().cpu(), 0.5 * ref + 2.0, rtol=tol, atol=tol * math.sqrt(K))


def test_gemm_nt_epilogues(backend):
    M, N, K = 136, 72, 128
    a, b = _rand((M, K), torch.bfloat16, 3), _rand((N, K), torch.bfloat16, 4)
    res = _rand((M, N), torch.bfloat16, 5)
    c0 = _rand((M, N), torch.float32, 6)
    ref = a.float() @ b.float().t()
    out = c0.clone().to(backend)
    ops.gemm_nt(_dev(backend, a), _dev(backend, b), out, alpha=0.5, beta=2.0)
    torch.testing.assert_close(out.cpu(), 0.5 * ref + 2.0 * c0, rtol=1e-3, atol=1e-2)
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(_dev(backend, a), _dev(backend, b), outb, residual=_dev(backend, res))
    expect = (ref.to(torch.bfloat16) + res).float()
    torch.testing.assert_close(outb.float().cpu(), expect, rtol=2e-2, atol=0.25)
    # strided views (column slices of wider buffers)
    wide = torch.zeros((M, N + 40), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(_dev(backend, a), _dev(backend, b), wide[:, 8:8 + N])
    torch.testing.assert_close(wide[:, 8:8 + N].float().cpu(), ref, rtol=2e-2, atol=0.25)
    assert float(wide[:, :8].abs().max()) == 0 and float(wide[:, 8 + N:].abs().max()) == 0


@pytest.mark.parametrize("M,N,K,ks", [(512, 512, 512, 2), (256, 512, 1024, 4), (512, 768, 256, 0), (520, 512, 512, 0), (300, 512, 512, 2),
                                      (136, 512, 1024, 4),
                                      (256, 512, 640, 4),      # 10 K-tiles in chunks of 3, 3, 3, 1: the last chunk is a single K-tile
                                      (256, 512, 704, 4),      # 11 K-tiles: 3, 3, 3, 2
                                      (300, 512, 640, 2),      # ragged M (two row tiles, the second one padded in the workspace) x 10 K-tiles in 5, 5
                                      (520, 256, 704, 2)])     # three row tiles x one column tile: 11 K-tiles in 6, 5
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt_split_k_for_under_filled_grids(backend, M, N, K, ks, out_dtype):
    """sf_gemm_nt_ws: with at most half as many 256-tiles as CUs (interpreter: 8 "CUs") K is cut into 2 / 4 chunks -- tiles x chunks work
    units of the 4-wave kernel, fp32 partials, fixed-order reduce; the residual joins after the rounding like in sf_gemm_nt.  Shapes that
    do not qualify (too many tiles) and a missing workspace go the usual way; a ragged M (real batches) splits too -- the partials are laid
    out in whole row tiles.  ``ks`` = the split the shape should take."""
    a, b = _rand((M, K), torch.bfloat16, 1), _rand((N, K), torch.bfloat16, 2)
    res = _rand((M, N), torch.bfloat16, 3) if out_dtype == torch.bfloat16 else None
    ref = a.float() @ b.float().t()
    Mpad = (M + 255) // 256 * 256
    ws = torch.full((4 * Mpad * N,), float("nan"), device=backend)
    d = lambda t: None if t is None else _dev(backend, t)
    out = torch.full((M, N), 7.0, dtype=out_dtype, device=backend)
    ops.gemm_nt(d(a), d(b), out, residual=d(res), workspace=ws)
    plain = torch.full((M, N), 7.0, dtype=out_dtype, device=backend)
    ops.gemm_nt(d(a), d(b), plain, residual=d(res))
    want = ref if res is None else (ref.to(torch.bfloat16) + res).float()
    tol = 1e-3 if out_dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float().cpu(), want, rtol=tol, atol=tol * math.sqrt(K))
    torch.testing.assert_close(out.float().cpu(), plain.float().cpu(), rtol=1e-2 if res is not None else 1e-5, atol=0.25 if res is not None else 1e-3)
    used = int(torch.isfinite(ws).sum())                   # the partials that were written say which split ran
    if backend == "cpu":
        assert used == ks * Mpad * N, (used, ks * Mpad * N)
    again = torch.empty_like(out)
    ops.gemm_nt(d(a), d(b), again, residual=d(res), workspace=torch.zeros_like(ws))
    assert torch.equal(again.cpu(), out.cpu())             # run-to-run identical, whatever the workspace held


@pytest.mark.parametrize("M,N,K", [(512, 512, 256),      # whole tiles: the register-transposed epilogue with the residual read at the lane's 16 bytes
                                   (520, 768, 128),      # ragged M: the shifted last row tile (the rows computed twice get the residual once each time)
                                   (512, 600, 128),      # an edge column tile takes the general store
                                   (1100, 608, 192)])    # persistent walk: interior + edge tiles per workgroup
def test_gemm_nt_residual_is_the_plain_output_plus_the_residual(backend, M, N, K):
    """bf16 output with a residual = round(round(a . b^T) + residual): bit for bit the plain GEMM's output followed by a bf16 add, whichever
    epilogue a tile takes (round 6: whole tiles transpose in registers and read the residual as 16 bytes per lane; edge tiles the general
    store); also through column slices of wider buffers (row strides that are not the width) and in place (residual == out)."""
    a, b = _rand((M, K), torch.bfloat16, 11), _rand((N, K), torch.bfloat16, 12)
    res = _rand((M, N), torch.bfloat16, 13, scale=4.0)
    d = lambda t: _dev(backend, t)
    plain = torch.empty((M, N), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(d(a), d(b), plain)
    want = (plain.cpu() + res)
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(d(a), d(b), out, residual=d(res))
    assert torch.equal(out.cpu(), want)
    wide_c = torch.zeros((M, N + 24), dtype=torch.bfloat16, device=backend)
    wide_r = torch.zeros((M, N + 40), dtype=torch.bfloat16, device=backend)
    wide_r[:, 16:16 + N] = d(res)
    ops.gemm_nt(d(a), d(b), wide_c[:, 8:8 + N], residual=wide_r[:, 16:16 + N])
    assert torch.equal(wide_c[:, 8:8 + N].cpu(), want)
    assert float(wide_c[:, :8].abs().max()) == 0 and float(wide_c[:, 8 + N:].abs().max()) == 0
    wide_r[:, 4:4 + N] = d(res)                       # a residual that is only 8-byte aligned: the general store
    out.fill_(7.0)
    ops.gemm_nt(d(a), d(b), out, residual=wide_r[:, 4:4 + N])
    assert torch.equal(out.cpu(), want)
    inplace = d(res).clone()
    ops.gemm_nt(d(a), d(b), inplace, residual=inplace)
    assert torch.equal(inplace.cpu(), want)


def test_gemm_nt_peeled_last_round_epilogues(backend):
    """the peeled column tile of a partly filled last round (interpreter: 2 x 5 tiles on 8 "CUs") carries the same epilogue:
    residual after the rounding (bf16), alpha / beta (fp32)"""
    M, N, K = 512, 1280, 512
    a, b = _rand((M, K), torch.bfloat16, 3), _rand((N, K), torch.bfloat16, 4)
    res = _rand((M, N), torch.bfloat16, 5)
    c0 = _rand((M, N), torch.float32, 6)
    ref = a.float() @ b.float().t()
    out = c0.clone().to(backend)
    ops.gemm_nt(_dev(backend, a), _dev(backend, b), out, alpha=0.5, beta=2.0)
    torch.testing.assert_close(out.cpu(), 0.5 * ref + 2.0 * c0, rtol=1e-3, atol=3e-2)
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(_dev(backend, a), _dev(backend, b), outb, residual=_dev(backend, res))
    expect = (ref.to(torch.bfloat16) + res).float()
    torch.testing.assert_close(outb.float().cpu(), expect, rtol=2e-2, atol=0.5)
    # the row-addend form (round 4: peeled too): the addend's columns move with the peeled C columns
    S, T = 128, 3
    B, Spad = M // S, S + T
    add = _rand((B * Spad, N), torch.float32, 7, scale=3.0)
    rows = (torch.arange(M) // S) * Spad + torch.arange(M) % S + T
    outr = torch.full((M, N), 5.0, dtype=torch.bfloat16, device=backend)
    ops.gemm_nt_rowadd(_dev(backend, a), _dev(backend, b), outr, _dev(backend, add), S=S, Spad=Spad, off=T)
    torch.testing.assert_close(outr.float().cpu(), ref + add[rows], rtol=2e-2, atol=2e-2 * math.sqrt(K))


# ------------------------------------------------------------------ fused CE
def _ce_reference(logits, target, pos_mask, pod_scale, d2t, tgt_ids, loss_mask, gs):
    x = logits.float().clone().requires_grad_(True)
    lp = torch.log_softmax(x, dim=-1)
    row_loss = -(target * lp).sum(-1) * pos_mask
    (row_loss.sum() * gs).backward()
    sm = torch.softmax(logits.float(), -1)
    accept = torch.minimum(target * pod_scale[:, None], sm).sum(-1) * pos_mask
    pred = logits.float().argmax(-1)
    correct = ((pred + d2t[pred]) == tgt_ids).float() * loss_mask
    return row_loss.detach(), x.grad, accept, correct, pred


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,S,V,T,off", [(2, 16, 256, 3, 0), (1, 24, 1000, 4, 2), (2, 8, 4096, 7, 5)])
def test_ce_fused(backend, dtype, tol, B, S, V, T, off):
    Spad = S + T
    g = torch.Generator().manual_seed(7)
    logits = (torch.randn(B * S, V, generator=g) * 2).to(dtype)
    logits[3, 5] = logits[3].max() + 1  # a clear argmax
    logits[4, 9] = logits[4, 2] = logits[4].float().max().to(dtype) + 2  # a tie -> lowest index
    target_pad = torch.softmax(torch.randn(B, Spad, V, generator=g) * 3, -1)
    pos_pad = (torch.rand(B, Spad, generator=g) > 0.3).int()
    lm_pad = (torch.rand(B, Spad, generator=g) > 0.2).int()
    pod_pad = torch.rand(B, Spad, generator=g)
    tsum_pad = target_pad.sum(-1)
    d2t = torch.randint(0, 50, (V,), generator=g).sort().values
    ids_pad = torch.randint(0, V + 50, (B, Spad), generator=g)
    sl = lambda t: t[:, off:off + S].reshape(B * S, *t.shape[2:])
    gs = 0.64 / (B * S)
    rl, grad, acc, cor, pred = _ce_reference(logits, sl(target_pad), sl(pos_pad).float(), sl(pod_pad), d2t, sl(ids_pad),
                                             sl(lm_pad).float(), gs)
    # make some rows provably "correct"
    pr = pred + d2t[pred]
    ids_pad[:, off:off + S] = torch.where(torch.arange(B * S).view(B, S) % 2 == 0, pr.view(B, S), ids_pad[:, off:off + S])
    cor = ((pr == sl(ids_pad)).float() * sl(lm_pad).float())
    d = lambda t: t.to(backend)
    x = d(logits.clone())
    row_loss, row_cor, row_acc = (torch.empty(B * S, device=backend) for _ in range(3))
    row_pred = torch.empty(B * S, dtype=torch.int32, device=backend)
    ops.ce_fused(x, d(target_pad), S=S, Spad=Spad, off=off, pos_mask_pad=d(pos_pad), loss_mask_pad=d(lm_pad),
                 tgt_ids_pad=d(ids_pad), pod_scale_pad=d(pod_pad), tsum_pad=d(tsum_pad), d2t=d(d2t), grad_scale=gs,
                 row_loss=row_loss, row_correct=row_cor, row_accept=row_acc, row_pred=row_pred)
    assert torch.equal(row_pred.cpu().long(), pred)          # integer artefact: bit-exact
    assert torch.equal(row_cor.cpu(), cor)                   # integer-valued
    torch.testing.assert_close(row_loss.cpu(), rl, rtol=tol, atol=tol)
    torch.testing.assert_close(row_acc.cpu(), acc, rtol=tol, atol=tol)
    gtol = tol * float(grad.abs().max())
    torch.testing.assert_close(x.float().cpu(), grad, rtol=tol, atol=gtol)
    # masked rows carry exactly zero gradient (core/loss.py:160-170)
    masked = x.float().cpu()[sl(pos_pad) == 0]
    assert masked.numel() == 0 or float(masked.abs().max()) == 0.0
    # drop-in mode: tsum computed in-kernel, no pod/ids
    x2 = d(logits.clone())
    ops.ce_fused(x2, d(target_pad), S=S, Spad=Spad, off=off, pos_mask_pad=d(pos_pad), loss_mask_pad=d(lm_pad),
                 grad_scale=gs, row_loss=row_loss, row_correct=row_cor, row_accept=row_acc)
    torch.testing.assert_close(x2.float().cpu(), grad, rtol=tol, atol=gtol)
    out = torch.empty(1, device=backend)
    ops.reduce_sum(row_loss, B * S, 1, out, 1.0 / (B * S))
    torch.testing.assert_close(out.cpu()[0], rl.mean(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ LK-loss gradient
@pytest.mark.parametrize("lk,dtype,tol", [("alpha", torch.float32, 1e-3), ("lambda", torch.float32, 1e-3),
                                          ("alpha", torch.bfloat16, 2e-2), ("lambda", torch.bfloat16, 2e-2)])
def test_ce_lk_grad(backend, lk, dtype, tol):
    """d(step_scale * lk_loss)/d(logits) vs autograd through the reference formulas
    (core/lk_loss.py:43-99, eagle3/model.py:74-96) on the same (rounded) logits."""
    B, S, V, T, off = 2, 8, 136, 3, 1
    Spad = S + T
    g = torch.Generator().manual_seed(11)
    logits = (torch.randn(B * S, V, generator=g) * 2).to(dtype)
    target_pad = torch.softmax(torch.randn(B, Spad, V, generator=g) * 3, -1)
    pos_pad = (torch.rand(B, Spad, generator=g) > 0.3).int()
    pod_pad = torch.rand(B, Spad, generator=g)
    tsum_pad = target_pad.sum(-1)
    sl = lambda t: t[:, off:off + S].reshape(B * S, *t.shape[2:])
    kl_scale, kl_decay, step = 0.7, 1.5, 0.64
    # ---- reference maths, autograd
    x = logits.float().clone().requires_grad_(True)
    p, m = sl(target_pad), sl(pos_pad).float()
    q = p * sl(pod_pad)[:, None]
    kl = -(m[:, None] * p * torch.log_softmax(x, -1)).sum(-1).mean()
    a_tok = torch.minimum(q, torch.softmax(x, -1)).sum(-1)
    den = m.sum().clamp_min(1e-8)
    alpha = (a_tok * m).sum() / den
    log_alpha = (torch.where(a_tok > 0, torch.log(a_tok), torch.zeros_like(a_tok)) * m).sum() / den
    if lk == "alpha":
        loss = -log_alpha
    else:
        w = kl_scale * torch.exp(-kl_decay * alpha.detach())
        loss = w * kl + (1 - w) * (1 - alpha)
    (step * loss).backward()
    # ---- kernel
    d = lambda t: t.to(backend)
    xk = d(logits.clone())
    accept_sum = d((a_tok.detach() * m).sum().reshape(1))
    mask_sum = d(m.sum().reshape(1))
    ops.ce_lk_grad(xk, d(target_pad), S=S, Spad=Spad, off=off, pos_mask_pad=d(pos_pad), pod_scale_pad=d(pod_pad),
                   tsum_pad=d(tsum_pad), lk_loss_type=lk, kl_scale=kl_scale, kl_decay=kl_decay, step_scale=step,
                   kl_row_scale=1.0 / (B * S), accept_sum=accept_sum, mask_sum=mask_sum)
    gtol = tol * float(x.grad.abs().max())
    torch.testing.assert_close(xk.float().cpu(), x.grad, rtol=tol, atol=gtol)
    assert float(xk.float().cpu()[sl(pos_pad) == 0].abs().max()) == 0.0   # masked rows: exactly zero


def test_shift_accum_and_split(backend):
    g = torch.Generator().manual_seed(5)
    B, S, T, C = 3, 8, 3, 24
    Spad = S + T
    dst = torch.randn(B * Spad, C, generator=g)
    ref = dst.clone()
    dd = dst.to(backend)
    for off in (0, 2, T):
        src = torch.randn(B * S, C, generator=g).to(torch.bfloat16)
        ops.shift_accum(src.to(backend), dd, B=B, S=S, Spad=Spad, off=off)
        ref.view(B, Spad, C)[:, off:off + S] += src.float().view(B, S, C)
    assert torch.equal(dd.cpu(), ref)   # fp32 adds of exactly representable terms in the same order
    hi = torch.empty(B * Spad, C, dtype=torch.bfloat16, device=backend)
    lo = torch.empty_like(hi)
    ops.split_bf16(dd, hi, lo)
    h = ref.to(torch.bfloat16)
    assert torch.equal(hi.cpu(), h) and torch.equal(lo.cpu(), (ref - h.float()).to(torch.bfloat16))
    err = (hi.cpu().float() + lo.cpu().float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2 ** -15


@pytest.mark.parametrize("B,S,T,C", [(3, 8, 3, 24), (2, 5, 7, 16), (1, 37, 9, 40)])
def test_shift_sum_split_equals_the_per_step_form(backend, B, S, T, C):
    """one pass over the T stacked step gradients == T x shift_accum (off = k, k = T-1 .. 0) + split_bf16, bit for bit"""
    g = torch.Generator().manual_seed(9)
    N, Spad = B * S, S + T
    src = torch.randn(T * N + 5, C, generator=g).to(torch.bfloat16)      # (+ pad rows the kernel must not read)
    dd = torch.zeros(B * Spad, C, device=backend)
    for k in range(T - 1, -1, -1):
        ops.shift_accum(src[k * N:(k + 1) * N].to(backend), dd, B=B, S=S, Spad=Spad, off=k)
    hi_ref = torch.empty(B * Spad, C, dtype=torch.bfloat16, device=backend)
    lo_ref = torch.empty_like(hi_ref)
    ops.split_bf16(dd, hi_ref, lo_ref)
    hi = torch.full((B * Spad + 3, C), 7.0, dtype=torch.bfloat16, device=backend)
    lo = torch.full_like(hi, 5.0)
    ops.shift_sum_split(src.to(backend), hi, lo, T=T, B=B, S=S, Spad=Spad)
    assert torch.equal(hi[:B * Spad].cpu(), hi_ref.cpu()) and torch.equal(lo[:B * Spad].cpu(), lo_ref.cpu())
    assert float((hi[B * Spad:].float() - 7.0).abs().max()) == 0.0 and float((lo[B * Spad:].float() - 5.0).abs().max()) == 0.0


def test_add_bf16(backend):
    g = torch.Generator().manual_seed(3)
    a = torch.randn(40, 24, generator=g).to(torch.bfloat16)
    b = torch.randn(40, 24, generator=g).to(torch.bfloat16)
    out = torch.empty(40, 24, dtype=torch.bfloat16, device=backend)
    ops.add_bf16(a.to(backend), b.to(backend), out)
    assert torch.equal(out.cpu(), a + b)   # fp32 add, one rounding == torch's bf16 add


# ------------------------------------------------------------------ teacher
@pytest.mark.parametrize("Vt,Vd", [(640, 256), (6008, 2600)])   # the second reaches the batched gather (Vd >= 8 x 256)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_teacher_reduce(backend, dtype, Vt, Vd):
    B, S, T = 2, 12, 3
    Spad = S + T
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(B * S, Vt, generator=g) * 3).to(dtype)
    z[2, 77] = z[2, 5] = z[2].float().max().to(dtype) + 1  # tie -> lowest index
    t2d, d2t = O.make_vocab_mapping(Vt, Vd, seed=1)
    loss_mask = (torch.rand(B, S, generator=g) > 0.2).long()
    tp, tpod, ids, pm = O.compute_target_p(z.view(B, S, Vt), t2d, loss_mask[..., None])
    d = lambda t: t.to(backend)
    lm_pad = torch.zeros(B, Spad, dtype=torch.int32)
    lm_pad[:, :S] = loss_mask.int()
    tp_pad = torch.full((B, Spad, Vd), 1.0 / Vd, device=backend)
    pod = torch.zeros(B, Spad, device=backend)
    tsum = torch.zeros(B, Spad, device=backend)
    ids_pad = torch.zeros(B, Spad, dtype=torch.int64, device=backend)
    pm_pad = torch.zeros(B, Spad, dtype=torch.int32, device=backend)
    ops.teacher_reduce(d(z), Vd=Vd, d2t=d(d2t), t2d_u8=d(t2d.to(torch.uint8)), loss_mask_pad=d(lm_pad), S=S, Spad=Spad,
                       target_p_pad=tp_pad, pod_scale_pad=pod, tsum_pad=tsum, ids_pad=ids_pad, pos_mask_pad=pm_pad)
    assert torch.equal(ids_pad.cpu()[:, :S], ids)                       # bit-exact
    assert torch.equal(pm_pad.cpu()[:, :S, None], pm.int())            # bit-exact
    torch.testing.assert_close(tp_pad.cpu()[:, :S], tp, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(tp_pad.cpu()[:, :S] * pod.cpu()[:, :S, None], tpod, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(tsum.cpu()[:, :S], tp.sum(-1), rtol=1e-5, atol=1e-6)
    assert float((tp_pad.cpu()[:, S:] - 1.0 / Vd).abs().max()) == 0    # padded tail untouched


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("R,H", [(20, 128), (37, 896), (19, 4096)])
def test_rmsnorm_bwd2_equals_the_sum_of_two_backwards(backend, dtype, tol, R, H):
    """two norms of the same x: dx = d_norm(dy1; w1) + d_norm(dy2; w2) + add and both weight gradients, against autograd in fp32 and
    against two sf_rmsnorm_bwd calls (fp32: same value up to summation order; bf16: one rounding fewer than the chained calls)"""
    x = _rand((R, H), dtype, 1)
    w1 = (1 + 0.1 * _rand((H,), torch.float32, 2)).to(dtype)
    w2 = (1 + 0.1 * _rand((H,), torch.float32, 3)).to(dtype)
    dy1, dy2, add = _rand((R, H), dtype, 4), _rand((R, H), dtype, 5), _rand((R, H), dtype, 6)
    xr = x.float().clone().requires_grad_(True)
    w1r, w2r = w1.float().clone().requires_grad_(True), w2.float().clone().requires_grad_(True)
    (O.rmsnorm(xr, w1r, 1e-5) * dy1.float()).sum().backward()
    (O.rmsnorm(xr, w2r, 1e-5) * dy2.float()).sum().backward()
    want_dx = xr.grad + add.float()
    d = lambda t: t.to(backend)
    y = torch.empty((R, H), dtype=dtype, device=backend)
    rstd = torch.empty(R, device=backend)
    ops.rmsnorm_fwd(d(x), d(w1), 1e-5, y, rstd)
    ws = torch.empty(2 * ops.rmsnorm_bwd_workspace(R, H), device=backend)
    dx = torch.empty((R, H), dtype=dtype, device=backend)
    dw1, dw2 = torch.full((H,), 0.5, device=backend), torch.full((H,), 9.0, device=backend)
    ops.rmsnorm_bwd2(d(dy1), d(w1), dw1, True, d(dy2), d(w2), dw2, False, d(x), rstd, dx=dx, add=d(add), workspace=ws)
    gmax = float(want_dx.abs().max())
    torch.testing.assert_close(dx.float().cpu(), want_dx, rtol=tol, atol=tol * gmax)
    torch.testing.assert_close(dw1.cpu(), 0.5 + w1r.grad, rtol=max(tol, 1e-4), atol=max(tol, 1e-4) * float(w1r.grad.abs().max()))
    torch.testing.assert_close(dw2.cpu(), w2r.grad, rtol=max(tol, 1e-4), atol=max(tol, 1e-4) * float(w2r.grad.abs().max()))
    # the chained form it replaces
    t1, t2 = torch.empty_like(dx), torch.empty_like(dx)
    a1, a2 = torch.zeros(H, device=backend), torch.zeros(H, device=backend)
    ops.rmsnorm_bwd(d(dy2), d(x), d(w2), rstd, dx=t1, add=d(add), dw_acc=a2, dw_accumulate=False, workspace=ws)
    ops.rmsnorm_bwd(d(dy1), d(x), d(w1), rstd, dx=t2, add=t1, dw_acc=a1, dw_accumulate=False, workspace=ws)
    torch.testing.assert_close(dx.float().cpu(), t2.float().cpu(), rtol=tol, atol=tol * gmax)
    torch.testing.assert_close(dw2.cpu(), a2.cpu(), rtol=1e-5, atol=1e-5 * float(a2.abs().max()))
    torch.testing.assert_close((dw1 - 0.5).cpu(), a1.cpu(), rtol=1e-4, atol=1e-4 * float(a1.abs().max()))


@pytest.mark.parametrize("R,H", [(20, 128), (70, 4096)])
def test_rmsnorm_bwd_partials_then_colsum_equals_the_fused_call(backend, R, H):
    """dw_accumulate == 2 (ABI 5): the norm backwards stop at their per-block partials and sf_colsum_accum finishes the weight gradient
    (the engine reduces the partials of a whole sweep at once: norm_colsum_batched) -- dx and dw bit-identical to the one-call form, = and += alike, for both entry points"""
    dt = torch.bfloat16
    x, dy1, dy2, add = (_rand((R, H), dt, i).to(backend) for i in (1, 2, 3, 4))
    w1 = (1 + 0.1 * _rand((H,), torch.float32, 5)).to(dt).to(backend)
    w2 = (1 + 0.1 * _rand((H,), torch.float32, 6)).to(dt).to(backend)
    rstd = torch.empty(R, device=backend)
    ops.rmsnorm_fwd(x, w1, 1e-5, torch.empty((R, H), dtype=dt, device=backend), rstd)
    n1 = ops.rmsnorm_bwd_workspace(R, H)
    nb = n1 // H
    for acc in (False, True):
        ws = torch.empty(2 * n1, device=backend)
        dxa, dwa = torch.empty((R, H), dtype=dt, device=backend), torch.full((H,), 0.25, device=backend)
        ops.rmsnorm_bwd(dy1, x, w1, rstd, dx=dxa, add=add, dw_acc=dwa, dw_accumulate=acc, workspace=ws)
        part = torch.full((2 * n1,), float("nan"), device=backend)
        dxb, dwb = torch.empty((R, H), dtype=dt, device=backend), torch.full((H,), 0.25, device=backend)
        ops.rmsnorm_bwd(dy1, x, w1, rstd, dx=dxb, add=add, workspace=part, partial_only=True)
        ops.colsum_accum(part, nb, H, dwb, acc)
        assert torch.equal(dxa.cpu(), dxb.cpu()) and torch.equal(dwa.cpu(), dwb.cpu())
        # the paired form
        dxc = torch.empty((R, H), dtype=dt, device=backend)
        c1, c2 = torch.full((H,), 0.25, device=backend), torch.full((H,), -1.0, device=backend)
        ops.rmsnorm_bwd2(dy1, w1, c1, acc, dy2, w2, c2, acc, x, rstd, dx=dxc, add=add, workspace=ws)
        part = torch.full((2 * n1,), float("nan"), device=backend)
        dxd = torch.empty((R, H), dtype=dt, device=backend)
        d1, d2 = torch.full((H,), 0.25, device=backend), torch.full((H,), -1.0, device=backend)
        ops.rmsnorm_bwd2(dy1, w1, None, False, dy2, w2, None, False, x, rstd, dx=dxd, add=add, workspace=part, partial_only=True)
        ops.colsum_accum(part[:n1], nb, H, d1, acc)
        ops.colsum_accum(part[n1:], nb, H, d2, acc)
        assert torch.equal(dxc.cpu(), dxd.cpu()) and torch.equal(c1.cpu(), d1.cpu()) and torch.equal(c2.cpu(), d2.cpu())
        # ... with the two partial blocks at destinations of the caller's choosing (the engine's per-weight arenas), the workspace untouched
        arena = torch.full((3 * n1 + 64,), float("nan"), device=backend)
        dst1, dst2 = arena[2 * n1 + 64:3 * n1 + 64], arena[16:n1 + 16]
        wsx = torch.full((2 * n1,), 7.0, device=backend)
        dxe = torch.empty((R, H), dtype=dt, device=backend)
        ops.rmsnorm_bwd2(dy1, w1, dst1, False, dy2, w2, dst2, False, x, rstd, dx=dxe, add=add, workspace=wsx, partial_only=True)
        assert torch.equal(dxe.cpu(), dxd.cpu()) and torch.equal(dst1.cpu(), part[:n1].cpu()) and torch.equal(dst2.cpu(), part[n1:].cpu())
        assert float((wsx - 7.0).abs().max()) == 0.0 and bool(torch.isnan(arena[n1 + 16:2 * n1 + 64]).all())


@pytest.mark.parametrize("nb,H", [(2049, 128), (7 * 1024, 200), (2600, 4096)])
def test_colsum_accum_over_many_partial_rows(backend, nb, H):
    """the partials of several launches reduced at once (nb > 2048: the 16-column x 64-row-lane kernel), = and +=, against an fp64 sum;
    run twice: same bits (fixed order)"""
    part = _rand((nb, H), torch.float32, 3).to(backend)
    ref = part.double().sum(0).cpu()
    acc = torch.full((H,), 0.5, device=backend)
    ops.colsum_accum(part, nb, H, acc, False)
    first = acc.clone()
    torch.testing.assert_close(acc.double().cpu(), ref, rtol=1e-5, atol=1e-3)
    ops.colsum_accum(part, nb, H, acc, True)
    torch.testing.assert_close(acc.double().cpu(), 2 * ref, rtol=1e-5, atol=2e-3)
    again = torch.empty(H, device=backend)
    ops.colsum_accum(part, nb, H, again, False)
    assert torch.equal(again.cpu(), first.cpu())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,H", [(20, 128), (19, 4096)])
def test_rmsnorm_fwd2_equals_two_norms(backend, dtype, R, H):
    x = _rand((R, H), dtype, 1)
    w1 = (1 + 0.1 * _rand((H,), torch.float32, 2)).to(dtype)
    w2 = (1 + 0.1 * _rand((H,), torch.float32, 3)).to(dtype)
    d = lambda t: t.to(backend)
    ya, yb = torch.empty((R, H), dtype=dtype, device=backend), torch.empty((R, H), dtype=dtype, device=backend)
    ra, rb = torch.empty(R, device=backend), torch.empty(R, device=backend)
    ops.rmsnorm_fwd(d(x), d(w1), 1e-5, ya, ra)
    ops.rmsnorm_fwd(d(x), d(w2), 1e-5, yb, rb)
    wide = torch.zeros((R, 2 * H + 16), dtype=dtype, device=backend)          # strided outputs
    y1, y2 = wide[:, :H], wide[:, H + 16:]
    r1, r2 = torch.empty(R, device=backend), torch.empty(R, device=backend)
    ops.rmsnorm_fwd2(d(x), d(w1), y1, r1, d(w2), y2, r2, 1e-5)
    assert torch.equal(y1.cpu(), ya.cpu()) and torch.equal(y2.cpu(), yb.cpu())
    assert torch.equal(r1.cpu(), ra.cpu()) and torch.equal(r2.cpu(), rb.cpu())
    assert float(wide[:, H:H + 16].abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Vt,Vd", [(640, 64), (1000, 96), (2056, 512)])
def test_teacher_reduce_perm(backend, dtype, Vt, Vd):
    """logits with permuted columns (draft sub-vocabulary first): the same targets as the reference's _compute_target_p on the
    natural layout; ties in the argmax -- inside the draft block, inside the rest, across the two -- go to the lowest ORIGINAL index"""
    B, S, T = 2, 12, 3
    Spad = S + T
    g = torch.Generator().manual_seed(13)
    z = (torch.randn(B * S, Vt, generator=g) * 3).to(dtype)
    t2d, d2t = O.make_vocab_mapping(Vt, Vd, seed=2)
    cols = torch.arange(Vd) + d2t
    rest = torch.nonzero(~t2d.bool()).flatten()
    top = lambda r: z[r].float().max().to(dtype) + 1
    z[2, rest[5]] = z[2, cols[3]] = top(2)                   # across the two blocks
    z[3, rest[1]] = z[3, rest[40]] = top(3)                  # inside the rest
    z[4, cols[7]] = z[4, cols[2]] = top(4)                   # inside the draft block
    z[5, cols[Vd - 1]] = z[5, rest[0]] = top(5)
    z[6, cols] -= 200                                        # every draft logit far below the row maximum: the exact-sum branch
    loss_mask = (torch.rand(B, S, generator=g) > 0.2).long()
    tp, tpod, ids, pm = O.compute_target_p(z.view(B, S, Vt), t2d, loss_mask[..., None])
    perm = torch.cat([cols, rest])
    zp = z[:, perm].contiguous()
    d = lambda t: t.to(backend)
    lm_pad = torch.zeros(B, Spad, dtype=torch.int32)
    lm_pad[:, :S] = loss_mask.int()
    tp_pad = torch.full((B, Spad, Vd), 1.0 / Vd, device=backend)
    pod = torch.zeros(B, Spad, device=backend)
    tsum = torch.zeros(B, Spad, device=backend)
    ids_pad = torch.zeros(B, Spad, dtype=torch.int64, device=backend)
    pm_pad = torch.zeros(B, Spad, dtype=torch.int32, device=backend)
    ops.teacher_reduce_perm(d(zp), Vt=Vt, Vd=Vd, perm=d(perm.to(torch.int32)), t2d_u8=d(t2d.to(torch.uint8)), loss_mask_pad=d(lm_pad),
                            S=S, Spad=Spad, target_p_pad=tp_pad, pod_scale_pad=pod, tsum_pad=tsum, ids_pad=ids_pad, pos_mask_pad=pm_pad)
    assert torch.equal(ids_pad.cpu()[:, :S], ids)                       # bit-exact
    assert torch.equal(pm_pad.cpu()[:, :S, None], pm.int())            # bit-exact
    torch.testing.assert_close(tp_pad.cpu()[:, :S], tp, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(tp_pad.cpu()[:, :S] * pod.cpu()[:, :S, None], tpod, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(tsum.cpu()[:, :S], tp.sum(-1), rtol=1e-5, atol=1e-6)
    assert float((tp_pad.cpu()[:, S:] - 1.0 / Vd).abs().max()) == 0    # padded tail untouched


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ce_fused_from_teacher_logits_equals_materialised_targets(backend, dtype):
    """sf_ce_fused_zt (soft target re-formed from the teacher's stored draft logits + per-row max / 1/sum-exp) == sf_ce_fused on the
    target_p array the same teacher kernel writes: losses, in-place gradients, accuracy and acceptance bit for bit, at two TTT offsets"""
    B, S, T, Vt, Vd = 2, 12, 3, 640, 384
    Spad = S + T
    g = torch.Generator().manual_seed(21)
    zfull = (torch.randn(B * S, Vt, generator=g) * 3).to(torch.bfloat16)
    t2d, d2t = O.make_vocab_mapping(Vt, Vd, seed=4)
    perm = torch.cat([torch.arange(Vd) + d2t, torch.nonzero(~t2d.bool()).flatten()])
    zp = zfull[:, perm].contiguous()
    d = lambda t: t.to(backend)
    lm_pad = torch.zeros(B, Spad, dtype=torch.int32)
    lm_pad[:, :S] = (torch.rand(B, S, generator=g) > 0.2).int()
    tp = torch.full((B, Spad, Vd), 1.0 / Vd, device=backend)
    o = dict(pod_scale_pad=torch.zeros(B, Spad, device=backend), tsum_pad=torch.full((B, Spad), 1.0, device=backend),
             ids_pad=torch.zeros(B, Spad, dtype=torch.int64, device=backend), pos_mask_pad=torch.zeros(B, Spad, dtype=torch.int32, device=backend))
    zmd, zinv = torch.zeros(B, Spad, device=backend), torch.zeros(B, Spad, device=backend)
    kw = dict(Vt=Vt, Vd=Vd, perm=d(perm.to(torch.int32)), t2d_u8=d(t2d.to(torch.uint8)), loss_mask_pad=d(lm_pad), S=S, Spad=Spad)
    ops.teacher_reduce_perm(d(zp), target_p_pad=tp, zmd_pad=zmd, zinv_pad=zinv, **kw, **o)
    o2 = {k: torch.zeros_like(v) for k, v in o.items()}
    zmd2, zinv2 = torch.zeros_like(zmd), torch.zeros_like(zinv)
    ops.teacher_reduce_perm(d(zp), target_p_pad=None, zmd_pad=zmd2, zinv_pad=zinv2, **kw, **o2)      # no materialised probabilities
    for k in o:
        if k == "tsum_pad":     # not materialised: the probabilities' sum is sd * (1 / sd), not the sum of Vd rounded terms
            torch.testing.assert_close(o[k].cpu()[:, :S], o2[k].cpu()[:, :S], rtol=2e-6, atol=0)
            continue
        assert torch.equal(o[k].cpu()[:, :S], o2[k].cpu()[:, :S]), k
    assert torch.equal(zmd.cpu(), zmd2.cpu()) and torch.equal(zinv.cpu(), zinv2.cpu())
    zd = torch.zeros(B * S, Vd + 8, dtype=torch.bfloat16)                                             # the stored draft logits (wider rows)
    zd[:, :Vd] = zp[:, :Vd]
    for off in (0, 2):
        logits = (torch.randn(B * S, Vd, generator=g) * 2).to(dtype)
        outs = []
        for form in ("tp", "zt"):
            x = d(logits.clone())
            rows = [torch.zeros(B * S, device=backend) for _ in range(3)]
            pred = torch.zeros(B * S, dtype=torch.int32, device=backend)
            common = dict(S=S, Spad=Spad, off=off, pos_mask_pad=o["pos_mask_pad"], loss_mask_pad=d(lm_pad), tgt_ids_pad=o["ids_pad"],
                          pod_scale_pad=o["pod_scale_pad"], tsum_pad=o["tsum_pad"], d2t=d(d2t), grad_scale=0.37, write_grad=True,
                          row_loss=rows[0], row_correct=rows[1], row_accept=rows[2], row_pred=pred)
            if form == "tp":
                ops.ce_fused(x, tp, **common)
            else:
                ops.ce_fused_zt(x, d(zd), zmd, zinv, **common)
            outs.append([x.cpu()] + [r.cpu() for r in rows] + [pred.cpu()])
        for a, b_ in zip(*outs):
            assert torch.equal(a, b_)
        assert float(outs[0][1].abs().sum()) > 0          # (some rows carry a position mask at this offset)
        # row_map (ABI 5, loss-row compaction): the kernels over a COMPACT copy of the rows with loss_mask[b, s + off] != 0 give, row
        # for row, the bits of the dense call -- and the rows left out are exactly the ones the dense call zeroes
        keep = torch.nonzero(lm_pad[:, off:off + S].reshape(-1)).view(-1)
        for form in ("tp", "zt"):
            xc = d(logits[keep].clone())
            rows = [torch.zeros(keep.numel(), device=backend) for _ in range(3)]
            pred = torch.zeros(keep.numel(), dtype=torch.int32, device=backend)
            common.update(row_loss=rows[0], row_correct=rows[1], row_accept=rows[2], row_pred=pred, row_map=d(keep))
            if form == "tp":
                ops.ce_fused(xc, tp, **common)
            else:
                ops.ce_fused_zt(xc, d(zd), zmd, zinv, **common)
            dense = outs[0]
            assert torch.equal(xc.cpu(), dense[0][keep])
            for i in range(3):
                assert torch.equal(rows[i].cpu(), dense[1 + i][keep])
            assert torch.equal(pred.cpu(), dense[4][keep])
            left = torch.ones(B * S, dtype=torch.bool)
            left[keep] = False
            assert float(dense[0][left].float().abs().max()) == 0 and all(float(dense[1 + i][left].abs().max()) == 0 for i in range(3))


@pytest.mark.parametrize("M,Vt,Vd,K", [(512, 1000, 256, 512),     # reduced range 256 .. 999: 6 blocks, the last one partly past Vt
                                       (300, 1408, 200, 576),     # ragged rows, Vd not a multiple of 256, Vt a multiple of 128 only
                                       (256, 512, 512, 512)])     # Vz == Vt: nothing to reduce, everything stored
def test_gemm_nt_teacher_matches_the_stored_logits(backend, M, Vt, Vd, K):
    """the head GEMM with the reduction epilogue + teacher_reduce_perm over (stored draft columns, partials) == the same GEMM
    storing every logit + teacher_reduce_perm over the full rows: ids / position mask bit-exact, probabilities to fp32 rounding"""
    B, S, T = 1, M, 2
    Spad = S + T
    g = torch.Generator().manual_seed(17)
    x = _rand((M, K), torch.bfloat16, 1)
    w = (_rand((Vt, K), torch.bfloat16, 2) * 0.25).to(torch.bfloat16)
    t2d, d2t = O.make_vocab_mapping(Vt, Vd, seed=3)
    cols = torch.arange(Vd) + d2t
    perm = torch.cat([cols, torch.nonzero(~t2d.bool()).flatten()])
    wp = w[perm].contiguous()
    if Vd + 300 < Vt:     # two identical, dominant columns in different reduced blocks: a tie for the argmax on about half the rows
        wp[Vd + 40] = (wp[Vd + 40].float() * 3).to(torch.bfloat16)
        wp[Vd + 300] = wp[Vd + 40]
    d = lambda t: t.to(backend)
    lm_pad = torch.ones(B, Spad, dtype=torch.int32)
    res = []
    for fused in (False, True):
        z = torch.full((M, Vt), 9.0, dtype=torch.bfloat16, device=backend)
        part = torch.full((M + 3, (Vt - Vd + 127) // 128 + 2, 4), -7.0, device=backend) if fused else None
        vz, nparts = ops.gemm_nt_teacher(d(x), d(wp), z, part, Vd=Vd)
        if nparts:        # (a GPU takes the reduction epilogue for chip-filling shapes only; the interpreter for every long-K shape)
            assert fused and vz == (Vd + 255) // 256 * 256 and nparts == (Vt - vz + 127) // 128
            assert float((z[:, vz:].float() - 9.0).abs().max()) == 0.0          # the reduced columns are never written
        else:
            assert vz == Vt
        if backend == "cpu":   # (= the interpreter)
            assert bool(nparts) == (fused and Vt > (Vd + 255) // 256 * 256)
        o = dict(target_p_pad=torch.zeros(B, Spad, Vd, device=backend), pod_scale_pad=torch.zeros(B, Spad, device=backend),
                 tsum_pad=torch.zeros(B, Spad, device=backend), ids_pad=torch.zeros(B, Spad, dtype=torch.int64, device=backend),
                 pos_mask_pad=torch.zeros(B, Spad, dtype=torch.int32, device=backend))
        ops.teacher_reduce_perm(z[:, :vz], Vt=Vt, Vd=Vd, perm=d(perm.to(torch.int32)), t2d_u8=d(t2d.to(torch.uint8)), loss_mask_pad=d(lm_pad),
                                S=S, Spad=Spad, part=part, nparts=nparts, **o)
        res.append({k: v.cpu() for k, v in o.items()})
    a, f = res
    assert torch.equal(a["ids_pad"], f["ids_pad"]) and torch.equal(a["pos_mask_pad"], f["pos_mask_pad"])
    torch.testing.assert_close(f["target_p_pad"], a["target_p_pad"], rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(f["pod_scale_pad"], a["pod_scale_pad"], rtol=1e-4, atol=1e-8)
    # ... and against the reference restatement on the natural layout (logits rounded to bf16 as TargetHead does)
    zn = (x.float() @ w.float().t())
    zn_p = (x.float() @ wp.float().t()).to(torch.bfloat16)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(Vt)
    tp, tpod, ids, pm = O.compute_target_p(zn_p[:, inv].view(B, S, Vt), t2d, lm_pad[:, :S, None].long())
    same = float((f["ids_pad"][:, :S] == ids).float().mean())
    assert same >= 0.99, same                                                    # (accumulation-order near-ties of the bf16 rounding)
    del zn


# ------------------------------------------------------------------ RMSNorm
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("R,H", [(20, 128), (33, 896), (19, 4096), (5, 7168)])   # 1 / 1 / 2 / 4 vectors per thread
def test_rmsnorm_fwd_bwd(backend, dtype, tol, R, H):
    x = _rand((R, H), dtype, 1)
    w = (1 + 0.1 * _rand((H,), torch.float32, 2)).to(dtype)
    dy = _rand((R, H), dtype, 3)
    add = _rand((R, H), dtype, 4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = O.rmsnorm(xr, wr, 1e-5)
    y_ref.backward(dy)
    d = lambda t: t.to(backend)
    ycat = torch.zeros((R, 2 * H), dtype=dtype, device=backend)
    rstd = torch.empty(R, device=backend)
    ops.rmsnorm_fwd(d(x), d(w), 1e-5, ycat[:, H:], rstd)
    torch.testing.assert_close(ycat[:, H:].float().cpu(), y_ref.detach().float(), rtol=tol, atol=tol)
    assert float(ycat[:, :H].abs().max()) == 0
    dx = torch.empty((R, H), dtype=dtype, device=backend)
    dw = torch.full((H,), 0.5, device=backend)
    ws = torch.empty(ops.rmsnorm_bwd_workspace(R, H), device=backend)
    ops.rmsnorm_bwd(d(dy), d(x), d(w), rstd, dx=dx, add=d(add), dw_acc=dw, dw_accumulate=True, workspace=ws)
    gx = xr.grad.float() + add.float()
    torch.testing.assert_close(dx.float().cpu(), gx, rtol=tol, atol=tol * float(gx.abs().max()))
    gw = wr.grad.float()
    torch.testing.assert_close(dw.cpu() - 0.5, gw, rtol=max(tol, 1e-4), atol=max(tol, 1e-4) * float(gw.abs().max()))


def test_rmsnorm_gather(backend):
    B, S, T, H, V = 2, 8, 3, 128, 50
    table = _rand((V, H), torch.bfloat16, 1)
    w = (1 + 0.1 * _rand((H,), torch.float32, 2)).to(torch.bfloat16)
    ids_pad = torch.zeros(B, S + T, dtype=torch.int64)
    ids_pad[:, :S] = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(3))
    for off in (0, 2):
        y = torch.empty((B * S, H), dtype=torch.bfloat16, device=backend)
        rstd = torch.empty(B * S, device=backend)
        ops.rmsnorm_fwd(table.to(backend), w.to(backend), 1e-5, y, rstd, ids_pad=ids_pad.to(backend), S=S, Spad=S + T, off=off)
        emb = table[ids_pad[:, off:off + S].reshape(-1)]
        torch.testing.assert_close(y.float().cpu(), O.rmsnorm(emb, w, 1e-5).float(), rtol=2e-2, atol=2e-2)


# ------------------------------------------------------------------ RoPE / SwiGLU / transpose / cast
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("hd", [64, 128])
def test_rope(backend, dtype, tol, hd):
    B, S, nh, nkv = 2, 8, 3, 1
    cfg = O.DraftConfig(hidden_size=nh * hd, intermediate_size=8, num_attention_heads=nh, num_key_value_heads=nkv,
                        vocab_size=8, draft_vocab_size=8, head_dim=hd, max_position_embeddings=64)
    cos, sin = O.rope_tables(cfg, 84, dtype)
    qkv = _rand((B * S, (nh + 2 * nkv) * hd), dtype, 1)
    pos = torch.arange(S).repeat(B)
    q = qkv[:, :nh * hd].reshape(B, S, nh, hd).transpose(1, 2).clone().requires_grad_(True)
    k = qkv[:, nh * hd:(nh + nkv) * hd].reshape(B, S, nkv, hd).transpose(1, 2).clone().requires_grad_(True)
    qe, ke = O.apply_rope(q, k, cos, sin, pos.view(B, S) + 3)
    x = qkv.clone().to(backend)
    ops.rope_(x, nh + nkv, hd, cos.to(backend), sin.to(backend), pos.to(backend), 3)
    got_q = x[:, :nh * hd].reshape(B, S, nh, hd).transpose(1, 2).float().cpu()
    got_k = x[:, nh * hd:(nh + nkv) * hd].reshape(B, S, nkv, hd).transpose(1, 2).float().cpu()
    torch.testing.assert_close(got_q, qe.detach().float(), rtol=tol, atol=tol)
    torch.testing.assert_close(got_k, ke.detach().float(), rtol=tol, atol=tol)
    assert torch.equal(x[:, (nh + nkv) * hd:].cpu(), qkv[:, (nh + nkv) * hd:])  # v untouched
    dq = _rand(q.shape, dtype, 2)
    qe.backward(dq)
    g = dq.transpose(1, 2).reshape(B * S, nh * hd).clone().to(backend)
    ops.rope_(g, nh, hd, cos.to(backend), sin.to(backend), pos.to(backend), 3, backward=True)
    torch.testing.assert_close(g.float().cpu(), q.grad.transpose(1, 2).reshape(B * S, nh * hd).float(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_swiglu(backend, dtype, tol):
    R, I = 37, 192
    gu = _rand((R, 2 * I), dtype, 1)
    dact = _rand((R, I), dtype, 2)
    gr = gu.clone().requires_grad_(True)
    ref = torch.nn.functional.silu(gr[:, :I]) * gr[:, I:]
    ref.backward(dact)
    act = torch.empty((R, I), dtype=dtype, device=backend)
    ops.swiglu_fwd(gu.to(backend), act)
    torch.testing.assert_close(act.float().cpu(), ref.detach().float(), rtol=tol, atol=tol)
    dgu = torch.empty((R, 2 * I), dtype=dtype, device=backend)
    ops.swiglu_bwd(dact.to(backend), gu.to(backend), dgu)
    torch.testing.assert_close(dgu.float().cpu(), gr.grad.float(), rtol=tol, atol=tol * float(gr.grad.abs().max()))


@pytest.mark.parametrize("M,I,K", [(512, 768, 512),    # whole 256 x 256 tiles, long K: the fused epilogue of the 4-wave kernel (interpreter)
                                   (600, 256, 512),    # ragged M: 3 fused row tiles, the last one shifted up to end at row 600 (mshift)
                                   (300, 264, 128)])   # everything else: gemm_nt + swiglu_bwd through the scratch
def test_gemm_nt_swiglu_bwd_equals_the_two_steps(backend, M, I, K):
    """d(act) = dY . W_down with d(SwiGLU) in the GEMM epilogue == the same GEMM followed by swiglu_bwd (same roundings)"""
    dy, w = _rand((M, K), torch.bfloat16, 1), _rand((I, K), torch.bfloat16, 2)
    gu = _rand((M, 2 * I), torch.bfloat16, 3)
    d = lambda t: t.to(backend)
    dact = torch.empty((M, I), dtype=torch.bfloat16, device=backend)
    ref = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(d(dy), d(w), dact)
    ops.swiglu_bwd(dact, d(gu), ref)
    out = torch.full((M, 2 * I), 7.0, dtype=torch.bfloat16, device=backend)
    scratch = torch.empty((M, I), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt_swiglu_bwd(d(dy), d(w), d(gu), out, scratch)
    same = float((out == ref).float().mean())
    assert same >= 0.999, same                       # fp contraction may differ between the two call sites: <= 1 bf16 ulp, rarely
    torch.testing.assert_close(out.float().cpu(), ref.float().cpu(), rtol=2 ** -6, atol=1e-30)


@pytest.mark.parametrize("M,I,K", [(512, 384, 512),    # whole tiles (3 n-tiles of 128 act columns), long K: the fused epilogue of the 4-wave kernel
                                   (768, 128, 576),    # one n-tile; K-tile count odd
                                   (600, 256, 512),    # ragged M: 3 fused row tiles, the last one shifted up to end at row 600 (mshift)
                                   (300, 264, 128)])   # everything else: gemm_nt + swiglu_fwd
def test_gemm_nt_swiglu_fwd_equals_the_two_steps(backend, M, I, K):
    """gate|up = x . Wgu^T with SwiGLU in the GEMM epilogue == the same GEMM followed by swiglu_fwd: gate|up bit-identical (the
    interleaved B-row order of the fused tile changes no accumulation order), act from the same rounded gate / up"""
    x, w = _rand((M, K), torch.bfloat16, 1), _rand((2 * I, K), torch.bfloat16, 2)
    d = lambda t: t.to(backend)
    gu_ref = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=backend)
    act_ref = torch.empty((M, I), dtype=torch.bfloat16, device=backend)
    ops.gemm_nt(d(x), d(w), gu_ref)
    ops.swiglu_fwd(gu_ref, act_ref)
    gu = torch.full((M, 2 * I), 7.0, dtype=torch.bfloat16, device=backend)
    act = torch.full((M, I), 5.0, dtype=torch.bfloat16, device=backend)
    ops.gemm_nt_swiglu_fwd(d(x), d(w), gu, act)
    assert torch.equal(gu.cpu(), gu_ref.cpu())
    same = float((act == act_ref).float().mean())
    assert same >= 0.999, same                       # fp contraction may differ between the two call sites: <= 1 bf16 ulp, rarely
    torch.testing.assert_close(act.float().cpu(), act_ref.float().cpu(), rtol=2 ** -6, atol=1e-30)
    ref32 = x.float() @ w.float().t()
    want = torch.nn.functional.silu(ref32[:, :I]) * ref32[:, I:]
    torch.testing.assert_close(act.float().cpu(), want, rtol=3e-2, atol=3e-2 * float(want.abs().max()) / 4)


def test_transpose_and_cast(backend):
    x = _rand((72, 200), torch.bfloat16, 1)
    out = torch.empty((200, 72), dtype=torch.bfloat16, device=backend)
    ops.transpose2d(x.to(backend), out)
    assert torch.equal(out.cpu(), x.t().contiguous())
    B, S, nh, hd = 2, 24, 3, 64
    wide = _rand((B * S, nh * hd + 64), torch.bfloat16, 2)
    outh = torch.empty((B, nh, hd, S), dtype=torch.bfloat16, device=backend)
    ops.transpose_heads(wide.to(backend)[:, 64:], outh, B, S, nh, hd)
    ref = wide[:, 64:].reshape(B, S, nh, hd).permute(0, 2, 3, 1).contiguous()
    assert torch.equal(outh.cpu(), ref)
    f = _rand((16, 128), torch.float32, 3)
    o = torch.zeros((16, 256), dtype=torch.bfloat16, device=backend)
    ops.cast_from_f32(f.to(backend), o[:, 128:], 0.5)
    assert torch.equal(o[:, 128:].cpu(), (f * 0.5).to(torch.bfloat16))


# ------------------------------------------------------------------ optimizer
def test_grad_norm_and_adamw_match_reference_run(backend, golden_dir):
    import os

    blob = torch.load(os.path.join(golden_dir, "optimizer_bf16.pt"), weights_only=False)
    sizes = [p.numel() for p in blob["p0"]]
    flat = torch.cat([p.reshape(-1) for p in blob["p0"]]).to(backend)           # bf16 params
    master = flat.float().clone()
    m = torch.zeros_like(master)
    v = torch.zeros_like(master)
    norm = torch.empty(1, device=backend)
    ws = torch.empty(1024, device=backend)
    for step in range(4):
        g = torch.cat([x.reshape(-1) for x in blob["grads"][step]]).to(backend)
        ops.grad_norm(g, norm, ws)
        torch.testing.assert_close(norm.cpu()[0], blob["norms"][step], rtol=1e-5, atol=1e-6)
        lr = O.cosine_warmup_lr(step, blob["lr"], blob["total_steps"], blob["warmup_steps"])
        assert abs(lr - blob["lrs"][step]) < 1e-12
        ops.adamw_step(g, master, m, v, flat, norm, max_norm=blob["max_grad_norm"], lr=lr, beta1=0.9, beta2=0.999,
                       eps=1e-8, wd=0.0, step=step + 1)
        want = torch.cat([p.reshape(-1) for p in blob["params_after"][step]])
        # bf16 params after the reference's BF16Optimizer.step: equal up to 1 bf16 ulp of rounding ties
        torch.testing.assert_close(flat.float().cpu(), want.float(), rtol=8e-3, atol=1e-6)
    want_master = torch.cat([t.reshape(-1) for t in blob["masters"]])
    torch.testing.assert_close(master.cpu(), want_master, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(m.cpu(), torch.cat([t.reshape(-1) for t in blob["exp_avg"]]), rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(v.cpu(), torch.cat([t.reshape(-1) for t in blob["exp_avg_sq"]]), rtol=1e-5, atol=1e-10)
