"""Tensor-level wrappers over the C-ABI (one function per entry point of include/specforge_amd.h).

torch is used for what it is here for: device memory and streams.  Every function takes
torch tensors living on the GPU, hands raw pointers / strides / the current HIP stream to
libsfhip.so and returns.  No arithmetic happens in Python; there is no fallback path.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib

BF16, F32 = 0, 1


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"specforge_amd: unsupported dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.is_cuda == _lib.is_emulated():
        # product: everything must be on the GPU.  (emulated test build: everything on the host)
        raise RuntimeError(
            "specforge_amd ops need CUDA/HIP tensors (libsfhip.so runs on the GPU; there is no CPU path)"
            if not _lib.is_emulated() else "emulated test build needs CPU tensors")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    if _lib.is_emulated():
        return None
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rowmajor(t: torch.Tensor) -> int:
    """leading dimension (elements) of a 2-D view whose inner stride is 1"""
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0)


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, alpha: float = 1.0, beta: float = 0.0,
            residual: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None):
    """out[M,N] = alpha * a[M,K] @ b[N,K]^T (+ beta*out) (+ residual); a, b bf16; out bf16|fp32.
    ``workspace`` (fp32, alpha 1 / beta 0 only): lets an under-filled grid run split-K (sf_gemm_nt_ws); other shapes ignore it."""
    L = _lib.lib()
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and out.shape == (M, N) and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    if workspace is not None and alpha == 1.0 and beta == 0.0:
        assert workspace.dtype == torch.float32 and workspace.is_contiguous()
        _lib.check(L.sf_gemm_nt_ws(_p(a), _rowmajor(a), _p(b), _rowmajor(b), _p(out), _dt(out), _rowmajor(out), M, N, K, _p(residual),
                                   _rowmajor(residual) if residual is not None else 0, _p(workspace), workspace.numel(), _stream()),
                   "sf_gemm_nt_ws")
        return out
    _lib.check(L.sf_gemm_nt(_p(a), _rowmajor(a), _p(b), _rowmajor(b), _p(out), _dt(out), _rowmajor(out), M, N, K,
                            alpha, beta, _p(residual), _rowmajor(residual) if residual is not None else 0, _stream()),
               "sf_gemm_nt")
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, alpha: float = 1.0, beta: float = 0.0,
            workspace: Optional[torch.Tensor] = None, ksplit: int = 0):
    """out[M,N] = alpha * a[K,M]^T @ b[K,N] (+ beta*out); a, b bf16 (row-major, K outermost); out bf16|fp32.
    workspace: optional fp32 scratch (>= 2*M*N) that allows a 2-way split of K for awkward tile counts
    (ksplit: 0 = the launcher decides, 1 = never, 2 = always)."""
    L = _lib.lib()
    K, M = a.shape
    K2, N = b.shape
    assert K == K2 and out.shape == (M, N) and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert workspace is None or workspace.dtype == torch.float32
    _lib.check(L.sf_gemm_tn(_p(a), _rowmajor(a), _p(b), _rowmajor(b), _p(out), _dt(out), _rowmajor(out), M, N, K, alpha,
                            beta, _p(workspace), workspace.numel() if workspace is not None else 0, ksplit, _stream()), "sf_gemm_tn")
    return out


def gemm_nt_rowadd(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, addend: torch.Tensor, *, S: int, Spad: int, off: int,
                   alpha: float = 1.0, workspace: Optional[torch.Tensor] = None):
    """out[r] = round(alpha * a[r] @ b^T + addend[(r // S) * Spad + r % S + off]); addend fp32 [B*Spad, N]."""
    L = _lib.lib()
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and out.shape == (M, N) and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert addend.dtype == torch.float32 and addend.shape[1] == N and addend.shape[0] >= (M // S) * Spad
    _lib.check(L.sf_gemm_nt_rowadd(_p(a), _rowmajor(a), _p(b), _rowmajor(b), _p(out), _dt(out), _rowmajor(out), M, N, K,
                                   alpha, _p(addend), _rowmajor(addend), S, Spad, off, _p(workspace),
                                   0 if workspace is None else workspace.numel(), _stream()), "sf_gemm_nt_rowadd")
    return out


def rows_expand(src: torch.Tensor, inv: torch.Tensor, dst: torch.Tensor):
    """dst[r] = src[inv[r]] where inv[r] >= 0, zeros elsewhere (the inverse of a row compaction); inv int32 [dst rows]"""
    L = _lib.lib()
    rows, C = dst.shape
    assert inv.dtype == torch.int32 and inv.numel() >= rows and inv.is_contiguous() and src.shape[1] == C and src.dtype == dst.dtype
    _lib.check(L.sf_rows_expand(_p(src), _dt(src), _rowmajor(src), _p(inv), _p(dst), _rowmajor(dst), rows, C, _stream()), "sf_rows_expand")
    return dst


def ce_fused(logits: torch.Tensor, target_pad: torch.Tensor, *, S: int, Spad: int, off: int, pos_mask_pad, loss_mask_pad,
             tgt_ids_pad=None, pod_scale_pad=None, tsum_pad=None, d2t=None, grad_scale: float = 1.0, write_grad: bool = True,
             row_loss, row_correct, row_accept, row_pred=None, row_map=None):
    """``row_map`` [rows] int64 (optional): logits row r is token row row_map[r] of the [B, S] grid (loss-row compaction)"""
    L = _lib.lib()
    rows, V = logits.shape
    assert target_pad.dtype == torch.float32 and target_pad.is_contiguous() and target_pad.shape[-1] == V
    assert row_map is None or (row_map.dtype == torch.int64 and row_map.numel() >= rows and row_map.is_contiguous())
    _lib.check(L.sf_ce_fused(_p(logits), _dt(logits), _rowmajor(logits), rows, V, _p(target_pad), S, Spad, off,
                             _p(pos_mask_pad), _p(loss_mask_pad), _p(tgt_ids_pad), _p(pod_scale_pad), _p(tsum_pad),
                             _p(d2t), grad_scale, 1 if write_grad else 0, _p(row_loss), _p(row_correct), _p(row_accept),
                             _p(row_pred), _p(row_map), _stream()), "sf_ce_fused")


def ce_fused_zt(logits: torch.Tensor, zt: torch.Tensor, zmd_pad: torch.Tensor, zinv_pad: torch.Tensor, *, S: int, Spad: int, off: int,
                pos_mask_pad, loss_mask_pad, tgt_ids_pad=None, pod_scale_pad=None, tsum_pad, d2t=None, grad_scale: float = 1.0,
                write_grad: bool = True, row_loss, row_correct, row_accept, row_pred=None, row_map=None):
    """ce_fused with the soft target re-formed from the teacher's stored draft logits ``zt`` [B*S, >= V] (natural rows) and the
    per-row (max, 1 / sum-exp) of teacher_reduce_perm"""
    L = _lib.lib()
    rows, V = logits.shape
    assert zt.dtype == torch.bfloat16 and (row_map is not None or zt.shape[0] >= rows) and zt.shape[1] >= V
    assert row_map is None or (row_map.dtype == torch.int64 and row_map.numel() >= rows and row_map.is_contiguous())
    assert zmd_pad.dtype == zinv_pad.dtype == torch.float32 and zmd_pad.is_contiguous() and zinv_pad.is_contiguous()
    _lib.check(L.sf_ce_fused_zt(_p(logits), _dt(logits), _rowmajor(logits), rows, V, _p(zt), _rowmajor(zt), _p(zmd_pad), _p(zinv_pad), S,
                                Spad, off, _p(pos_mask_pad), _p(loss_mask_pad), _p(tgt_ids_pad), _p(pod_scale_pad), _p(tsum_pad), _p(d2t),
                                grad_scale, 1 if write_grad else 0, _p(row_loss), _p(row_correct), _p(row_accept), _p(row_pred),
                                _p(row_map), _stream()), "sf_ce_fused_zt")


def ce_lk_grad(logits: torch.Tensor, target_pad: torch.Tensor, *, S: int, Spad: int, off: int, pos_mask_pad, pod_scale_pad,
               tsum_pad=None, lk_loss_type: str, kl_scale: float, kl_decay: float, step_scale: float, kl_row_scale: float,
               accept_sum: torch.Tensor, mask_sum: torch.Tensor):
    """in-place d(step_scale * lk_loss)/d(logits); see include/specforge_amd.h"""
    L = _lib.lib()
    rows, V = logits.shape
    mode = {"alpha": 1, "lambda": 2}[lk_loss_type]
    assert target_pad.dtype == torch.float32 and target_pad.is_contiguous() and target_pad.shape[-1] == V
    assert accept_sum.dtype == torch.float32 and mask_sum.dtype == torch.float32
    _lib.check(L.sf_ce_lk_grad(_p(logits), _dt(logits), _rowmajor(logits), rows, V, _p(target_pad), S, Spad, off,
                               _p(pos_mask_pad), _p(pod_scale_pad), _p(tsum_pad), mode, kl_scale, kl_decay, step_scale,
                               kl_row_scale, _p(accept_sum), _p(mask_sum), _stream()), "sf_ce_lk_grad")


def reduce_sum(inp: torch.Tensor, n: int, nseg: int, out: torch.Tensor, scale: float = 1.0):
    L = _lib.lib()
    assert inp.dtype == torch.float32 and out.dtype == torch.float32 and inp.numel() >= n * nseg and out.numel() >= nseg
    _lib.check(L.sf_reduce_sum(_p(inp), n, nseg, _p(out), scale, _stream()), "sf_reduce_sum")
    return out


def eagle3_metrics(met: torch.Tensor, loss_mask_pad: torch.Tensor, pos_mask_pad: torch.Tensor, out: torch.Tensor, *, B: int, S: int,
                   Spad: int, T: int):
    """out[k] = {ploss, acc_correct, acc_denom, acc, acceptance_rate, pos_denom, B*S, ploss} of TTT step k"""
    L = _lib.lib()
    assert met.dtype == out.dtype == torch.float32 and met.is_contiguous() and out.is_contiguous()
    assert loss_mask_pad.dtype == pos_mask_pad.dtype == torch.int32 and loss_mask_pad.is_contiguous() and pos_mask_pad.is_contiguous()
    assert met.numel() >= 3 * T and out.numel() >= 8 * T and loss_mask_pad.shape == pos_mask_pad.shape == (B, Spad)
    _lib.check(L.sf_eagle3_metrics(_p(met), _p(loss_mask_pad), _p(pos_mask_pad), B, S, Spad, T, _p(out), _stream()), "sf_eagle3_metrics")


def teacher_reduce(z: torch.Tensor, *, Vd: int, d2t, t2d_u8, loss_mask_pad, S: int, Spad: int, target_p_pad,
                   pod_scale_pad, tsum_pad, ids_pad, pos_mask_pad, row0: int = 0):
    """rows of z are tokens row0 .. row0+rows (row index b*S+s)."""
    L = _lib.lib()
    rows, Vt = z.shape
    assert row0 == 0, "chunked teacher rows are addressed by offsetting the padded outputs on the caller side"
    _lib.check(L.sf_teacher_reduce(_p(z), _dt(z), _rowmajor(z), rows, Vt, Vd, _p(d2t), _p(t2d_u8), _p(loss_mask_pad), S,
                                   Spad, _p(target_p_pad), _p(pod_scale_pad), _p(tsum_pad), _p(ids_pad),
                                   _p(pos_mask_pad), _stream()), "sf_teacher_reduce")


def gemm_nt_teacher(a: torch.Tensor, w_perm: torch.Tensor, z: torch.Tensor, part: Optional[torch.Tensor], *, Vd: int):
    """z = a @ w_perm.T for the row-permuted teacher head.  Returns (Vz, nparts): the first Vz columns of z were stored, the
    rest left as ``nparts`` per-128-column-block partials in ``part`` [>= rows, part_stride >= nparts, 4] (nparts = 0: all stored)."""
    import ctypes
    L = _lib.lib()
    M, K = a.shape
    Vt = w_perm.shape[0]
    assert z.shape[0] >= M and a.dtype == w_perm.dtype == z.dtype == torch.bfloat16      # (z narrower than Vt: the C side checks)
    stride = 0
    if part is not None:
        assert part.dtype == torch.float32 and part.dim() == 3 and part.shape[2] == 4 and part.is_contiguous()
        assert part.shape[0] >= M and part.shape[1] >= (Vt - Vd + 127) // 128
        stride = part.shape[1]
    vz, npart = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.sf_gemm_nt_teacher(_p(a), _rowmajor(a), _p(w_perm), _rowmajor(w_perm), M, Vt, K, Vd, _p(z), _rowmajor(z), _p(part),
                                    stride, ctypes.byref(vz), ctypes.byref(npart), _stream()), "sf_gemm_nt_teacher")
    return vz.value, npart.value


def gemm_nt_teacher_reduces(M: int, Vt: int, K: int, Vd: int) -> bool:
    return bool(_lib.lib().sf_gemm_nt_teacher_reduces(M, Vt, K, Vd))


def teacher_reduce_perm(z: torch.Tensor, *, Vt: int, Vd: int, perm, t2d_u8, loss_mask_pad, S: int, Spad: int, target_p_pad,
                        pod_scale_pad, tsum_pad, ids_pad, pos_mask_pad, part=None, nparts: int = 0, zmd_pad=None, zinv_pad=None):
    """teacher_reduce for logits with permuted columns (draft sub-vocabulary first; column c = vocabulary entry perm[c]).
    ``part`` [>= rows, part_stride, 4] fp32, ``nparts`` blocks per row in use: per-column-block partials of the columns z does not hold
    (sf_gemm_nt_teacher)."""
    L = _lib.lib()
    rows, Vz = z.shape
    assert perm.dtype == torch.int32 and perm.numel() == Vt
    part_stride = 0
    if nparts:
        assert part is not None and part.shape[0] >= rows and part.shape[1] >= nparts and part.is_contiguous()
        part_stride = part.shape[1]
    _lib.check(L.sf_teacher_reduce_perm(_p(z), _dt(z), _rowmajor(z), rows, Vz, Vt, Vd, _p(perm), _p(t2d_u8), _p(part) if nparts else None, nparts,
                                        part_stride, _p(loss_mask_pad), S, Spad, _p(target_p_pad), _p(pod_scale_pad),
                                        _p(tsum_pad), _p(ids_pad), _p(pos_mask_pad), _p(zmd_pad), _p(zinv_pad), _stream()),
               "sf_teacher_reduce_perm")


def rmsnorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float, y: torch.Tensor, rstd: Optional[torch.Tensor], *,
                ids_pad=None, S: int = 1, Spad: int = 1, off: int = 0, rows: Optional[int] = None):
    L = _lib.lib()
    H = w.numel()
    rows = y.shape[0] if rows is None else rows
    _lib.check(L.sf_rmsnorm_fwd(_p(x), _dt(x), _rowmajor(x), _p(ids_pad), S, Spad, off, _p(w), eps, rows, H, _p(y),
                                _rowmajor(y), _p(rstd), _stream()), "sf_rmsnorm_fwd")
    return y


def rmsnorm_fwd2(x: torch.Tensor, w1: torch.Tensor, y1: torch.Tensor, rstd1, w2: torch.Tensor, y2: torch.Tensor, rstd2, eps: float):
    """y1 = rmsnorm(x; w1), y2 = rmsnorm(x; w2) in one pass over x (same bits as two rmsnorm_fwd calls)"""
    L = _lib.lib()
    rows, H = y1.shape
    assert y2.shape == (rows, H) and w1.numel() == w2.numel() == H and x.dtype == y1.dtype == y2.dtype
    _lib.check(L.sf_rmsnorm_fwd2(_p(x), _dt(x), _rowmajor(x), _p(w1), _p(y1), _rowmajor(y1), _p(rstd1), _p(w2), _p(y2), _rowmajor(y2),
                                 _p(rstd2), eps, rows, H, _stream()), "sf_rmsnorm_fwd2")


def rmsnorm_bwd_workspace(rows: int, H: int) -> int:
    return int(_lib.lib().sf_rmsnorm_bwd_workspace_floats(rows, H))


def rmsnorm_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, rstd: torch.Tensor, *, dx=None, add=None, dw_acc=None,
                dw_accumulate: bool = True, workspace=None, ids_pad=None, S: int = 1, Spad: int = 1, off: int = 0, partial_only: bool = False):
    """``partial_only``: the weight gradient stops at the per-block partials in ``workspace`` [nb, H]; reduce them with ``colsum_accum``"""
    L = _lib.lib()
    rows, H = dy.shape[0], w.numel()
    _lib.check(L.sf_rmsnorm_bwd(_p(dy), _dt(dy), _rowmajor(dy), _p(x), _rowmajor(x), _p(ids_pad), S, Spad, off, _p(w),
                                _p(rstd), rows, H, _p(add), _rowmajor(add) if add is not None else 0, _p(dx),
                                _rowmajor(dx) if dx is not None else 0, _p(dw_acc), 2 if partial_only else (1 if dw_accumulate else 0),
                                _p(workspace), _stream()), "sf_rmsnorm_bwd")


def colsum_accum(partial: torch.Tensor, nb: int, H: int, acc: torch.Tensor, accumulate: bool):
    """acc[H] (= / +=) column sums of partial[nb, H] (fixed order); runs on the CURRENT stream"""
    L = _lib.lib()
    assert partial.dtype == torch.float32 and acc.dtype == torch.float32 and partial.numel() >= nb * H and acc.numel() >= H
    _lib.check(L.sf_colsum_accum(_p(partial), nb, H, _p(acc), 1 if accumulate else 0, _stream()), "sf_colsum_accum")


def rmsnorm_bwd2(dy1: torch.Tensor, w1: torch.Tensor, dw1_acc: torch.Tensor, dw1_accumulate: bool, dy2: torch.Tensor, w2: torch.Tensor,
                 dw2_acc: torch.Tensor, dw2_accumulate: bool, x: torch.Tensor, rstd: torch.Tensor, *, dx: torch.Tensor, add=None, workspace,
                 partial_only: bool = False):
    """dx = d_norm(dy1; w1) + d_norm(dy2; w2) (+ add) for two RMSNorms of the same rows x; dw*_acc (+)= their weight gradients.
    workspace: >= 2 * rmsnorm_bwd_workspace(rows, H) floats.  ``partial_only``: the weight gradients stop at the per-block partials [nb, H] --
    in the two halves of ``workspace``, or at ``dw1_acc`` / ``dw2_acc`` where those are given (reduce them with ``colsum_accum``)."""
    L = _lib.lib()
    rows, H = dy1.shape[0], w1.numel()
    assert dy2.shape[0] == rows and w2.numel() == H and workspace.numel() >= 2 * rmsnorm_bwd_workspace(rows, H)
    if partial_only:
        need = rmsnorm_bwd_workspace(rows, H)
        assert all(t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= need) for t in (dw1_acc, dw2_acc))
    m1, m2 = (2, 2) if partial_only else (1 if dw1_accumulate else 0, 1 if dw2_accumulate else 0)
    _lib.check(L.sf_rmsnorm_bwd2(_p(dy1), _rowmajor(dy1), _p(w1), _p(dw1_acc), m1, _p(dy2), _rowmajor(dy2), _p(w2),
                                 _p(dw2_acc), m2, _dt(dy1), _p(x), _rowmajor(x), _p(rstd), rows, H, _p(add),
                                 _rowmajor(add) if add is not None else 0, _p(dx), _rowmajor(dx), _p(workspace), _stream()),
               "sf_rmsnorm_bwd2")


def rope_(x: torch.Tensor, nheads: int, hd: int, cos_t: torch.Tensor, sin_t: torch.Tensor, pos_ids: torch.Tensor,
          pos_off: int, backward: bool = False):
    """in place on the first nheads*hd columns of the 2-D view x"""
    L = _lib.lib()
    assert cos_t.dtype == x.dtype and cos_t.is_contiguous() and sin_t.is_contiguous() and pos_ids.dtype == torch.int64
    _lib.check(L.sf_rope(_p(x), _dt(x), _rowmajor(x), x.shape[0], nheads, hd, _p(cos_t), _p(sin_t), _p(pos_ids), pos_off,
                         cos_t.shape[0], 1 if backward else 0, _stream()), "sf_rope")
    return x


def swiglu_fwd(gu: torch.Tensor, act: torch.Tensor):
    L = _lib.lib()
    rows, I = act.shape
    assert gu.shape == (rows, 2 * I)
    _lib.check(L.sf_swiglu_fwd(_p(gu), _dt(gu), _rowmajor(gu), rows, I, _p(act), _rowmajor(act), _stream()), "sf_swiglu_fwd")
    return act


def gemm_nt_swiglu_fwd(a: torch.Tensor, w_gu: torch.Tensor, gu: torch.Tensor, act: torch.Tensor):
    """gu = a @ w_gu.T ([M, 2I] gate|up) and act = round(silu(gate)) * up: the fused gate|up projection with SwiGLU in its
    epilogue (one launch for chip-filling shapes; gemm_nt + swiglu_fwd otherwise)"""
    L = _lib.lib()
    M, K = a.shape
    I = w_gu.shape[0] // 2
    assert w_gu.shape == (2 * I, K) and gu.shape == (M, 2 * I) and act.shape == (M, I)
    assert a.dtype == w_gu.dtype == gu.dtype == act.dtype == torch.bfloat16
    _lib.check(L.sf_gemm_nt_swiglu_fwd(_p(a), _rowmajor(a), _p(w_gu), _rowmajor(w_gu), M, I, K, _p(gu), _rowmajor(gu), _p(act),
                                       _rowmajor(act), _stream()), "sf_gemm_nt_swiglu_fwd")


def gemm_nt_swiglu_bwd(a: torch.Tensor, b: torch.Tensor, gu: torch.Tensor, dgu: torch.Tensor, dact: torch.Tensor):
    """dgu = d(SwiGLU)(a @ b.T, gu): the down-projection input gradient with the activation's backward in its epilogue
    (one launch for chip-filling shapes; gemm_nt + swiglu_bwd through `dact` otherwise)"""
    L = _lib.lib()
    M, K = a.shape
    I = b.shape[0]
    assert b.shape[1] == K and gu.shape == (M, 2 * I) and dgu.shape == (M, 2 * I) and dact.shape == (M, I)
    assert a.dtype == b.dtype == gu.dtype == dgu.dtype == dact.dtype == torch.bfloat16
    _lib.check(L.sf_gemm_nt_swiglu_bwd(_p(a), _rowmajor(a), _p(b), _rowmajor(b), M, I, K, _p(gu), _rowmajor(gu), _p(dgu),
                                       _rowmajor(dgu), _p(dact), _rowmajor(dact), _stream()), "sf_gemm_nt_swiglu_bwd")


def swiglu_bwd(dact: torch.Tensor, gu: torch.Tensor, dgu: torch.Tensor):
    L = _lib.lib()
    rows, I = dact.shape
    _lib.check(L.sf_swiglu_bwd(_p(dact), _dt(dact), _rowmajor(dact), _p(gu), _rowmajor(gu), rows, I, _p(dgu),
                               _rowmajor(dgu), _stream()), "sf_swiglu_bwd")
    return dgu


def transpose2d(inp: torch.Tensor, out: torch.Tensor):
    """out[C,R] = inp[R,C]^T for 2-D row-major views"""
    L = _lib.lib()
    R, C = inp.shape
    assert out.shape == (C, R)
    _lib.check(L.sf_transpose(_p(inp), _dt(inp), 0, 0, _rowmajor(inp), _p(out), 0, 0, _rowmajor(out), 1, 1, R, C,
                              _stream()), "sf_transpose")
    return out


def transpose_heads(inp: torch.Tensor, out: torch.Tensor, B: int, S: int, nheads: int, hd: int):
    """inp: [B*S, >= nheads*hd] view (heads at columns h*hd) -> out [B, nheads, hd, S] contiguous"""
    L = _lib.lib()
    ld = _rowmajor(inp)
    assert out.is_contiguous() and out.numel() == B * nheads * hd * S
    _lib.check(L.sf_transpose(_p(inp), _dt(inp), S * ld, hd, ld, _p(out), nheads * hd * S, hd * S, S, B, nheads, S, hd,
                              _stream()), "sf_transpose")
    return out


def cast_from_f32(inp: torch.Tensor, out: torch.Tensor, scale: float = 1.0):
    L = _lib.lib()
    rows, C = inp.shape
    assert inp.dtype == torch.float32 and out.shape == (rows, C)
    _lib.check(L.sf_cast_from_f32(_p(inp), _rowmajor(inp), _p(out), _dt(out), _rowmajor(out), rows, C, scale, _stream()),
               "sf_cast_from_f32")
    return out


def add_bf16(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor):
    L = _lib.lib()
    assert a.dtype == b.dtype == out.dtype == torch.bfloat16 and a.numel() == b.numel() == out.numel()
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
    _lib.check(L.sf_add_bf16(a.numel(), _p(a), _p(b), _p(out), _stream()), "sf_add_bf16")
    return out


def shift_accum(src: torch.Tensor, dst: torch.Tensor, *, B: int, S: int, Spad: int, off: int):
    """dst[b*Spad + s + off] += src[b*S + s]  (src bf16 [B*S, C], dst fp32 [B*Spad, C])"""
    L = _lib.lib()
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src.shape == (B * S, dst.shape[1])
    assert dst.shape[0] >= B * Spad
    _lib.check(L.sf_shift_accum(_p(src), _rowmajor(src), _p(dst), _rowmajor(dst), B, S, Spad, off, src.shape[1], _stream()),
               "sf_shift_accum")
    return dst


def split_bf16(x: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor):
    """hi = bf16(x), lo = bf16(x - hi)"""
    L = _lib.lib()
    assert x.dtype == torch.float32 and hi.dtype == lo.dtype == torch.bfloat16 and hi.shape == lo.shape == x.shape
    assert _rowmajor(hi) == _rowmajor(lo)
    _lib.check(L.sf_split_bf16(_p(x), _rowmajor(x), _p(hi), _p(lo), _rowmajor(hi), x.shape[0], x.shape[1], _stream()),
               "sf_split_bf16")


def shift_sum_split(src: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, *, T: int, B: int, S: int, Spad: int):
    """hi + lo ~= sum_k src[k*B*S + b*S + (p - k)] re-aligned to the padded positions p (src bf16 [>= T*B*S, C]; hi / lo bf16 [B*Spad, C])"""
    L = _lib.lib()
    assert src.dtype == hi.dtype == lo.dtype == torch.bfloat16 and src.shape[0] >= T * B * S
    assert hi.shape == lo.shape and hi.shape[0] >= B * Spad and hi.shape[1] == src.shape[1] and _rowmajor(hi) == _rowmajor(lo)
    _lib.check(L.sf_shift_sum_split(_p(src), _rowmajor(src), T, B, S, Spad, src.shape[1], _p(hi), _p(lo), _rowmajor(hi), _stream()),
               "sf_shift_sum_split")


def axpy_f32(alpha: float, x: torch.Tensor, y: torch.Tensor, accumulate: bool = True):
    L = _lib.lib()
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.numel() == y.numel()
    _lib.check(L.sf_axpy_f32(x.numel(), alpha, _p(x), _p(y), 1 if accumulate else 0, _stream()), "sf_axpy_f32")
    return y


MAX_DIAG = 32     # kMaxDiag of the attention kernels: diagonal branches per launch (ttt_length <= MAX_DIAG + 1)


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * max(1, len(ts)))()
    for i, t in enumerate(ts):
        arr[i] = _p(t).value if t is not None else None
    return arr


def attn_fwd(q, k0, v0, kd: List[torch.Tensor], vd: List[torch.Tensor], kv_len, o, lse, *, B, S, nh, nkv, hd, scale):
    L = _lib.lib()
    ldk = _rowmajor(k0)
    for t in [v0] + list(kd) + list(vd):
        assert _rowmajor(t) == ldk
    _lib.check(L.sf_attn_fwd(_p(q), _rowmajor(q), _p(k0), ldk, _p(v0), _ptr_array(kd), _ptr_array(vd), len(kd),
                             _p(kv_len), _p(o), _rowmajor(o), _p(lse), B, S, nh, nkv, hd, scale, _stream()), "sf_attn_fwd")


def attn_bwd_pre(q, o, dout, kd, vd, dkd, dvd, lse, delta, dq_init, *, B, S, nh, nkv, hd, scale, dk_last=None, dv_last=None):
    """``dk_last`` / ``dv_last`` (bf16 [B*S, nkv*hd] views, optional): the finished sums of the LAST branch of the list leave there,
    rounded once, instead of going back to dkd[-1] / dvd[-1] in fp32"""
    L = _lib.lib()
    ldk = _rowmajor(kd[0]) if kd else 0
    lddk = _rowmajor(dkd[0]) if dkd else 0
    ld_last = 0
    if dk_last is not None:
        assert dv_last is not None and dk_last.dtype == dv_last.dtype == torch.bfloat16 and _rowmajor(dk_last) == _rowmajor(dv_last)
        ld_last = _rowmajor(dk_last)
    _lib.check(L.sf_attn_bwd_pre(_p(q), _rowmajor(q), _p(o), _rowmajor(o), _p(dout), _rowmajor(dout), _ptr_array(kd),
                                 _ptr_array(vd), _ptr_array(dkd), _ptr_array(dvd), ldk, lddk, len(kd), _p(lse), _p(delta),
                                 _p(dq_init), B, S, nh, nkv, hd, scale, _p(dk_last), _p(dv_last), ld_last, _stream()),
               "sf_attn_bwd_pre")


DIAG_READ, DIAG_ACC, DIAG_X = 6, 4, 8      # sf_attn_bwd_diag: branches read / accumulating / later steps streamed per launch


def attn_bwd_diag(*, q=None, o=None, dout=None, lse=None, delta=None, dq_init=None, dq_accumulate=False, kd=(), vd=(), dkd=(), dvd=(),
                  first=(), dk_out=(), dv_out=(), xq=(), xdo=(), xlse=(), xdelta=(), B, S, nh, nkv, hd, scale):
    """One launch of the blocked diagonal-branch backward (include/specforge_amd.h: sf_attn_bwd_diag).  ``kd`` / ``vd``: the branches
    read (own step's dq); the first ``len(first)`` of them accumulate dK / dV -- ``dkd`` / ``dvd`` their fp32 sums (entries may be
    None when first AND final), ``first`` flags, ``dk_out`` / ``dv_out`` bf16 destinations of the final ones (None = not final).
    ``xq`` / ``xdo`` / ``xlse`` / ``xdelta``: later TTT steps streamed into the accumulating branches."""
    L = _lib.lib()
    nacc = len(first)
    assert len(kd) == len(vd) <= DIAG_READ and nacc <= min(DIAG_ACC, len(kd)) and len(xq) == len(xdo) == len(xlse) == len(xdelta) <= DIAG_X
    assert len(dkd) == len(dvd) == len(dk_out) == len(dv_out) == nacc
    some = lambda ts: next((t for t in ts if t is not None), None)
    ldk = _rowmajor(kd[0]) if kd else 0
    for t in list(kd) + list(vd):
        assert _rowmajor(t) == ldk
    acc0, out0 = some(list(dkd) + list(dvd)), some(list(dk_out) + list(dv_out))
    lddk = _rowmajor(acc0) if acc0 is not None else 0
    ld_out = _rowmajor(out0) if out0 is not None else 0
    for t in list(dkd) + list(dvd):
        assert t is None or (t.dtype == torch.float32 and _rowmajor(t) == lddk)
    for t in list(dk_out) + list(dv_out):
        assert t is None or (t.dtype == torch.bfloat16 and _rowmajor(t) == ld_out)
    src = q if q is not None else xq[0]
    ldq = _rowmajor(src)
    lddo = _rowmajor(dout if dout is not None else xdo[0])
    for t in xq:
        assert _rowmajor(t) == ldq
    for t in xdo:
        assert _rowmajor(t) == lddo
    flags = (ctypes.c_int * max(1, nacc))(*[1 if f else 0 for f in first])
    _lib.check(L.sf_attn_bwd_diag(_p(q), ldq, _p(o), _rowmajor(o) if o is not None else 0, _p(dout), lddo, _p(lse), _p(delta), _p(dq_init),
                                  1 if dq_accumulate else 0, _ptr_array(kd), _ptr_array(vd), ldk, len(kd), _ptr_array(dkd), _ptr_array(dvd),
                                  lddk, nacc, flags, _ptr_array(dk_out), _ptr_array(dv_out), ld_out, _ptr_array(xq), _ptr_array(xdo),
                                  _ptr_array(xlse), _ptr_array(xdelta), len(xq), B, S, nh, nkv, hd, scale, _stream()), "sf_attn_bwd_diag")


def attn_bwd_dq(q, dout, k0, v0, kv_len, lse, delta, dq_init, dq, *, B, S, nh, nkv, hd, scale):
    L = _lib.lib()
    _lib.check(L.sf_attn_bwd_dq(_p(q), _rowmajor(q), _p(dout), _rowmajor(dout), _p(k0), _rowmajor(k0), _p(v0),
                                _rowmajor(v0), _p(kv_len), _p(lse), _p(delta), _p(dq_init), _p(dq),
                                _rowmajor(dq), B, S, nh, nkv, hd, scale, _stream()), "sf_attn_bwd_dq")


def attn_bwd_dkv_workspace_floats(B, S, nh, nkv, hd) -> int:
    """fp32 workspace the head-split form of attn_bwd_dkv wants for this shape (0: it runs unsplit)"""
    return int(_lib.lib().sf_attn_bwd_dkv_workspace_floats(B, S, nh, nkv, hd))


def attn_bwd_dkv(q, dout, k0, v0, kv_len, lse, delta, dk, dv, *, B, S, nh, nkv, hd, scale, workspace=None):
    L = _lib.lib()
    assert dk.dtype == torch.float32 and dv.dtype == torch.float32 and _rowmajor(dk) == _rowmajor(dv)
    assert workspace is None or (workspace.dtype == torch.float32 and workspace.is_contiguous())
    _lib.check(L.sf_attn_bwd_dkv(_p(q), _rowmajor(q), _p(dout), _rowmajor(dout), _p(k0), _rowmajor(k0),
                                 _p(v0), _rowmajor(v0), _p(kv_len), _p(lse), _p(delta), _p(dk), _p(dv), _rowmajor(dk),
                                 B, S, nh, nkv, hd, scale, _p(workspace), workspace.numel() if workspace is not None else 0,
                                 _stream()), "sf_attn_bwd_dkv")


def grad_norm(g: torch.Tensor, norm_out: torch.Tensor, workspace: torch.Tensor, prescale: float = 1.0):
    L = _lib.lib()
    assert g.is_contiguous() and workspace.numel() >= L.sf_grad_norm_workspace_floats()
    _lib.check(L.sf_grad_norm(_p(g), _dt(g), g.numel(), prescale, _p(norm_out), _p(workspace), _stream()), "sf_grad_norm")
    return norm_out


def adamw_step(g, master, m, v, param, norm, *, max_norm, lr, beta1, beta2, eps, wd, step, grad_prescale=1.0):
    L = _lib.lib()
    n = g.numel()
    assert master.numel() == n and m.numel() == n and v.numel() == n and param.numel() == n
    _lib.check(L.sf_adamw_step(_p(g), _dt(g), _p(master), _p(m), _p(v), _p(param), n, _p(norm), max_norm, lr, beta1, beta2,
                               eps, wd, step, grad_prescale, _stream()), "sf_adamw_step")
