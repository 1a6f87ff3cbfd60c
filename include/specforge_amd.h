/* specforge_amd -- C-ABI of the MI355X-native EAGLE3 draft-training hot path.
 *
 * libsfhip.so exports exactly these symbols.  Every entry point takes raw device
 * pointers + sizes + element strides + a hipStream_t (as void*), enqueues work on that
 * stream and returns an int status (SF_OK == 0); sf_last_error() gives the message of
 * the last failing call on the calling thread.  No torch types, no ownership transfer:
 * all buffers are caller-owned and must stay alive until the stream has passed the call.
 *
 * The reference (sgl-project/SpecForge) is pure Python with no FFI: each entry point
 * below names the Python function(s) of the reference it replaces (paths relative to
 * the reference root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * dtype codes: SF_BF16 (bf16 storage, fp32 arithmetic) is the product path; SF_F32 is
 * accepted by the HBM-bound row kernels so they can be checked against fp32 goldens.
 * "padded" arrays are [B, Spad] (Spad = S + ttt_length) with zero-filled tails: the TTT
 * shift of step k is then the offset `off = k` into them.
 */
#ifndef SPECFORGE_AMD_H
#define SPECFORGE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_ERR_INVALID 1 /* argument / shape / alignment check failed */
#define SF_ERR_LAUNCH 2  /* HIP launch error */

#define SF_BF16 0
#define SF_F32 1

/* 6: sf_rmsnorm_bwd2 with dwX_accumulate == 2 WRITES [nb, H] partials to a non-null dwX_acc (ABI 5 ignored the pointer in that mode: an
 * ABI-5 caller passing its H-sized accumulator there must pass NULL or move to mode 0 / 1) */
#define SF_ABI_VERSION 6

int sf_abi_version(void);
/* 1 if this library is the SIMT-emulator test build, 0 for the gfx950 product build */
int sf_is_emulated(void);
const char* sf_last_error(void);

/* ---- bf16 GEMM, C[M,N] = alpha*A[M,K].B[N,K]^T (+beta*C) (+R) ------------------------------
 * replaces the ATen GEMMs behind nn.Linear: specforge/modeling/draft/llama3_eagle.py:555-566
 * (q/k/v/o_proj), 1513-1515 (gate/up/down), 1674-1678 (fc), 1691-1693 (lm_head),
 * specforge/modeling/target/target_head.py:31,100-101 (teacher head) and their autograd
 * dgrad/wgrad.  c_dtype: SF_BF16 or SF_F32.  R (bf16, optional): residual added after the
 * projection has been rounded to bf16 (llama3_eagle.py:1641,1650). */
int sf_gemm_nt(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N,
               int K, float alpha, float beta, const void* R, long ldr, void* stream);
/* sf_gemm_nt(alpha 1, beta 0) with an fp32 workspace (ABI 5): when the grid is under-filled -- at most half as many 256 x 256 tiles as CUs,
 * N a multiple of 256 (M may be ragged) -- the 4-wave kernel runs tiles x c work units, c = 2 ... 8 runs of K-tiles of >= 1024 each (the last
 * may be shorter), c picked by a small cost model among the splits with tiles x c <= CUs whose partials fit: split-K with fp32 partials in
 * `workspace` (c * roundup(M, 256) * N floats: 8 * roundup(M, 256) * N covers every split the launcher can pick), fixed-order reduce:
 * deterministic; the residual R joins after the single rounding as in sf_gemm_nt.  Any other shape, a NULL / too small workspace:
 * sf_gemm_nt.  Same result up to fp32 summation order. */
int sf_gemm_nt_ws(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N, int K,
                  const void* R, long ldr, float* workspace, long workspace_floats, void* stream);

/* Weight-gradient form: C[M,N] = alpha * A^T . B (+ beta * C) with A [K, M] and B [K, N] row-major (bf16), i.e.
 * dW = dY^T . X on the tensors exactly as the sweep produced them -- replaces autograd's `grad_output.t() @ input`
 * of every linear layer on the path (llama3_eagle.py:555-566,1513-1515,1674-1693) without materialising a
 * transposed operand.  K % 64 == 0 (pad the token dimension with zero rows), M, N multiples of 8.
 * workspace (optional, fp32, >= 2*M*N floats) + ksplit: 0 = the launcher splits K in two when the tile count would
 * leave the last round of CUs half empty (and the workspace is there), 1 = never split, 2 = always split (error if the
 * shape / workspace do not allow it); the partials are reduced in a fixed order (deterministic).
 * The LAST 4096 floats of a workspace that is large enough (4096 floats beyond the split-K partials, if any) are used as
 * pace-keeping counters: zeroed and incremented by every launch, they carry no data (timing only) and the result is
 * bit-identical with and without them.  A caller that wants pacing passes 2*M*N + 4096 floats (or 4096 when it never splits). */
int sf_gemm_tn(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N, int K,
               float alpha, float beta, float* workspace, long workspace_floats, int ksplit, void* stream);

/* Same GEMM with an fp32 row-mapped addend joined to the accumulator before the single rounding:
 *   C[r][n] = round( alpha * (A.B^T)[r][n] + Cadd[(r / S) * Spad + r % S + off][n] )
 * The TTT step k projects cat(input_layernorm(embed(ids << k)), hidden_norm(h_k)) through q/k/v
 * (specforge/modeling/draft/llama3_eagle.py:1625-1630, 661-700).  The embedding half of that product depends only
 * on the token, and step k's token at position s is step 0's token at s + k, so it is computed once over the
 * padded [B, S+T] positions (Cadd) and re-used by every step through this epilogue; the GEMM itself then
 * contracts over the hidden half only. */
int sf_gemm_nt_rowadd(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N,
                      int K, float alpha, const float* Cadd, long ldadd, int S, int Spad, int off,
                      float* workspace, long workspace_floats,   /* optional (ABI 5): split-K of under-filled grids as in sf_gemm_nt_ws */
                      void* stream);

/* ---- fused soft-target CE step: loss + in-place dlogits + accuracy + acceptance -----------
 * replaces specforge/core/loss.py:173-228 (LogSoftmaxLoss fwd/bwd), eagle3/model.py:161-173
 * (accuracy), core/lk_loss.py:43-80 (acceptance rate).  Per row r=b*S+s (padded index
 * pr=b*Spad+s+off): row_loss = pos_mask*(-sum_v p*log_softmax(x)); row_accept = pos_mask*
 * sum_v min(p*pod_scale, softmax(x)); row_correct = loss_mask*[argmax+d2t[argmax]==tgt_id];
 * if write_grad, logits are overwritten by grad_scale*pos_mask*(softmax*tsum - p).
 * pod_scale_pad / tgt_ids_pad / d2t / row_pred may be NULL; tsum_pad NULL => summed in-kernel. */
int sf_ce_fused(void* logits, int dtype, long ld, int rows, int V, const float* target, int S, int Spad, int off,
                const int* pos_mask_pad, const int* loss_mask_pad, const long long* tgt_ids_pad,
                const float* pod_scale_pad, const float* tsum_pad, const long long* d2t, float grad_scale,
                int write_grad, float* row_loss, float* row_correct, float* row_accept, int* row_pred,
                const long long* row_map, void* stream);

/* ---- LK-loss gradient (optional objective, lk_loss_type in {"alpha","lambda"}) ----------------
 * replaces the autograd path through specforge/core/lk_loss.py:43-99 selected at
 * specforge/algorithms/eagle3/model.py:78-96.  Overwrites the logits of one TTT step with
 * d(step_scale * lk_loss)/d(logits).  lk_mode 1 = "alpha" (-masked mean log acceptance), 2 = "lambda"
 * (w*KL + (1-w)(1-acceptance), w = kl_scale*exp(-kl_decay*acceptance) detached).  kl_row_scale is
 * the KL mean factor 1/(B*S); accept_sum / mask_sum are device scalars: sum_r pos_mask_r*accept_r
 * (third output of sf_ce_fused, reduced) and sum_r pos_mask_r over the rows of this step.
 * Masked rows get zero gradient; ties of torch.minimum split the gradient evenly as torch does. */
/* sf_ce_fused with the soft target given as the teacher's stored draft logits (ABI 4): zt [B*S, ldzt] bf16 in NATURAL token rows,
 * zmd_pad / zinv_pad [B, Spad] from sf_teacher_reduce_perm; row r = b*S+s of TTT step `off` uses target_p[j] = exp(zt[b*S+s+off][j] -
 * zmd) * zinv -- the expression the teacher kernel evaluates when it writes target_p, so results are bit-identical to sf_ce_fused on
 * that array; 2 bytes per element instead of 4, and [B, S, Vd] fp32 is never written or read.  tsum_pad is required.
 * row_map (ABI 5, optional; the same argument of sf_ce_fused): `rows` COMPACT logits rows -- row r of `logits` / of the row_* outputs is token row row_map[r] of the
 * [B, S] grid (loss-row compaction: only rows with a loss mask went through lm_head); targets and masks are addressed by token row. */
int sf_ce_fused_zt(void* logits, int dtype, long ld, int rows, int V, const void* zt, long ldzt, const float* zmd_pad,
                   const float* zinv_pad, int S, int Spad, int off, const int* pos_mask_pad, const int* loss_mask_pad,
                   const long long* tgt_ids_pad, const float* pod_scale_pad, const float* tsum_pad, const long long* d2t,
                   float grad_scale, int write_grad, float* row_loss, float* row_correct, float* row_accept, int* row_pred, const long long* row_map,
                   void* stream);
int sf_ce_lk_grad(void* logits, int dtype, long ld, int rows, int V, const float* target, int S, int Spad, int off,
                  const int* pos_mask_pad, const float* pod_scale_pad, const float* tsum_pad, int lk_mode,
                  float kl_scale, float kl_decay, float step_scale, float kl_row_scale, const float* accept_sum,
                  const float* mask_sum, void* stream);

/* out[i] = scale * sum(in[i*n : (i+1)*n]) in a fixed order (double accumulation): deterministic
 * replacement of the .sum()/.mean() reductions in eagle3/model.py:161-190. */
int sf_reduce_sum(const float* in, long n, int nsegments, float* out, float scale, void* stream);

/* The per-TTT-step metric scalars (eagle3/model.py:161-190; strategies/base.py:237-304) in one launch (ABI 4): met [T, 3] = the
 * per-step sums {row loss, correct, acceptance} (sf_reduce_sum of sf_ce_fused's row outputs); masks are the zero-padded
 * [B, Spad] arrays, step k reads them at offset k.  out [T, 8] = {ploss = loss_sum / (B S), acc_correct, acc_denom =
 * max(count(loss_mask), 1e-6), acc, acceptance_rate = accept_sum / max(count(position_mask), 1e-8), that denominator, B S, ploss again}. */
int sf_eagle3_metrics(const float* met, const int* loss_mask_pad, const int* pos_mask_pad, int B, int S, int Spad, int T, float* out,
                      void* stream);

/* ---- teacher soft targets from target logits ---------------------------------------------
 * replaces specforge/algorithms/eagle3/model.py:487-501 (_compute_target_p): argmax id
 * (lowest index on ties), position_mask = t2d[id]*loss_mask, target_p = softmax over the
 * draft sub-vocabulary {j + d2t[j]}, pod_scale with target_p_on_draft = target_p*pod_scale,
 * tsum = sum(target_p).  Outputs are written at padded index b*Spad+s. */
int sf_teacher_reduce(const void* z, int dtype, long ldz, int rows, int Vt, int Vd, const long long* d2t,
                      const unsigned char* t2d, const int* loss_mask_pad, int S, int Spad, float* target_p_pad,
                      float* pod_scale_pad, float* tsum_pad, long long* ids_pad, int* pos_mask_pad, void* stream);

/* The same reduction for logits whose COLUMNS ARE PERMUTED, draft sub-vocabulary first (ABI 4): column c of z is vocabulary entry
 * perm[c], perm[j] = j + d2t[j] for j < Vd, the other entries after them.  The caller permutes the rows of the frozen teacher head
 * once (specforge/modeling/target/target_head.py:93-101 computes logits = hidden . W^T: W[perm] gives this layout for free), so the
 * draft softmax reads Vd contiguous logits instead of gathering them through d2t.  Same outputs as sf_teacher_reduce (argmax in
 * original indices, lowest original index on ties).  z holds the first Vz >= Vd permuted columns; the columns from Vz on, if any,
 * arrive as `nparts` per-column-block partials {max, sum exp(z - max), argmax column, -} per row, block q of row r at
 * part[(r * part_stride + q) * 4] (written by sf_gemm_nt_teacher: those logits are never stored).
 * zmd_pad / zinv_pad (optional, together): the row's draft maximum and 1 / sum exp(z_draft - max) at padded index b*Spad+s, i.e.
 * target_p[j] = exp(z[j] - zmd) * zinv.  With them target_p_pad may be NULL: the probabilities are then never materialised and
 * sf_ce_fused_zt re-forms them from the stored draft logits. */
int sf_teacher_reduce_perm(const void* z, int dtype, long ldz, int rows, int Vz, int Vt, int Vd, const int* perm,
                           const unsigned char* t2d, const float* part, int nparts, long part_stride, const int* loss_mask_pad,
                           int S, int Spad, float* target_p_pad, float* pod_scale_pad, float* tsum_pad, long long* ids_pad,
                           int* pos_mask_pad, float* zmd_pad, float* zinv_pad, void* stream);

/* The teacher head GEMM for that layout (TargetHead.forward, target_head.py:93-101, on the row-permuted weight Wp [Vt, K]):
 * z = A . Wp^T in bf16.  With `part` given and a chip-filling shape, only the first *vz_out = roundup(Vd, 256) columns of z are
 * stored (the draft sub-vocabulary's logits); every later 128-column block of a row is reduced in the GEMM epilogue to one record
 * {max, sum exp(z - max), argmax column, 0} at part[(row * part_stride + block) * 4] (*nparts_out blocks <= part_stride) -- 3/4 of
 * the [rows, Vt] logits are never written or read back.  Otherwise (small shapes, part == NULL) all Vt columns are stored and
 * *nparts_out = 0.  z needs room for Vt columns either way.  Feed both to sf_teacher_reduce_perm (ABI 4). */
int sf_gemm_nt_teacher(const void* A, long lda, const void* Wp, long ldw, int M, int Vt, int K, int Vd, void* z, long ldz, float* part,
                       long part_stride, int* vz_out, int* nparts_out, void* stream);
/* 1 when sf_gemm_nt_teacher (given `part`) takes the reduced form for this shape: z then only needs roundup(Vd, 256) columns (ldz may be
 * that narrow), e.g. the rows of a persistent [tokens, roundup(Vd, 256)] array of teacher draft logits that sf_ce_fused_zt reads. */
int sf_gemm_nt_teacher_reduces(int M, int Vt, int K, int Vd);

/* ---- RMSNorm (+frozen-embedding gather) ---------------------------------------------------
 * replaces llama3_eagle.py:1561-1567 (LlamaRMSNorm.forward), 1759-1760 (embed_input_ids) and
 * the cat() placement of 1625-1630 (y/ldy address one half of the 2H-wide layer input).
 * ids_pad != NULL: row r reads x + ids_pad[b*Spad+s+off]*ldx (x is the embedding table). */
int sf_rmsnorm_fwd(const void* x, int dtype, long ldx, const long long* ids_pad, int S, int Spad, int off,
                   const void* w, float eps, int rows, int H, void* y, long ldy, float* rstd, void* stream);
/* two norms of the SAME rows with two weight vectors in one pass over x (ABI 4): y1 = w1 * round(x * rstd), y2 = w2 * round(x * rstd);
 * rstd1 / rstd2 (optional) both receive the row's rstd.  The final norm of TTT step k (llama3_eagle.py:1772-1777) and the hidden_norm of
 * step k + 1 (1625-1630) read the same hidden state.  Same bits as two sf_rmsnorm_fwd calls. */
int sf_rmsnorm_fwd2(const void* x, int dtype, long ldx, const void* w1, void* y1, long ldy1, float* rstd1, const void* w2, void* y2,
                    long ldy2, float* rstd2, float eps, int rows, int H, void* stream);
long sf_rmsnorm_bwd_workspace_floats(int rows, int H);
/* dx (optional) = add (optional) + d/dx; dw_acc[H] (optional, fp32) = or += d/dw.
 * dw_accumulate == 2 (ABI 5, here and in sf_rmsnorm_bwd2): only the per-block partials are written -- workspace[nb, H] with
 * nb = sf_rmsnorm_bwd_workspace_floats(rows, H) / H -- and the column sum is the caller's (sf_colsum_accum).  The engine uses it for the
 * three norm weights that every TTT step differentiates: the T launches of a sweep write their partials side by side and ONE column sum per
 * weight follows the sweep (engine.norm_colsum_batched). */
int sf_rmsnorm_bwd(const void* dy, int dtype, long lddy, const void* x, long ldx, const long long* ids_pad, int S,
                   int Spad, int off, const void* w, const float* rstd, int rows, int H, const void* add,
                   long ldadd, void* dx, long lddx, float* dw_acc, int dw_accumulate, float* workspace,
                   void* stream);

/* Two RMSNorm backwards of the SAME rows x in one pass (ABI 4): dx = d_norm(dy1; w1) + d_norm(dy2; w2) + add (optional), dw1_acc (+)=
 * d/dw1, dw2_acc (+)= d/dw2 (fp32).  H <= 4096; workspace = 2 x sf_rmsnorm_bwd_workspace_floats(rows, H) floats.  In the TTT sweep the
 * hidden state of step k feeds the final norm of step k - 1 (llama3_eagle.py:1772-1777) and the hidden_norm of step k (1625-1630): autograd
 * sums the two input gradients; run apart they read x twice and round the first partial sum to bf16.
 * With dwX_accumulate == 2 (ABI 6) the partials [nb, H] of weight X go to dwX_acc when that is non-null (a destination of the caller's choosing,
 * which must hold nb * H floats, NOT H: the
 * engine's per-weight arenas), else to the X-th half of the workspace. */
int sf_rmsnorm_bwd2(const void* dy1, long lddy1, const void* w1, float* dw1_acc, int dw1_accumulate, const void* dy2, long lddy2,
                    const void* w2, float* dw2_acc, int dw2_accumulate, int dtype, const void* x, long ldx, const float* rstd, int rows,
                    int H, const void* add, long ldadd, void* dx, long lddx, float* workspace, void* stream);

/* acc[H] (= or +=) the column sums of partial[nb, H] in a fixed order (deterministic) -- the second half of a norm backward's weight
 * gradient when it ran with dw_accumulate == 2 (ABI 5).  nb may span the partials of many launches (nb > 2048: a kernel with four times
 * the workgroups and row lanes). */
int sf_colsum_accum(const float* partial, int nb, int H, float* acc, int accumulate, void* stream);

/* ---- RoPE in place on `nheads` consecutive heads (llama3_eagle.py:133-142; positions
 * position_ids + pos_off as in 718-734); backward = transposed rotation. */
int sf_rope(void* x, int dtype, long ld, int rows, int nheads, int hd, const void* cos_t, const void* sin_t,
            const long long* pos_ids, int pos_off, int max_pos, int backward, void* stream);

/* ---- SwiGLU on a fused [rows, 2I] gate|up buffer (llama3_eagle.py:1518-1549) --------------- */
int sf_swiglu_fwd(const void* gu, int dtype, long ldgu, long rows, int I, void* act, long ldact, void* stream);
int sf_swiglu_bwd(const void* dact, int dtype, long lddact, const void* gu, long ldgu, long rows, int I, void* dgu,
                  long lddgu, void* stream);

/* The down-projection input gradient fused with d(SwiGLU) (llama3_eagle.py:1518-1549 under autograd: d(act) = dY . W_down,
 * then d(gate), d(up) from the saved gate / up): A [M, K] = dY, B [I, K] = W_down^T image, gu [M, 2I] saved gate|up,
 * dgu [M, 2I] out.  d(act) [M, I] is never written when the chip-filling kernel can fuse the two (whole 256 x 256 tiles);
 * `dact` [M, I] is the scratch the two-step form of every other shape goes through.  Same bits either way (ABI 3). */
int sf_gemm_nt_swiglu_bwd(const void* A, long lda, const void* B, long ldb, int M, int I, int K, const void* gu, long ldgu,
                          void* dgu, long lddgu, void* dact, long lddact, void* stream);

/* The fused gate|up projection with the activation in its epilogue (llama3_eagle.py:1518-1549 forward:
 * down_proj(act_fn(gate_proj(x)) * up_proj(x))): A [M, K] = x, Wgu [2I, K] = [gate_proj.weight ; up_proj.weight] (adjacent in the
 * flat parameter buffer), gu [M, 2I] = gate|up out (the backward needs both), act [M, I] = round(silu(gate)) * up out.  One launch
 * when the chip-filling kernel takes the shape (M % 256 == 0, I % 128 == 0, K % 64 == 0), sf_gemm_nt + sf_swiglu_fwd otherwise;
 * same bits either way (ABI 4). */
int sf_gemm_nt_swiglu_fwd(const void* A, long lda, const void* Wgu, long ldw, int M, int I, int K, void* gu, long ldgu,
                          void* act, long ldact, void* stream);

/* out[b1][b2][c][r] = in[b1][b2][r][c]: operand transposes for dgrad/wgrad and the K^T/V^T/Q^T/dO^T
 * images of the attention kernels (no reference equivalent: autograd transposes are views). */
int sf_transpose(const void* in, int dtype, long in_b1, long in_b2, long in_ld, void* out, long out_b1, long out_b2,
                 long out_ld, int nb1, int nb2, int R, int C, void* stream);
/* out = a + b over n bf16 elements (n % 8 == 0; fp32 add, one rounding): the gradient sum autograd forms where the
 * un-normed hidden state feeds both lm_head and the next TTT step when norm_output=False (llama3_eagle.py:1772-1777). */
int sf_add_bf16(long n, const void* a, const void* b, void* out, void* stream);
/* dst[b*Spad + s + off][:] += src[b*S + s][:] (bf16 -> fp32): running sum over the TTT steps of the q/k/v gradient
 * re-aligned to token positions, so that the embedding half of the QKV dgrad / wgrad (whose input is the same
 * token shifted by the step index, specforge/algorithms/eagle3/model.py:428-432) is contracted once, not T times. */
int sf_shift_accum(const void* src, long ldsrc, float* dst, long lddst, int B, int S, int Spad, int off, int C,
                   void* stream);
/* hi = bf16(x), lo = bf16(x - hi): two-term bf16 expansion of an fp32 matrix, so the summed gradient above enters the
 * bf16 MFMA GEMM with 16 mantissa bits (the reference adds T bf16 products in fp32; rounding the sum once to 8 bits
 * would be coarser than that). */
int sf_split_bf16(const float* in, long ldin, void* hi, void* lo, long ldout, long rows, int C, void* stream);
/* The two above over all T steps in one pass (ABI 4): src [T*N, C] bf16 = the per-step q/k/v gradients stacked by step (N = B*S
 * rows each: the weight-gradient stash), hi | lo [B*Spad, C] = the two-term expansion of
 * sum_k src[k*N + b*S + (p - k)] over the steps k with 0 <= p - k < S, added in fp32 for k = T-1 .. 0 (the order T calls of
 * sf_shift_accum with off = k ran them, so the bits are the same).  Reads every stash row once instead of an fp32
 * read-modify-write of the whole sum per step. */
int sf_shift_sum_split(const void* src, long ldsrc, int T, int B, int S, int Spad, int C, void* hi, void* lo, long ldout,
                       void* stream);
/* The inverse of a row compaction (ABI 5; loss-row compaction of the lm_head part, eagle3/model.py:364-433 scores row (b, s) of TTT step k
 * only where loss_mask[b, s + k] != 0): dst [rows, C] = src row inv[r] where inv[r] >= 0, zeros elsewhere.  C % 8 == 0, 16-byte aligned rows. */
int sf_rows_expand(const void* src, int dtype, long ld_src, const int* inv, void* dst, long ld_dst, long rows, int C, void* stream);
/* y = (accumulate ? y : 0) + alpha*x, fp32: carries norm-weight gradients across micro-steps. */
int sf_axpy_f32(long n, float alpha, const float* x, float* y, int accumulate, void* stream);
int sf_cast_from_f32(const float* in, long ldin, void* out, int dtype, long ldout, long rows, int C, float scale,
                     void* stream);

/* ---- TTT attention (llama3_eagle.py:745-778; lse-merge blueprint 1024-1151) ----------------
 * q/o/dout/dq: [B*S, nh*hd] views; k0/v0 and the diagonal-branch kd[i]/vd[i]: [B*S, nkv*hd]
 * views (transposed operand fragments are taken from the same tiles with ds_read_b64_tr_b16);
 * lse/delta: [B,nh,S] fp32; kv_len: [B] valid (right-padded) key count or NULL.  hd in {64, 128, 256}. */
int sf_attn_fwd(const void* q, long ldq, const void* k0, long ldk, const void* v0, const void* const* kd,
                const void* const* vd, int ndiag, const int* kv_len, void* o, long ldo, float* lse, int B, int S,
                int nh, int nkv, int hd, float scale, void* stream);
/* dk_last / dv_last (ABI 4, optional, both or neither): the sums of the LAST branch of the list (kd[ndiag-1]: the branch of the TTT
 * step whose backward this is -- no later launch adds to it) leave as bf16 [B*S, nkv*hd] (row stride ld_last) instead of going back
 * to dkd / dvd[ndiag-1] in fp32: the cast of the finished gradient happens where its last term is added. */
int sf_attn_bwd_pre(const void* q, long ldq, const void* o, long ldo, const void* dout, long lddo,
                    const void* const* kd, const void* const* vd, float* const* dkd, float* const* dvd, long ldk,
                    long lddk, int ndiag, const float* lse, float* delta, float* dq_init, int B, int S, int nh,
                    int nkv, int hd, float scale, void* dk_last, void* dv_last, long ld_last, void* stream);
/* The diagonal-branch backward, blocked (ABI 5; sf_attn_bwd_pre remains the one-pair-per-step form).  One launch serves sweep step s:
 *   own step (q non-null): delta = rowsum(dO * O); dq_init (=, or += with dq_accumulate) = sum over the `nread` <= 6 branches kd / vd
 *     of ds_i k_i, with p_i = exp(q.k_i scale - lse), ds_i = p_i (dO.v_i - delta) scale;
 *   the FIRST `nacc` <= 4 of those branches accumulate dK_i += ds_i q, dV_i += p_i dO (summed over the query heads of the kv group)
 *     over the own step AND over `nx` <= 8 later TTT steps given as (xq, xdo, xlse, xdelta) -- same strides as q / dout, lse / delta
 *     [B, nh, S] as written at those steps -- whose p / ds are recomputed;
 *   per accumulating branch j: first[j] != 0 -> the fp32 sums dkd[j] / dvd[j] are NOT read (first touch, no zero fill needed);
 *     dk_out[j] / dv_out[j] non-null -> the finished sums leave as bf16 [B*S, nkv*hd] (row stride ld_out) instead of going back to
 *     dkd[j] / dvd[j] (which may then be null when first[j] is set too).
 * A pair (step k, branch i) may run at any sweep step in [i, k]; specforge_amd/engine.py:diag_plan blocks them so that only the
 * pairs inside a block of 4 branches remain read-modify-write. */
int sf_attn_bwd_diag(const void* q, long ldq, const void* o, long ldo, const void* dout, long lddo, const float* lse, float* delta,
                     float* dq_init, int dq_accumulate, const void* const* kd, const void* const* vd, long ldk, int nread,
                     float* const* dkd, float* const* dvd, long lddk, int nacc, const int* first, void* const* dk_out,
                     void* const* dv_out, long ld_out, const void* const* xq, const void* const* xdo, const float* const* xlse,
                     const float* const* xdelta, int nx, int B, int S, int nh, int nkv, int hd, float scale, void* stream);
int sf_attn_bwd_dq(const void* q, long ldq, const void* dout, long lddo, const void* k0, long ldk, const void* v0,
                   long ldv, const int* kv_len, const float* lse, const float* delta,
                   const float* dq_init, void* dq, long lddq, int B, int S, int nh, int nkv, int hd, float scale,
                   void* stream);
/* workspace (ABI 5, optional): with few (batch, kv head, key block) workgroups -- B * nkv * ceil(S / 128) < 512, e.g. a bs 1 x 4096 recipe --
 * the query heads of a kv group are divided over several workgroups whose partial sums go through `workspace`
 * (sf_attn_bwd_dkv_workspace_floats floats, 16-byte aligned; 0 = this shape needs none) and are added to dk / dv in a fixed order
 * (deterministic).  NULL / too small: the unsplit kernel, same result up to fp32 summation order. */
long sf_attn_bwd_dkv_workspace_floats(int B, int S, int nh, int nkv, int hd);
int sf_attn_bwd_dkv(const void* q, long ldq, const void* dout, long lddo, const void* k0, long ldk, const void* v0, long ldv, const int* kv_len, const float* lse,
                    const float* delta, float* dk, float* dv, long lddk, int B, int S, int nh, int nkv, int hd,
                    float scale, float* workspace, long workspace_floats, void* stream);

/* ---- optimizer: BF16Optimizer.step on flat buffers (specforge/optimizer.py:95-168) ---------
 * norm_out[0] = prescale*sqrt(sum float(g)^2); adamw: clip = min(1, max_norm/(norm+1e-6))
 * (max_norm <= 0 disables), g32 = float(g)*clip*grad_prescale, torch.optim.AdamW update of the
 * fp32 master, param = storage(master). */
long sf_grad_norm_workspace_floats(void);
int sf_grad_norm(const void* g, int dtype, long n, float prescale, float* norm_out, float* workspace, void* stream);
int sf_adamw_step(const void* g, int dtype, float* master, float* m, float* v, void* param, long n,
                  const float* norm, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                  int step, float grad_prescale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPECFORGE_AMD_H */
