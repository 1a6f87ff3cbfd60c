// HBM-bound row kernels of the draft layer: RMSNorm (+embedding gather, +concat placement)
// forward/backward, RoPE forward/backward (in place), SwiGLU forward/backward, batched
// tile transpose, fp32->storage cast, and the fused clip + AdamW optimizer step.
//
// Replaces, for the EAGLE3 offline path of the reference:
//   specforge/modeling/draft/llama3_eagle.py:1561-1567  LlamaRMSNorm.forward (torch.compile)
//   specforge/modeling/draft/llama3_eagle.py:133-142    apply_rotary_pos_emb (torch.compile)
//   specforge/modeling/draft/llama3_eagle.py:1518-1549  LlamaMLP act_fn(gate) * up
//   specforge/modeling/draft/llama3_eagle.py:1759-1760  embed_input_ids (frozen gather, fused into the norm)
//   specforge/optimizer.py:95-168                       BF16Optimizer.step (norm, clip, AdamW, bf16 copy)
// plus the autograd backward of each.  All loads/stores are 16-byte vectors (8 bf16).
#include "sf_api_internal.h"
#include "sf_util.h"

namespace {

constexpr int kNormVecs = 4;  // 256 threads x 4 x 8 = 8192 columns max

// ------------------------------------------------------------------ RMSNorm
// y[r, :] = w * round_T(x[r, :] * rstd[r]); optional row gather x_row = table[ids_pad[b*Spad+s+off]]
// (w2 / y2 / rstd2_out, optional: a SECOND norm of the same row with another weight vector -- the final norm of TTT step k and the
// hidden_norm of step k + 1 both read h[k+1]; one pass over x, one row statistic, two outputs)
template <typename T>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
rmsnorm_fwd_kernel(const T* x, long ldx, const long long* ids_pad, int S, int Spad, int off, const T* w, float eps,
                   int H, T* y, long ldy, float* rstd_out, const T* w2 = nullptr, T* y2 = nullptr, long ldy2 = 0,
                   float* rstd2_out = nullptr) {
    SF_SHARED float red[16];
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x;
    const T* xr;
    if (ids_pad) {
        const int b = r / S, s = r - b * S;
        xr = x + ids_pad[(long)b * Spad + s + off] * ldx;
    } else {
        xr = x + (long)r * ldx;
    }
    float c[kNormVecs][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kNormVecs; ++i) {
        const int col = (tid + i * 256) * 8;
        if (col < H) {
            SfVec8<T>::ld(xr + col, c[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += c[i][j] * c[i][j];
        }
    }
    ss = sf_block_sum(ss, red);
    const float rstd = sf_rsqrt(ss / (float)H + eps);
    if (tid == 0 && rstd_out) rstd_out[r] = rstd;
    if (tid == 0 && rstd2_out) rstd2_out[r] = rstd;
    T* yr = y + (long)r * ldy;
    T* yr2 = y2 ? y2 + (long)r * ldy2 : nullptr;
#pragma unroll
    for (int i = 0; i < kNormVecs; ++i) {
        const int col = (tid + i * 256) * 8;
        if (col < H) {
            float wv[8], o[8], xh[8];
            SfVec8<T>::ld(w + col, wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { xh[j] = SfElem<T>::rnd(c[i][j] * rstd); o[j] = wv[j] * xh[j]; }
            SfVec8<T>::st(yr + col, o);
            if (yr2) {   // (uniform)
                SfVec8<T>::ld(w2 + col, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = wv[j] * xh[j];
                SfVec8<T>::st(yr2 + col, o);
            }
        }
    }
}

// dx = rstd * (g - xhat * mean(g*xhat)), g = dy*w, xhat = round_T(x*rstd); dw_partial += dy*xhat.
// dx (optional) gets `add` (optional residual-stream gradient) summed in.
// A workgroup walks `rows_per_block` rows; the loads of row r+1 (x, dy, add) are issued BEFORE the block-wide sum of row r,
// so the two HBM round trips of a row overlap the previous row's reduction and stores (one row at a time left the kernel
// latency-bound at 2.6 TB/s).  NV = 16-byte vectors per thread (H <= 2048 * NV).
template <typename T, int NV>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
rmsnorm_bwd_kernel(const T* dy, long lddy, const T* x, long ldx, const long long* ids_pad, int S, int Spad, int off,
                   const T* w, const float* rstd_in, int H, int R, int rows_per_block, const T* add, long ldadd, T* dx,
                   long lddx, float* dw_partial) {
    SF_SHARED float red[16];
    const int tid = (int)threadIdx.x;
    float dwacc[NV][8];
    float wv[NV][8];
    int colc[NV];     // this thread's columns; threads past H shadow column 0 (loads stay unconditional) and store nothing
    bool live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (tid + i * 256) * 8;
        live[i] = col < H;
        colc[i] = live[i] ? col : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwacc[i][j] = 0.f; wv[i][j] = 0.f; }
        if (live[i]) SfVec8<T>::ld(w + col, wv[i]);
    }
    const int r0 = (int)blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const bool has_add = add && dx;
    auto load_row = [&](int r, SfRaw8<T> (&xo)[NV], SfRaw8<T> (&dvo)[NV], SfRaw8<T> (&ao)[NV], float& rs) {
        const T* xr;
        if (ids_pad) {
            const int b = r / S, s = r - b * S;
            xr = x + ids_pad[(long)b * Spad + s + off] * ldx;
        } else {
            xr = x + (long)r * ldx;
        }
        rs = rstd_in[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xo[i].ld(xr + colc[i]);
            dvo[i].ld(dy + (long)r * lddy + colc[i]);
            if (has_add) ao[i].ld(add + (long)r * ldadd + colc[i]);   // uniform branch
        }
    };
    auto process = [&](int r, const SfRaw8<T> (&xc)[NV], const SfRaw8<T> (&dc)[NV], const SfRaw8<T> (&ac)[NV], float rs) {
        float g[NV][8], xh[NV][8];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dvj = live[i] ? dc[i].at(j) : 0.f;
                xh[i][j] = SfElem<T>::rnd(xc[i].at(j) * rs);
                g[i][j] = dvj * wv[i][j];
                dot += g[i][j] * xh[i][j];
                dwacc[i][j] += dvj * xh[i][j];
            }
        }
        if (dx) {
            dot = sf_block_sum(dot, red);
            const float cmean = dot / (float)H;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (g[i][j] - xh[i][j] * cmean);
                if (has_add) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += ac[i].at(j);
                }
                if (live[i]) SfVec8<T>::st(dx + (long)r * lddx + colc[i], o);
            }
        }
    };
    // groups of RG rows: all loads of a group are issued first, then the rows are reduced one after the other, so the
    // HBM round trip is paid once per group and the later rows land under the earlier rows' reductions.  (Prefetching
    // across loop trips does not work here: the compiler's wait-count insertion drains loop-carried loads with vmcnt(0).)
    constexpr int RG = NV <= 2 ? 4 : 2;
    for (int r = r0; r < r1; r += RG) {
        SfRaw8<T> xg[RG][NV], dg[RG][NV], ag[RG][NV];
        float rsg[RG];
#pragma unroll
        for (int u = 0; u < RG; ++u) load_row(r + u < r1 ? r + u : r1 - 1, xg[u], dg[u], ag[u], rsg[u]);
#pragma unroll
        for (int u = 0; u < RG; ++u)
            if (r + u < r1) process(r + u, xg[u], dg[u], ag[u], rsg[u]);
    }
    if (dw_partial) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (live[i]) SfVec8<float>::st(dw_partial + (long)blockIdx.x * H + colc[i], dwacc[i]);
    }
}

// Two norms of the SAME x in one backward pass: dx = d_norm1(dy1; w1) + d_norm2(dy2; w2) + add.  The backward of RMSNorm is linear in
// g = dy * w for a fixed x, so  dx = rstd * ((g1 + g2) - xhat * mean((g1 + g2) * xhat)) + add;  only the weight gradients stay apart
// (dw_i += dy_i * xhat).  In the TTT sweep h[k] feeds both the final norm of step k - 1 and the hidden_norm of step k: run apart they
// read x twice and pass a bf16 intermediate through HBM (and round it once more).  H <= 4096 (NV <= 2: the row groups stay in registers).
template <typename T, int NV>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
rmsnorm_bwd2_kernel(const T* dy1, long lddy1, const T* w1, const T* dy2, long lddy2, const T* w2, const T* x, long ldx,
                    const float* rstd_in, int H, int R, int rows_per_block, const T* add, long ldadd, T* dx, long lddx,
                    float* dw1_partial, float* dw2_partial) {
    SF_SHARED float red[16];
    const int tid = (int)threadIdx.x;
    float dwa1[NV][8], dwa2[NV][8], wv1[NV][8], wv2[NV][8];
    int colc[NV];
    bool live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (tid + i * 256) * 8;
        live[i] = col < H;
        colc[i] = live[i] ? col : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwa1[i][j] = 0.f; dwa2[i][j] = 0.f; wv1[i][j] = 0.f; wv2[i][j] = 0.f; }
        if (live[i]) { SfVec8<T>::ld(w1 + col, wv1[i]); SfVec8<T>::ld(w2 + col, wv2[i]); }
    }
    const int r0 = (int)blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    constexpr int RG = 2;     // rows whose loads are in flight together (see rmsnorm_bwd_kernel)
    for (int r = r0; r < r1; r += RG) {
        SfRaw8<T> xg[RG][NV], d1[RG][NV], d2[RG][NV], ag[RG][NV];
        float rsg[RG];
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            const int rr = r + u < r1 ? r + u : r1 - 1;
            rsg[u] = rstd_in[rr];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                xg[u][i].ld(x + (long)rr * ldx + colc[i]);
                d1[u][i].ld(dy1 + (long)rr * lddy1 + colc[i]);
                d2[u][i].ld(dy2 + (long)rr * lddy2 + colc[i]);
                if (add) ag[u][i].ld(add + (long)rr * ldadd + colc[i]);   // uniform branch
            }
        }
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            if (r + u >= r1) break;
            const float rs = rsg[u];
            float g[NV][8], xh[NV][8];
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = live[i] ? d1[u][i].at(j) : 0.f, bb = live[i] ? d2[u][i].at(j) : 0.f;
                    xh[i][j] = SfElem<T>::rnd(xg[u][i].at(j) * rs);
                    g[i][j] = a * wv1[i][j] + bb * wv2[i][j];
                    dot += g[i][j] * xh[i][j];
                    dwa1[i][j] += a * xh[i][j];
                    dwa2[i][j] += bb * xh[i][j];
                }
            }
            dot = sf_block_sum(dot, red);
            const float cmean = dot / (float)H;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (g[i][j] - xh[i][j] * cmean);
                if (add) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += ag[u][i].at(j);
                }
                if (live[i]) SfVec8<T>::st(dx + (long)(r + u) * lddx + colc[i], o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (live[i]) {
            SfVec8<float>::st(dw1_partial + (long)blockIdx.x * H + colc[i], dwa1[i]);
            SfVec8<float>::st(dw2_partial + (long)blockIdx.x * H + colc[i], dwa2[i]);
        }
}

// acc[col] (+)= sum_b partial[b][col]   (deterministic: fixed order)
// 1024 threads = 64 columns x 16 row lanes; lane order of the final sum is fixed.
SF_GLOBAL void colsum_accum_kernel(const float* partial, int nb, int H, float* acc, int accumulate) {
    SF_SHARED float red[16][64];
    const int cl = (int)threadIdx.x & 63, rl = (int)threadIdx.x >> 6;
    const int col = (int)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < H)
        for (int b = rl; b < nb; b += 16) s += partial[(long)b * H + col];
    red[rl][cl] = s;
    sf_syncthreads();
    if (rl == 0 && col < H) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        acc[col] = accumulate ? acc[col] + t : t;
    }
}

// The same sum over MANY partial rows (the engine reduces a norm weight's partials of all TTT steps in one launch: 7 x 1024 rows at the
// headline shape): 16 columns x 64 row lanes per workgroup, H / 16 workgroups -- four times the workgroups and row lanes of the kernel
// above, which is latency-bound on a long column (64 workgroups, each lane walking nb / 16 rows).  64-byte row segments: half a cache
// line per row and workgroup, the neighbouring workgroup takes the other half.  Fixed order as above.
SF_GLOBAL void colsum_accum_tall_kernel(const float* partial, int nb, int H, float* acc, int accumulate) {
    SF_SHARED float red[64][16];
    const int cl = (int)threadIdx.x & 15, rl = (int)threadIdx.x >> 4;
    const int col = (int)blockIdx.x * 16 + cl;
    float s = 0.f;
    if (col < H)
        for (int b = rl; b < nb; b += 64) s += partial[(long)b * H + col];
    red[rl][cl] = s;
    sf_syncthreads();
    if (rl == 0 && col < H) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) t += red[i][cl];
        acc[col] = accumulate ? acc[col] + t : t;
    }
}

// --------------------------------------------------------------------- RoPE
// In place on `nheads` consecutive heads of width hd starting at column 0 of row r (row stride ld).
// forward : y1 = x1*c1 - x2*s1 ; y2 = x2*c2 + x1*s2     (q*cos + rotate_half(q)*sin, neox halves)
// backward: dx1 = dy1*c1 + dy2*s2 ; dx2 = dy2*c2 - dy1*s1
template <typename T>
SF_GLOBAL void rope_kernel(T* x, long ld, int nheads, int hd, const T* cos_t, const T* sin_t, const long long* pos_ids,
                           int pos_off, int max_pos, int backward) {
    const int r = (int)blockIdx.x;
    long pos = pos_ids[r] + pos_off;
    if (pos < 0) pos = 0;
    if (pos >= max_pos) pos = max_pos - 1;
    const T* cr = cos_t + pos * hd;
    const T* sr = sin_t + pos * hd;
    const int half = hd >> 1, upr = half >> 3;  // 8-wide units per head-half
    T* xr = x + (long)r * ld;
    for (int u = (int)threadIdx.x; u < nheads * upr; u += (int)blockDim.x) {
        const int h = u / upr, j = (u - h * upr) * 8;
        T* p1 = xr + h * hd + j;
        T* p2 = p1 + half;
        float x1[8], x2[8], c1[8], c2[8], s1[8], s2[8], o1[8], o2[8];
        SfVec8<T>::ld(p1, x1);
        SfVec8<T>::ld(p2, x2);
        SfVec8<T>::ld(cr + j, c1);
        SfVec8<T>::ld(cr + half + j, c2);
        SfVec8<T>::ld(sr + j, s1);
        SfVec8<T>::ld(sr + half + j, s2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!backward) {
                o1[i] = SfElem<T>::rnd(x1[i] * c1[i]) + SfElem<T>::rnd(-x2[i] * s1[i]);
                o2[i] = SfElem<T>::rnd(x2[i] * c2[i]) + SfElem<T>::rnd(x1[i] * s2[i]);
            } else {
                o1[i] = x1[i] * c1[i] + x2[i] * s2[i];
                o2[i] = x2[i] * c2[i] - x1[i] * s1[i];
            }
        }
        SfVec8<T>::st(p1, o1);
        SfVec8<T>::st(p2, o2);
    }
}

// ------------------------------------------------------------------- SwiGLU
template <typename T>
SF_GLOBAL void swiglu_fwd_kernel(const T* gu, long ldgu, int I, long rows, T* act, long ldact) {
    const long upr = I >> 3, total = rows * upr;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const long r = u / upr;
        const int j = (int)(u - r * upr) * 8;
        float g[8], up[8], o[8];
        SfVec8<T>::ld(gu + r * ldgu + j, g);
        SfVec8<T>::ld(gu + r * ldgu + I + j, up);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = sf_swiglu_fwd_elem<T>(g[i], up[i]);
        SfVec8<T>::st(act + r * ldact + j, o);
    }
}
template <typename T>
SF_GLOBAL void swiglu_bwd_kernel(const T* dact, long lddact, const T* gu, long ldgu, int I, long rows, T* dgu, long lddgu) {
    const long upr = I >> 3, total = rows * upr;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const long r = u / upr;
        const int j = (int)(u - r * upr) * 8;
        float g[8], up[8], da[8], dg[8], du[8];
        SfVec8<T>::ld(gu + r * ldgu + j, g);
        SfVec8<T>::ld(gu + r * ldgu + I + j, up);
        SfVec8<T>::ld(dact + r * lddact + j, da);
#pragma unroll
        for (int i = 0; i < 8; ++i) sf_swiglu_bwd_elem<T>(g[i], up[i], da[i], dg[i], du[i]);
        SfVec8<T>::st(dgu + r * lddgu + j, dg);
        SfVec8<T>::st(dgu + r * lddgu + I + j, du);
    }
}

// ---------------------------------------------------------------- transpose
// out[b1][b2][c][r] = in[b1][b2][r][c]   (R rows x C cols, 2-level batch, unit inner strides)
template <typename T>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
transpose_kernel(const T* in, long in_b1, long in_b2, long in_ld, T* out, long out_b1, long out_b2, long out_ld, int R,
                 int C, int nb2) {
    SF_SHARED T tile[64][72];
    const int bz = (int)blockIdx.z;
    const int b1 = bz / nb2, b2 = bz - b1 * nb2;
    const T* src = in + b1 * in_b1 + b2 * in_b2;
    T* dst = out + b1 * out_b1 + b2 * out_b2;
    const int r0 = (int)blockIdx.y * 64, c0 = (int)blockIdx.x * 64, tid = (int)threadIdx.x;
    for (int v = tid; v < 512; v += 256) {
        const int rr = v >> 3, cc = (v & 7) * 8;
        float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + rr < R && c0 + cc < C) SfVec8<T>::ld(src + (long)(r0 + rr) * in_ld + c0 + cc, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) SfElem<T>::st(&tile[rr][cc + j], t[j]);
    }
    sf_syncthreads();
    for (int v = tid; v < 512; v += 256) {
        const int cc = v >> 3, rr = (v & 7) * 8;  // output row = input col
        if (c0 + cc < C && r0 + rr < R) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = SfElem<T>::ld(&tile[rr + j][cc]);
            SfVec8<T>::st(dst + (long)(c0 + cc) * out_ld + r0 + rr, t);
        }
    }
}

// out[r, c] = (T) in[r, c] * scale    (fp32 accumulators -> storage dtype, strided)
template <typename T>
SF_GLOBAL void cast_from_f32_kernel(const float* in, long ldin, T* out, long ldout, long rows, int C, float scale) {
    const long upr = C >> 3, total = rows * upr;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const long r = u / upr;
        const int j = (int)(u - r * upr) * 8;
        float t[8];
        SfVec8<float>::ld(in + r * ldin + j, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] *= scale;
        SfVec8<T>::st(out + r * ldout + j, t);
    }
}

// ---------------------------------------------------------------- optimizer
// partial[b] = sum over this block's slice of float(g)^2
template <typename T>
SF_GLOBAL void sumsq_kernel(const T* g, long n, float* partial) {
    SF_SHARED float red[16];
    float acc = 0.f;
    const long n8 = n >> 3;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < n8; u += (long)gridDim.x * blockDim.x) {
        float t[8];
        SfVec8<T>::ld(g + u * 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += t[i] * t[i];
    }
    if (blockIdx.x == 0)
        for (long j = n8 * 8 + threadIdx.x; j < n; j += blockDim.x) {
            float v = SfElem<T>::ld(g + j);
            acc += v * v;
        }
    acc = sf_block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
SF_GLOBAL void norm_finish_kernel(const float* partial, int nb, float* norm_out, float prescale) {
    SF_SHARED double part[256];
    const int tid = (int)threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < nb; i += 256) acc += (double)partial[i];
    part[tid] = acc;
    sf_syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (tid < sft) part[tid] += part[tid + sft];
        sf_syncthreads();
    }
    if (tid == 0) norm_out[0] = (float)(sqrt(part[0]) * (double)prescale);
}
// BF16Optimizer.step on flat buffers (optimizer.py:95-168): clip = min(1, max_norm/(norm+1e-6)),
// g32 = float(g)*clip, torch.optim.AdamW update of the fp32 master, param = T(master).
template <typename T>
SF_GLOBAL void adamw_kernel(const T* g, float* master, float* m, float* v, T* param, long n, const float* norm,
                            float max_norm, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                            float bc2_sqrt, float grad_prescale, int vec) {
    float clip = 1.0f;
    if (max_norm > 0.f) clip = fminf(1.0f, max_norm / (norm[0] + 1e-6f));
    const float gsc = clip * grad_prescale;
    const float step_size = lr / bc1;
    auto update = [&](float g32, float& p, float& mi, float& vi) {
        g32 *= gsc;
        p = p * (1.0f - lr * wd);
        mi = mi * beta1 + g32 * (1.0f - beta1);
        vi = vi * beta2 + g32 * g32 * (1.0f - beta2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p = p - step_size * (mi / denom);
    };
    // 28 bytes per parameter, nothing else: 8 parameters per thread and trip, every stream as 16-byte accesses (round 4: the scalar
    // form -- 2- and 4-byte accesses, one parameter per trip -- ran at 4.8 TB/s; same arithmetic per element, bit-identical results)
    const long n8 = vec ? n >> 3 : 0;      // (vec = 0: some stream is not 16-byte aligned -- one parameter per trip)
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < n8; u += (long)gridDim.x * blockDim.x) {
        const long i = u * 8;
        float gg[8], pp[8], mm[8], vv[8];
        SfVec8<T>::ld(g + i, gg);
        SfVec8<float>::ld(master + i, pp);
        SfVec8<float>::ld(m + i, mm);
        SfVec8<float>::ld(v + i, vv);
#pragma unroll
        for (int e = 0; e < 8; ++e) update(gg[e], pp[e], mm[e], vv[e]);
        SfVec8<float>::st(master + i, pp);
        SfVec8<float>::st(m + i, mm);
        SfVec8<float>::st(v + i, vv);
        SfVec8<T>::st(param + i, pp);
    }
    for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        {
            float p = master[i], mi = m[i], vi = v[i];
            update(SfElem<T>::ld(g + i), p, mi, vi);
            master[i] = p;
            m[i] = mi;
            v[i] = vi;
            SfElem<T>::st(param + i, p);
        }
    }
}

inline int grid_for(long units, int block = 256, int cap = 256 * 8) {
    long g = (units + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

#define SF_DISPATCH_T(dtype, CALL)                          \
    do {                                                    \
        if ((dtype) == SF_BF16) { typedef sf_bf16 T; CALL; } \
        else if ((dtype) == SF_F32) { typedef float T; CALL; } \
        else SF_CHECK_ARG(false, "unsupported dtype");      \
    } while (0)

extern "C" int sf_rmsnorm_fwd(const void* x, int dtype, long ldx, const long long* ids_pad, int S, int Spad, int off,
                              const void* w, float eps, int rows, int H, void* y, long ldy, float* rstd, void* stream) {
    SF_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 256 * 8 * kNormVecs, "sf_rmsnorm_fwd: H must be a multiple of 8, <= 8192");
    SF_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0, "sf_rmsnorm_fwd: strides must be multiples of 8");
    SF_CHECK_ARG(!ids_pad || (S > 0 && Spad >= S + off), "sf_rmsnorm_fwd: gather shape");
    if (rows == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((rmsnorm_fwd_kernel<T>), dim3(rows), dim3(256), 0, stream, (const T*)x, ldx, ids_pad, S,
                                   Spad, off, (const T*)w, eps, H, (T*)y, ldy, rstd));
    return sf_check_launch("sf_rmsnorm_fwd");
}

extern "C" int sf_rmsnorm_fwd2(const void* x, int dtype, long ldx, const void* w1, void* y1, long ldy1, float* rstd1, const void* w2,
                               void* y2, long ldy2, float* rstd2, float eps, int rows, int H, void* stream) {
    SF_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 256 * 8 * kNormVecs, "sf_rmsnorm_fwd2: H must be a multiple of 8, <= 8192");
    SF_CHECK_ARG(ldx % 8 == 0 && ldy1 % 8 == 0 && ldy2 % 8 == 0 && w1 && y1 && w2 && y2, "sf_rmsnorm_fwd2: strides must be multiples of 8; both outputs required");
    if (rows == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((rmsnorm_fwd_kernel<T>), dim3(rows), dim3(256), 0, stream, (const T*)x, ldx, (const long long*)nullptr, 1,
                                   1, 0, (const T*)w1, eps, H, (T*)y1, ldy1, rstd1, (const T*)w2, (T*)y2, ldy2, rstd2));
    return sf_check_launch("sf_rmsnorm_fwd2");
}

extern "C" long sf_rmsnorm_bwd_workspace_floats(int rows, int H) {
    const int rpb = 16;
    return (long)((rows + rpb - 1) / rpb) * H;
}

extern "C" int sf_rmsnorm_bwd(const void* dy, int dtype, long lddy, const void* x, long ldx, const long long* ids_pad,
                              int S, int Spad, int off, const void* w, const float* rstd, int rows, int H,
                              const void* add, long ldadd, void* dx, long lddx, float* dw_acc, int dw_accumulate,
                              float* workspace, void* stream) {
    SF_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 256 * 8 * kNormVecs, "sf_rmsnorm_bwd: H must be a multiple of 8, <= 8192");
    SF_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && ldadd % 8 == 0, "sf_rmsnorm_bwd: strides");
    const bool partial_only = dw_accumulate == 2;
    SF_CHECK_ARG((!dw_acc && !partial_only) || workspace, "sf_rmsnorm_bwd: workspace required for dw");
    if (rows == 0) return 0;
    const int rpb = 16, nb = (rows + rpb - 1) / rpb;
#define SF_NORM_BWD(NV)                                                                                                    \
    SF_DISPATCH_T(dtype, SF_LAUNCH((rmsnorm_bwd_kernel<T, NV>), dim3(nb), dim3(256), 0, stream, (const T*)dy, lddy, (const T*)x, \
                                   ldx, ids_pad, S, Spad, off, (const T*)w, rstd, H, rows, rpb, (const T*)add, ldadd,      \
                                   (T*)dx, lddx, (dw_acc || partial_only) ? workspace : (float*)nullptr))
    if (H <= 2048) { SF_NORM_BWD(1); } else if (H <= 4096) { SF_NORM_BWD(2); } else { SF_NORM_BWD(kNormVecs); }
#undef SF_NORM_BWD
    if (dw_acc && !partial_only)
        SF_LAUNCH(colsum_accum_kernel, dim3((H + 63) / 64), dim3(1024), 0, stream, (const float*)workspace, nb, H, dw_acc,
                  dw_accumulate);
    return sf_check_launch("sf_rmsnorm_bwd");
}

// The column sum of a norm backward's per-block partials as a call of its own (ABI 5): with dw_accumulate == 2 the two entry points above
// only WRITE their partials (workspace: sf_rmsnorm_bwd_workspace_floats per weight vector, or the caller's destinations in sf_rmsnorm_bwd2)
// and the caller reduces them when it likes.  The engine keeps the partials of a sweep's T launches side by side per weight and reduces them
// once after the sweep (round 5: 3 column sums instead of 21 per 7-step sweep, -0.2 ... -0.4 ms per step in the same-process A/B,
// profiles/r5_norm_colsum_batched_ab.jsonl).  Reducing on a SIDE STREAM was measured in round 4 and rejected: the persistent GEMMs hold
// every CU, so a side-stream kernel only delays its neighbours.
extern "C" int sf_colsum_accum(const float* partial, int nb, int H, float* acc, int accumulate, void* stream) {
    SF_CHECK_ARG(partial && acc && nb >= 0 && H > 0, "sf_colsum_accum: bad args");
    if (nb > 2048)      // (several launches' partials at once)
        SF_LAUNCH(colsum_accum_tall_kernel, dim3((H + 15) / 16), dim3(1024), 0, stream, partial, nb, H, acc, accumulate ? 1 : 0);
    else
        SF_LAUNCH(colsum_accum_kernel, dim3((H + 63) / 64), dim3(1024), 0, stream, partial, nb, H, acc, accumulate ? 1 : 0);
    return sf_check_launch("sf_colsum_accum");
}

extern "C" int sf_rmsnorm_bwd2(const void* dy1, long lddy1, const void* w1, float* dw1_acc, int dw1_accumulate, const void* dy2,
                               long lddy2, const void* w2, float* dw2_acc, int dw2_accumulate, int dtype, const void* x, long ldx,
                               const float* rstd, int rows, int H, const void* add, long ldadd, void* dx, long lddx,
                               float* workspace, void* stream) {
    SF_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 4096, "sf_rmsnorm_bwd2: H must be a multiple of 8, <= 4096");
    SF_CHECK_ARG(lddy1 % 8 == 0 && lddy2 % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && ldadd % 8 == 0, "sf_rmsnorm_bwd2: strides");
    SF_CHECK_ARG(dy1 && dy2 && w1 && w2 && x && rstd && dx && (dw1_acc || dw1_accumulate == 2) && (dw2_acc || dw2_accumulate == 2) && workspace,
                 "sf_rmsnorm_bwd2: missing argument");
    if (rows == 0) return 0;
    const int rpb = 16, nb = (rows + rpb - 1) / rpb;
    // per-block partials of the two weight gradients: the two halves of the workspace (2 x sf_rmsnorm_bwd_workspace_floats(rows, H)) -- or,
    // with dwX_accumulate == 2 and a non-null dwX_acc, [nb, H] at dwX_acc (a caller collecting the partials of many launches side by side)
    float* ws1 = (dw1_accumulate == 2 && dw1_acc) ? dw1_acc : workspace;
    float* ws2 = (dw2_accumulate == 2 && dw2_acc) ? dw2_acc : workspace + (long)nb * H;
#define SF_NORM_BWD2(NV)                                                                                                     \
    SF_DISPATCH_T(dtype, SF_LAUNCH((rmsnorm_bwd2_kernel<T, NV>), dim3(nb), dim3(256), 0, stream, (const T*)dy1, lddy1, (const T*)w1, \
                                   (const T*)dy2, lddy2, (const T*)w2, (const T*)x, ldx, rstd, H, rows, rpb, (const T*)add, ldadd,  \
                                   (T*)dx, lddx, ws1, ws2))
    if (H <= 2048) { SF_NORM_BWD2(1); } else { SF_NORM_BWD2(2); }
#undef SF_NORM_BWD2
    if (dw1_accumulate != 2)
        SF_LAUNCH(colsum_accum_kernel, dim3((H + 63) / 64), dim3(1024), 0, stream, (const float*)ws1, nb, H, dw1_acc, dw1_accumulate);
    if (dw2_accumulate != 2)
        SF_LAUNCH(colsum_accum_kernel, dim3((H + 63) / 64), dim3(1024), 0, stream, (const float*)ws2, nb, H, dw2_acc, dw2_accumulate);
    return sf_check_launch("sf_rmsnorm_bwd2");
}

extern "C" int sf_rope(void* x, int dtype, long ld, int rows, int nheads, int hd, const void* cos_t, const void* sin_t,
                       const long long* pos_ids, int pos_off, int max_pos, int backward, void* stream) {
    SF_CHECK_ARG(rows >= 0 && nheads > 0 && hd % 16 == 0 && ld % 8 == 0 && max_pos > 0, "sf_rope: bad shape");
    if (rows == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((rope_kernel<T>), dim3(rows), dim3(256), 0, stream, (T*)x, ld, nheads, hd, (const T*)cos_t,
                                   (const T*)sin_t, pos_ids, pos_off, max_pos, backward));
    return sf_check_launch("sf_rope");
}

extern "C" int sf_swiglu_fwd(const void* gu, int dtype, long ldgu, long rows, int I, void* act, long ldact, void* stream) {
    SF_CHECK_ARG(rows >= 0 && I > 0 && I % 8 == 0 && ldgu % 8 == 0 && ldact % 8 == 0, "sf_swiglu_fwd: bad shape");
    if (rows == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((swiglu_fwd_kernel<T>), dim3(grid_for(rows * (I / 8))), dim3(256), 0, stream, (const T*)gu,
                                   ldgu, I, rows, (T*)act, ldact));
    return sf_check_launch("sf_swiglu_fwd");
}

extern "C" int sf_swiglu_bwd(const void* dact, int dtype, long lddact, const void* gu, long ldgu, long rows, int I,
                             void* dgu, long lddgu, void* stream) {
    SF_CHECK_ARG(rows >= 0 && I > 0 && I % 8 == 0 && ldgu % 8 == 0 && lddact % 8 == 0 && lddgu % 8 == 0, "sf_swiglu_bwd: bad shape");
    if (rows == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((swiglu_bwd_kernel<T>), dim3(grid_for(rows * (I / 8))), dim3(256), 0, stream,
                                   (const T*)dact, lddact, (const T*)gu, ldgu, I, rows, (T*)dgu, lddgu));
    return sf_check_launch("sf_swiglu_bwd");
}

extern "C" int sf_transpose(const void* in, int dtype, long in_b1, long in_b2, long in_ld, void* out, long out_b1,
                            long out_b2, long out_ld, int nb1, int nb2, int R, int C, void* stream) {
    SF_CHECK_ARG(nb1 >= 1 && nb2 >= 1 && R >= 0 && C >= 0 && R % 8 == 0 && C % 8 == 0, "sf_transpose: R and C must be multiples of 8");
    SF_CHECK_ARG(in_ld % 8 == 0 && out_ld % 8 == 0 && in_b1 % 8 == 0 && in_b2 % 8 == 0 && out_b1 % 8 == 0 && out_b2 % 8 == 0,
                 "sf_transpose: strides must be multiples of 8");
    if (R == 0 || C == 0) return 0;
    SF_CHECK_ARG((long)nb1 * nb2 <= 65535, "sf_transpose: too many batches");
    dim3 grid((C + 63) / 64, (R + 63) / 64, nb1 * nb2);
    SF_DISPATCH_T(dtype, SF_LAUNCH((transpose_kernel<T>), grid, dim3(256), 0, stream, (const T*)in, in_b1, in_b2, in_ld,
                                   (T*)out, out_b1, out_b2, out_ld, R, C, nb2));
    return sf_check_launch("sf_transpose");
}

extern "C" int sf_cast_from_f32(const float* in, long ldin, void* out, int dtype, long ldout, long rows, int C, float scale,
                                void* stream) {
    SF_CHECK_ARG(rows >= 0 && C >= 0 && C % 8 == 0 && ldin % 8 == 0 && ldout % 8 == 0, "sf_cast_from_f32: bad shape");
    if (rows == 0 || C == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((cast_from_f32_kernel<T>), dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, in, ldin,
                                   (T*)out, ldout, rows, C, scale));
    return sf_check_launch("sf_cast_from_f32");
}

namespace {
// dst[b*Spad + s + off][c] += src[b*S + s][c]   (bf16 -> fp32 running sum; 8 columns per thread)
SF_GLOBAL void shift_accum_kernel(const sf_bf16* src, long ldsrc, float* dst, long lddst, long rows, int S, int Spad,
                                  int off, int C8) {
    const long total = rows * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        const long b = r / S;
        const long dr = b * Spad + (r - b * S) + off;
        float x[8], y[8];
        SfVec8<sf_bf16>::ld(src + r * ldsrc + c, x);
        SfVec8<float>::ld(dst + dr * lddst + c, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] += x[j];
        SfVec8<float>::st(dst + dr * lddst + c, y);
    }
}
// hi = bf16(x), lo = bf16(x - hi): a two-term bf16 expansion (16 mantissa bits) of an fp32 matrix
SF_GLOBAL void split_bf16_kernel(const float* in, long ldin, sf_bf16* hi, sf_bf16* lo, long ldout, long rows, int C8) {
    const long total = rows * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        float x[8], h[8], l[8];
        SfVec8<float>::ld(in + r * ldin + c, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = sf_round_bf(x[j]);
            l[j] = x[j] - h[j];
        }
        SfVec8<sf_bf16>::st(hi + r * ldout + c, h);
        SfVec8<sf_bf16>::st(lo + r * ldout + c, l);
    }
}
// hi | lo [b*Spad + p][c] = two-term bf16 expansion of  sum_{k = T-1 .. 0} src[k*N + b*S + (p - k)][c]  (terms with p - k outside
// 0 .. S-1 are absent): the T per-step shift-accumulate passes (an fp32 read-modify-write of the whole [B*Spad, C] sum each) and the
// split, in ONE pass that reads each stash row once.  The fp32 adds run in the order the per-step form ran them (k descending).
SF_GLOBAL void shift_sum_split_kernel(const sf_bf16* src, long ldsrc, long N, int T, sf_bf16* hi, sf_bf16* lo, long ldout, long rows_pad,
                                      int S, int Spad, int C8) {
    const long total = rows_pad * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        const long b = r / Spad;
        const int p = (int)(r - b * Spad);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = p < T - 1 ? p : T - 1;                 // s = p - k >= 0
        const int klo = p - (S - 1) > 0 ? p - (S - 1) : 0;   // s <= S - 1
        // the (at most T) rows are independent loads: batches of 4 in flight per lane
        for (; k - 3 >= klo; k -= 4) {
            SfRaw8<sf_bf16> x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u].ld(src + ((long)(k - u) * N + b * S + (p - (k - u))) * ldsrc + c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += x[u].at(j);
        }
        for (; k >= klo; --k) {
            SfRaw8<sf_bf16> x;
            x.ld(src + ((long)k * N + b * S + (p - k)) * ldsrc + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += x.at(j);
        }
        float h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = sf_round_bf(acc[j]);
            l[j] = acc[j] - h[j];
        }
        SfVec8<sf_bf16>::st(hi + r * ldout + c, h);
        SfVec8<sf_bf16>::st(lo + r * ldout + c, l);
    }
}
}  // namespace

extern "C" int sf_shift_sum_split(const void* src, long ldsrc, int T, int B, int S, int Spad, int C, void* hi, void* lo, long ldout,
                                  void* stream) {
    SF_CHECK_ARG(T >= 1 && B >= 0 && S > 0 && Spad >= S + T - 1, "sf_shift_sum_split: bad shape (Spad must hold S + T - 1 positions)");
    SF_CHECK_ARG(C >= 0 && C % 8 == 0 && ldsrc % 8 == 0 && ldout % 8 == 0, "sf_shift_sum_split: C and strides must be multiples of 8");
    const long rows_pad = (long)B * Spad;
    if (rows_pad == 0 || C == 0) return 0;
    SF_LAUNCH(shift_sum_split_kernel, dim3(grid_for(rows_pad * (C / 8))), dim3(256), 0, stream, (const sf_bf16*)src, ldsrc, (long)B * S, T,
              (sf_bf16*)hi, (sf_bf16*)lo, ldout, rows_pad, S, Spad, C / 8);
    return sf_check_launch("sf_shift_sum_split");
}

extern "C" int sf_shift_accum(const void* src, long ldsrc, float* dst, long lddst, int B, int S, int Spad, int off, int C,
                              void* stream) {
    SF_CHECK_ARG(B >= 0 && S > 0 && Spad >= S && off >= 0 && off + S <= Spad, "sf_shift_accum: bad shape");
    SF_CHECK_ARG(C >= 0 && C % 8 == 0 && ldsrc % 8 == 0 && lddst % 8 == 0, "sf_shift_accum: C and strides must be multiples of 8");
    const long rows = (long)B * S;
    if (rows == 0 || C == 0) return 0;
    SF_LAUNCH(shift_accum_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, (const sf_bf16*)src, ldsrc, dst, lddst,
              rows, S, Spad, off, C / 8);
    return sf_check_launch("sf_shift_accum");
}

extern "C" int sf_split_bf16(const float* in, long ldin, void* hi, void* lo, long ldout, long rows, int C, void* stream) {
    SF_CHECK_ARG(rows >= 0 && C >= 0 && C % 8 == 0 && ldin % 8 == 0 && ldout % 8 == 0, "sf_split_bf16: bad shape");
    if (rows == 0 || C == 0) return 0;
    SF_LAUNCH(split_bf16_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, in, ldin, (sf_bf16*)hi, (sf_bf16*)lo,
              ldout, rows, C / 8);
    return sf_check_launch("sf_split_bf16");
}

extern "C" long sf_grad_norm_workspace_floats(void) { return 1024; }

// norm_out[0] = prescale * sqrt(sum float(g)^2)   (prescale = 1/world for SUM-reduced grads)
extern "C" int sf_grad_norm(const void* g, int dtype, long n, float prescale, float* norm_out, float* workspace, void* stream) {
    SF_CHECK_ARG(n >= 0 && workspace && norm_out, "sf_grad_norm: bad args");
    const int nb = (int)((n / 8 + 255) / 256 < 1 ? 1 : ((n / 8 + 255) / 256 > 1024 ? 1024 : (n / 8 + 255) / 256));
    SF_DISPATCH_T(dtype, SF_LAUNCH((sumsq_kernel<T>), dim3(nb), dim3(256), 0, stream, (const T*)g, n, workspace));
    SF_LAUNCH(norm_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, nb, norm_out, prescale);
    return sf_check_launch("sf_grad_norm");
}

extern "C" int sf_adamw_step(const void* g, int dtype, float* master, float* m, float* v, void* param, long n,
                             const float* norm, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                             int step, float grad_prescale, void* stream) {
    SF_CHECK_ARG(n >= 0 && step >= 1, "sf_adamw_step: bad args");
    if (n == 0) return 0;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const int vec = (((size_t)g | (size_t)master | (size_t)m | (size_t)v | (size_t)param) & 15) == 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((adamw_kernel<T>), dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, stream, (const T*)g,
                                   master, m, v, (T*)param, n, norm, max_norm, lr, beta1, beta2, eps, wd, bc1, bc2s,
                                   grad_prescale, vec));
    return sf_check_launch("sf_adamw_step");
}

namespace {
// y = (accumulate ? y : 0) + alpha * x      (fp32, grid-stride)
SF_GLOBAL void axpy_f32_kernel(long n, float alpha, const float* x, float* y, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = (accumulate ? y[i] : 0.f) + alpha * x[i];
}
}  // namespace

namespace {
// out = a + b (bf16 operands, fp32 add, one rounding) -- the gradient sum at a residual fork
SF_GLOBAL void add_bf16_kernel(long n8, const sf_bf16* a, const sf_bf16* b, sf_bf16* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float x[8], y[8];
        SfVec8<sf_bf16>::ld(a + i * 8, x);
        SfVec8<sf_bf16>::ld(b + i * 8, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        SfVec8<sf_bf16>::st(out + i * 8, x);
    }
}
}  // namespace

extern "C" int sf_add_bf16(long n, const void* a, const void* b, void* out, void* stream) {
    SF_CHECK_ARG(n >= 0 && n % 8 == 0, "sf_add_bf16: n must be a non-negative multiple of 8");
    if (n == 0) return 0;
    SF_LAUNCH(add_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, n / 8, (const sf_bf16*)a, (const sf_bf16*)b,
              (sf_bf16*)out);
    return sf_check_launch("sf_add_bf16");
}

namespace {
// dst[r][:] = inv[r] >= 0 ? src[inv[r]][:] : 0   (16-byte pieces; the inverse of a row compaction: rows that were left out come back as zeros)
template <typename T>
SF_GLOBAL void rows_expand_kernel(const T* src, long lds, const int* inv, T* dst, long ldd, long rows, int C8) {
    const long total = rows * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C8;
        const int c = (int)(i - r * C8) * 8;
        const int j = inv[r];
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (j >= 0) SfVec8<T>::ld(src + (long)j * lds + c, v);        // (bf16 <-> fp32 widening is exact both ways)
        SfVec8<T>::st(dst + r * ldd + c, v);
    }
}
}  // namespace

// The inverse of a row compaction (round 4, loss-row compaction): dst [rows, C] gets src row inv[r] where inv[r] >= 0 and zeros elsewhere --
// one pass over dst instead of a fill plus an indexed copy.  C % 8 == 0, 16-byte aligned rows.
extern "C" int sf_rows_expand(const void* src, int dtype, long ld_src, const int* inv, void* dst, long ld_dst, long rows, int C, void* stream) {
    SF_CHECK_ARG(rows >= 0 && C >= 0 && C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && inv && src && dst, "sf_rows_expand: bad shape");
    SF_CHECK_ARG((((size_t)src | (size_t)dst) & 15) == 0, "sf_rows_expand: rows must be 16-byte aligned");
    if (rows == 0 || C == 0) return 0;
    SF_DISPATCH_T(dtype, SF_LAUNCH((rows_expand_kernel<T>), dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, (const T*)src, ld_src, inv,
                                   (T*)dst, ld_dst, rows, C / 8));
    return sf_check_launch("sf_rows_expand");
}

extern "C" int sf_axpy_f32(long n, float alpha, const float* x, float* y, int accumulate, void* stream) {
    SF_CHECK_ARG(n >= 0, "sf_axpy_f32: bad size");
    if (n == 0) return 0;
    SF_LAUNCH(axpy_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, alpha, x, y, accumulate);
    return sf_check_launch("sf_axpy_f32");
}
