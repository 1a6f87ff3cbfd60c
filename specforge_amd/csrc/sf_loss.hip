// Soft-target cross-entropy of one TTT step (forward + in-place backward + accuracy +
// acceptance rate in ONE launch) and the teacher reduction that builds its targets.
//
// Replaces, for the EAGLE3 offline path of the reference:
//   specforge/core/loss.py:49-228        LogSoftmaxLoss (Triton fwd/bwd, grad written in place)
//   specforge/algorithms/eagle3/model.py:161-173   argmax + d2t accuracy
//   specforge/core/lk_loss.py:43-80      acceptance rate  sum_v min(p_target_on_draft, softmax)
//   specforge/algorithms/eagle3/model.py:487-501   _compute_target_p (teacher softmax / lse / argmax)
//
// HBM-bound: one workgroup per token row.  Algorithmic traffic per row of the fused CE:
// logits 2*V (bf16, read; second read is an L2/MALL hit) + target_p 4*V (fp32, read once)
// + dlogits 2*V (bf16, written in place) = 8*V bytes.
#include "sf_api_internal.h"
#include "sf_util.h"
#include <type_traits>

namespace {

struct ArgMax {
    float v;
    int i;
};
SF_DEVICE ArgMax am_pick(ArgMax a, ArgMax b) {  // larger value; ties -> lower index (torch.argmax)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
SF_DEVICE ArgMax am_wave(ArgMax a) {
    for (int m = 32; m >= 1; m >>= 1) {
        ArgMax o;
        o.v = sf_shfl_xor(a.v, m);
        o.i = sf_shfl_xor(a.i, m);
        a = am_pick(a, o);
    }
    return a;
}
// online (max, sum-exp) pair merge
SF_DEVICE void md_merge(float& m, float& d, float m2, float d2) {
    float mn = fmaxf(m, m2);
    float a = (m == SF_NEG_BIG) ? 0.f : d * sf_exp(m - mn);
    float b = (m2 == SF_NEG_BIG) ? 0.f : d2 * sf_exp(m2 - mn);
    m = mn;
    d = a + b;
}

// ---------------------------------------------------------------- fused CE
// Row r = b*S + s reads its soft target / masks / ids at padded index b*Spad + s + off
// (the TTT shift of step `off`: specforge/algorithms/eagle3/model.py:364-433 slices
// target_p[:, idx:idx+S] and shifts ids/masks left with zero fill -- zero-padded tails
// make the shift an offset).
// ZT = 1: the soft target is not read as fp32 probabilities but formed on the fly from the teacher's stored draft logits,
// p_j = exp(zt[row'][j] - zmd) * zinv with row' = b*S + s + off (NATURAL rows: a row with a position mask has s + off < S) and
// (zmd, zinv) = the row's draft maximum and 1 / sum-exp from sf_teacher_reduce_perm -- the expression that kernel uses when it
// writes target_p, so both forms give the same bits; 2 bytes per element instead of 4, and [B, S, Vd] fp32 is never written.
template <typename T, int ZT = 0>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
ce_fused_kernel(T* logits, long ld, int V, const float* target, int S, int Spad, int off,
                const int* pos_mask_pad, const int* loss_mask_pad, const long long* tgt_ids_pad,
                const float* pod_scale_pad, const float* tsum_pad, const long long* d2t, float grad_scale,
                int write_grad, float* row_loss, float* row_correct, float* row_accept, int* row_pred,
                const sf_bf16* zt = nullptr, long ldzt = 0, const float* zmd_pad = nullptr, const float* zinv_pad = nullptr,
                const long long* row_map = nullptr) {
    SF_SHARED float red[32];
    SF_SHARED int redi[16];
    const int r = (int)blockIdx.x;
    // row_map (round 4: loss-row compaction): logits row r / its row_* outputs belong to token row row_map[r] of the [B, S] grid --
    // only the rows that carry a loss mask went through lm_head; the targets and masks are still addressed by token position
    const int rt = row_map ? (int)row_map[r] : r;
    const int b = rt / S, s = rt - b * S;
    const long pr = (long)b * Spad + s + off;
    T* x = logits + (long)r * ld;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const int V8 = V >> 3;

    // pass 1: online max / sum-exp / argmax over the draft vocabulary
    float m = SF_NEG_BIG, d = 0.f;
    ArgMax am{SF_NEG_BIG, 0x7fffffff};
    auto chunk1 = [&](int c, const SfRaw8<T>& raw) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = raw.at(i);
        float cm = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v[i]);
        if (cm > am.v) {   // rare after the first chunks; same winner as an unconditional scan (strict >, ascending i)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v[i] > am.v) { am.v = v[i]; am.i = c * 8 + i; }
        }
        float mn = fmaxf(m, cm);
        float acc = (m == SF_NEG_BIG) ? 0.f : d * sf_exp_fast(m - mn);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += sf_exp_fast(v[i] - mn);
        m = mn;
        d = acc;
    };
    // four 16-byte chunks in flight per lane (one per trip leaves the row latency-bound); folded in ascending order
    constexpr int PU = 4;
    {
        int c = tid;
        for (; c + (PU - 1) * nt < V8; c += PU * nt) {
            SfRaw8<T> raw[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) raw[u].ld(x + (c + u * nt) * 8);
#pragma unroll
            for (int u = 0; u < PU; ++u) chunk1(c + u * nt, raw[u]);
        }
        for (; c < V8; c += nt) {
            SfRaw8<T> raw;
            raw.ld(x + c * 8);
            chunk1(c, raw);
        }
    }
    for (int j = V8 * 8 + tid; j < V; j += nt) {  // tail (V % 8)
        float v = SfElem<T>::ld(x + j);
        if (v > am.v) { am.v = v; am.i = j; }
        md_merge(m, d, v, 1.f);
    }
    // workgroup merge
    for (int k = 32; k >= 1; k >>= 1) {
        float m2 = sf_shfl_xor(m, k), d2 = sf_shfl_xor(d, k);
        md_merge(m, d, m2, d2);
    }
    am = am_wave(am);
    const int w = tid >> 6, nw = nt >> 6;
    if (sf_lane() == 0) { red[w] = m; red[8 + w] = d; red[16 + w] = am.v; redi[w] = am.i; }
    sf_syncthreads();
    m = red[0]; d = red[8];
    am.v = red[16]; am.i = redi[0];
    for (int i = 1; i < nw; ++i) {
        md_merge(m, d, red[i], red[8 + i]);
        am = am_pick(am, ArgMax{red[16 + i], redi[i]});
    }
    const float lse = m + sf_log(d);
    const int pm = (ZT && s + off >= S) ? 0 : pos_mask_pad[pr];     // (ZT: padded positions have no teacher row; their mask is 0 anyway)

    float loss = 0.f, acc_min = 0.f;
    if (pm != 0) {
        const float* tp = ZT ? nullptr : target + pr * (long)V;
        const sf_bf16* zr = ZT ? zt + ((long)b * S + s + off) * ldzt : nullptr;
        const float zmd = ZT ? zmd_pad[pr] : 0.f, zinv = ZT ? zinv_pad[pr] : 0.f;
        const float podc = pod_scale_pad ? pod_scale_pad[pr] : 0.f;
        float tsum;
        if (tsum_pad) {
            tsum = tsum_pad[pr];
        } else {  // drop-in LogSoftmaxLoss: sum of the target row (core/loss.py:113-170 pass 1)   (never with ZT: its launcher requires tsum)
            float t = 0.f;
            for (int j = tid; j < V; j += nt) t += ZT ? 0.f : tp[j];
            tsum = sf_block_sum(t, red);
            sf_syncthreads();
        }
        const float gs = grad_scale * (float)pm;
        using PRaw = std::conditional_t<ZT != 0, SfRaw8<sf_bf16>, SfRaw8<float>>;
        auto pld = [&](PRaw& pr8, int c) {
            if constexpr (ZT) pr8.ld(zr + c * 8);
            else pr8.ld(tp + c * 8);
        };
        auto chunk2 = [&](int c, const SfRaw8<T>& raw, const PRaw& praw) {
            float g[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float pi = ZT ? sf_exp_fast(praw.at(i) - zmd) * zinv : praw.at(i);
                float lp = raw.at(i) - lse;
                float sm = sf_exp_fast(lp);
                loss -= pi * lp;
                acc_min += fminf(pi * podc, sm);
                g[i] = (sm * tsum - pi) * gs;
            }
            if (write_grad) SfVec8<T>::st(x + c * 8, g);
        };
        int c = tid;
        for (; c + (PU - 1) * nt < V8; c += PU * nt) {
            SfRaw8<T> raw[PU];
            PRaw praw[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                raw[u].ld(x + (c + u * nt) * 8);
                pld(praw[u], c + u * nt);
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) chunk2(c + u * nt, raw[u], praw[u]);
        }
        for (; c < V8; c += nt) {
            SfRaw8<T> raw;
            PRaw praw;
            raw.ld(x + c * 8);
            pld(praw, c);
            chunk2(c, raw, praw);
        }
        for (int j = V8 * 8 + tid; j < V; j += nt) {
            float v = SfElem<T>::ld(x + j), p = ZT ? sf_exp_fast(sf_bf2f(zr[j]) - zmd) * zinv : tp[j];
            float lp = v - lse, sm = sf_exp_fast(lp);
            loss -= p * lp;
            acc_min += fminf(p * podc, sm);
            if (write_grad) SfElem<T>::st(x + j, (sm * tsum - p) * gs);
        }
        loss = sf_block_sum(loss, red) * (float)pm;
        acc_min = sf_block_sum(acc_min, red) * (float)pm;
    } else if (write_grad) {  // masked row: zero gradient (core/loss.py:160-170)
        float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = tid; c < V8; c += nt) SfVec8<T>::st(x + c * 8, z);
        for (int j = V8 * 8 + tid; j < V; j += nt) SfElem<T>::st(x + j, 0.f);
    }
    if (tid == 0) {
        const int lm = loss_mask_pad[pr];
        long long pred_t = (long long)am.i + (d2t ? d2t[am.i] : 0);
        row_loss[r] = loss;
        row_accept[r] = acc_min;
        row_correct[r] = (tgt_ids_pad && pred_t == tgt_ids_pad[pr]) ? (float)lm : 0.f;
        if (row_pred) row_pred[r] = am.i;
    }
}


// ---------------------------------------------------------------- LK-loss gradient
// d(step loss)/d(logits) for the LK objectives (specforge/core/lk_loss.py:83-99, eagle3/model.py:78-96),
// written in place of the logits like the CE gradient.  Per row r (position mask m_r in {0,1}):
//   s = softmax(x_r), q = target_p * pod_scale (= target_p_on_draft), a_r = sum_v min(q_v, s_v),
//   ind_v = d min(q_v, s_v)/d s_v = [s_v < q_v] (1/2 on exact ties, torch.minimum's rule), c_r = sum_v ind_v s_v
//   d a_r / d x_j = s_j (ind_j - c_r)                                      (softmax Jacobian)
//   D = max(sum_r m_r, 1e-8), alpha = sum_r m_r a_r / D                    (masked means, lk_loss.py:21-38)
//   "alpha":  loss = -sum_r m_r log a_r / D          -> g_rj = -step * m_r / (D a_r) * s_j (ind_j - c_r)
//   "lambda": w = kl_scale exp(-kl_decay alpha) (detached); loss = w KL + (1 - w)(1 - alpha)
//             -> g_rj = step * m_r * [ w * kl_row_scale * (s_j tsum_r - p_j) - (1 - w) / D * s_j (ind_j - c_r) ]
// accept_sum = sum_r m_r a_r and mask_sum = sum_r m_r are device scalars produced by sf_ce_fused +
// sf_reduce_sum (the global reductions must finish before any gradient can be scaled).
template <typename T>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
ce_lk_grad_kernel(T* logits, long ld, int V, const float* target, int S, int Spad, int off, const int* pos_mask_pad,
                  const float* pod_scale_pad, const float* tsum_pad, int lk_mode, float kl_scale, float kl_decay,
                  float step_scale, float kl_row_scale, const float* accept_sum, const float* mask_sum) {
    SF_SHARED float red[32];
    const int r = (int)blockIdx.x;
    const int b = r / S, s = r - b * S;
    const long pr = (long)b * Spad + s + off;
    T* x = logits + (long)r * ld;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const int pm = pos_mask_pad[pr];
    if (pm == 0) {
        for (int j = tid; j < V; j += nt) SfElem<T>::st(x + j, 0.f);
        return;
    }
    // pass 1: log-sum-exp
    float m = SF_NEG_BIG;
    for (int j = tid; j < V; j += nt) m = fmaxf(m, SfElem<T>::ld(x + j));
    m = sf_block_max(m, red);
    sf_syncthreads();
    float d = 0.f;
    for (int j = tid; j < V; j += nt) d += sf_exp(SfElem<T>::ld(x + j) - m);
    d = sf_block_sum(d, red);
    sf_syncthreads();
    const float lse = m + sf_log(d);
    // pass 2: a_r and c_r
    const float* tp = target + pr * (long)V;
    const float podc = pod_scale_pad[pr];
    float a = 0.f, c = 0.f;
    for (int j = tid; j < V; j += nt) {
        const float sm = sf_exp(SfElem<T>::ld(x + j) - lse), q = tp[j] * podc;
        a += fminf(q, sm);
        c += sm < q ? sm : (sm == q ? 0.5f * sm : 0.f);
    }
    a = sf_block_sum(a, red);
    sf_syncthreads();
    c = sf_block_sum(c, red);
    sf_syncthreads();
    const float D = fmaxf(mask_sum[0], 1e-8f);
    float ka, kb;  // g = ka * (s tsum - p) + kb * s (ind - c)
    if (lk_mode == 1) {
        ka = 0.f;
        kb = a > 0.f ? -step_scale / (D * a) : 0.f;
    } else {
        const float alpha = accept_sum[0] / D;
        const float w = kl_scale * sf_exp(-kl_decay * alpha);
        ka = step_scale * kl_row_scale * w;
        kb = -step_scale * (1.f - w) / D;
    }
    float tsum = 0.f;
    if (ka != 0.f) {
        if (tsum_pad) tsum = tsum_pad[pr];
        else {
            float t = 0.f;
            for (int j = tid; j < V; j += nt) t += tp[j];
            tsum = sf_block_sum(t, red);
            sf_syncthreads();
        }
    }
    // pass 3: gradient in place
    const float pmf = (float)pm;
    for (int j = tid; j < V; j += nt) {
        const float p = tp[j];
        const float sm = sf_exp(SfElem<T>::ld(x + j) - lse), q = p * podc;
        const float ind = sm < q ? 1.f : (sm == q ? 0.5f : 0.f);
        SfElem<T>::st(x + j, pmf * (ka * (sm * tsum - p) + kb * sm * (ind - c)));
    }
}

// deterministic sum of n floats (fixed order, double accumulation): workgroup b sums segment b, out[b] = sum
SF_GLOBAL void reduce_sum_kernel(const float* in_all, long n, float* out, float scale) {
    SF_SHARED double part[256];
    const int tid = (int)threadIdx.x;
    const float* in = in_all + (long)blockIdx.x * n;
    double acc = 0.0;
    for (long i = tid; i < n; i += 256) acc += (double)in[i];
    part[tid] = acc;
    sf_syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (tid < sft) part[tid] += part[tid + sft];
        sf_syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = (float)(part[0] * (double)scale);
}

// The per-TTT-step metric scalars of Eagle3TrainStrategy.forward_loss in ONE launch (they were ~40 tiny ATen kernels per step):
// workgroup k counts the loss / position mask of step k (the masks shifted by k: offsets into the zero-padded [B, S+T] arrays) and
// forms  ploss = row_loss_sum / (B S),  acc = correct / max(count_lm, 1e-6),  acceptance = accept_sum / max(count_pm, 1e-8)  with the
// same fp32 operations torch used (counts of 0/1 are exact in fp32; one IEEE division each)   (eagle3/model.py:161-190)
SF_GLOBAL void eagle3_metrics_kernel(const float* met, const int* lm_pad, const int* pm_pad, int B, int S, int Spad, float* out) {
    SF_SHARED int cnt[2][256];
    const int k = (int)blockIdx.x, tid = (int)threadIdx.x;
    int nl = 0, np = 0;
    for (long i = tid; i < (long)B * S; i += 256) {
        const long b = i / S, idx = b * Spad + (i - b * S) + k;
        nl += lm_pad[idx] != 0;
        np += pm_pad[idx] != 0;
    }
    cnt[0][tid] = nl;
    cnt[1][tid] = np;
    sf_syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (tid < sft) { cnt[0][tid] += cnt[0][tid + sft]; cnt[1][tid] += cnt[1][tid + sft]; }
        sf_syncthreads();
    }
    if (tid == 0) {
        const float n = (float)((long)B * S);
        const float denom = fmaxf((float)cnt[0][0], 1e-6f), pden = fmaxf((float)cnt[1][0], 1e-8f);
        float* o = out + (long)k * 8;
        o[0] = met[k * 3 + 0] / n;          // ploss
        o[1] = met[k * 3 + 1];              // acc_correct
        o[2] = denom;                       // acc_denom
        o[3] = met[k * 3 + 1] / denom;      // acc
        o[4] = met[k * 3 + 2] / pden;       // acceptance rate
        o[5] = pden;
        o[6] = n;                           // metric_loss_denom
        o[7] = o[0];                        // metric_loss (its own element: the caller hands out views)
    }
}

// ------------------------------------------------------------ teacher reduce
// One row of teacher logits z[Vt] -> argmax id, position mask, softmax over the draft
// sub-vocabulary (target_p), and pod_scale = sum_d exp(z_d - logsumexp(z)) / so that
// target_p_on_draft = target_p * pod_scale (eagle3/model.py:487-501).
template <typename T>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
teacher_reduce_kernel(const T* z, long ldz, int Vt, int Vd, const long long* d2t, const unsigned char* t2d,
                      const int* loss_mask_pad, int S, int Spad, float* target_p_pad, float* pod_scale_pad,
                      float* tsum_pad, long long* ids_pad, int* pos_mask_pad) {
    SF_SHARED float red[32];
    SF_SHARED int redi[16];
    const int r = (int)blockIdx.x;
    const int b = r / S, s = r - b * S;
    const long pr = (long)b * Spad + s;
    const T* x = z + (long)r * ldz;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const int V8 = Vt >> 3;
    // (m, d): online max / sum-exp over the full vocabulary; (md, sdd): the same over the draft sub-vocabulary (t2d
    // mask), so that the gathered pass below can write normalised probabilities in ONE pass
    float m = SF_NEG_BIG, d = 0.f, md = SF_NEG_BIG, sdd = 0.f;
    ArgMax am{SF_NEG_BIG, 0x7fffffff};
    auto chunk = [&](int c, const SfRaw8<T>& raw, unsigned long long mk) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = raw.at(i);
        if (mk) {
            float cmd = SF_NEG_BIG;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((mk >> (8 * i)) & 0xffull) cmd = fmaxf(cmd, v[i]);
            const float mdn = fmaxf(md, cmd);
            float accd = (md == SF_NEG_BIG) ? 0.f : sdd * sf_exp_fast(md - mdn);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((mk >> (8 * i)) & 0xffull) accd += sf_exp_fast(v[i] - mdn);
            md = mdn;
            sdd = accd;
        }
        float cm = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v[i]);
        if (cm > am.v) {   // rare after the first chunks; same winner as an unconditional scan (strict >, ascending i)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v[i] > am.v) { am.v = v[i]; am.i = c * 8 + i; }
        }
        float mn = fmaxf(m, cm);
        float acc = (m == SF_NEG_BIG) ? 0.f : d * sf_exp_fast(m - mn);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += sf_exp_fast(v[i] - mn);
        m = mn;
        d = acc;
    };
    // four chunks (and their t2d mask words) in flight per lane; chunks are folded in ascending order, as one at a time did
    constexpr int PU = 4;
    int c = tid;
    for (; c + (PU - 1) * nt < V8; c += PU * nt) {
        SfRaw8<T> raw[PU];
        unsigned long long mk[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            raw[u].ld(x + (c + u * nt) * 8);
            mk[u] = *reinterpret_cast<const unsigned long long*>(t2d + (c + u * nt) * 8);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) chunk(c + u * nt, raw[u], mk[u]);
    }
    for (; c < V8; c += nt) {
        SfRaw8<T> raw;
        raw.ld(x + c * 8);
        chunk(c, raw, *reinterpret_cast<const unsigned long long*>(t2d + c * 8));
    }
    for (int j = V8 * 8 + tid; j < Vt; j += nt) {
        float v = SfElem<T>::ld(x + j);
        if (v > am.v) { am.v = v; am.i = j; }
        if (t2d[j]) md_merge(md, sdd, v, 1.f);
        md_merge(m, d, v, 1.f);
    }
    for (int k = 32; k >= 1; k >>= 1) {
        float m2 = sf_shfl_xor(m, k), d2 = sf_shfl_xor(d, k);
        md_merge(m, d, m2, d2);
        float m3 = sf_shfl_xor(md, k), d3 = sf_shfl_xor(sdd, k);
        md_merge(md, sdd, m3, d3);
    }
    am = am_wave(am);
    const int w = tid >> 6, nw = nt >> 6;
    if (sf_lane() == 0) { red[w] = m; red[8 + w] = d; red[16 + w] = am.v; redi[w] = am.i; red[24 + w] = md; }
    sf_syncthreads();
    m = red[0]; d = red[8];
    am.v = red[16]; am.i = redi[0];
    float mdb = red[24];
    for (int i = 1; i < nw; ++i) {
        md_merge(m, d, red[i], red[8 + i]);
        am = am_pick(am, ArgMax{red[16 + i], redi[i]});
        mdb = fmaxf(mdb, red[24 + i]);
    }
    const float lse_full = m + sf_log(d);
    sf_syncthreads();
    // block-wide draft sum-exp relative to the block max (second small exchange through LDS)
    sdd = (md == SF_NEG_BIG) ? 0.f : sdd * sf_exp(md - mdb);   // this wave's partial, rescaled (lane-uniform after the butterfly)
    if (sf_lane() == 0) red[w] = sdd;
    sf_syncthreads();
    float sd = 0.f;
    for (int i = 0; i < nw; ++i) sd += red[i];
    sf_syncthreads();
    md = mdb;
    const float inv = 1.0f / sd;
    // draft sub-vocabulary: one gathered pass writes the normalised probabilities (torch.softmax: exp(x - max) / sum)
    float* tp = target_p_pad + pr * (long)Vd;
    float ts = 0.f;
    // the gather is two dependent loads per element (d2t[j], then the logit): batches of 8 keep 8 of each in flight per lane
    // (one element per trip left the launch latency-bound at 1.2 TB/s); the per-lane summation order is unchanged
    constexpr int GU = 8;
    int j = tid;
    for (; j + (GU - 1) * nt < Vd; j += GU * nt) {
        long long o[GU];
        float v[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) o[u] = d2t[j + u * nt];
#pragma unroll
        for (int u = 0; u < GU; ++u) v[u] = SfElem<T>::ld(x + j + u * nt + o[u]);
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const float p = sf_exp_fast(v[u] - md) * inv;
            tp[j + u * nt] = p;
            ts += p;
        }
    }
    for (; j < Vd; j += nt) {
        const float p = sf_exp_fast(SfElem<T>::ld(x + j + d2t[j]) - md) * inv;
        tp[j] = p;
        ts += p;
    }
    ts = sf_block_sum(ts, red);
    if (tid == 0) {
        pod_scale_pad[pr] = sd * sf_exp(md - lse_full);
        tsum_pad[pr] = ts;
        ids_pad[pr] = (long long)am.i;
        pos_mask_pad[pr] = (t2d[am.i] ? 1 : 0) * loss_mask_pad[pr];
    }
}

// The same reduction over a row whose columns are PERMUTED so that the draft sub-vocabulary comes first, in draft order:
// column c of z is target-vocabulary entry perm[c]; perm[j] = j + d2t[j] for j < Vd, the remaining entries follow in ascending
// order (the host permutes the rows of the frozen teacher head once, so the head GEMM writes this layout for free).  What it
// buys: the draft softmax reads Vd CONTIGUOUS logits (the natural layout gathers them through d2t: two dependent loads per
// element over a 256 KiB row, which kept the launch at 2.8 TB/s), and the streaming pass needs no t2d mask words.  Same results:
// the argmax is taken in original indices (lowest original index among equal maxima, as torch.argmax).
// `part` (optional): the row's columns from Vz on are not in z -- they arrive as per-block partials written by the head GEMM's
// reduction epilogue (sf_gemm_nt_teacher): part[r * part_stride + blk] = {max, sum exp(z - max), argmax column (permuted), 0}.
template <typename T>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2)
teacher_reduce_perm_kernel(const T* z, long ldz, int Vz, int Vd, const int* perm, const unsigned char* t2d, const sf_v4f* part,
                           int nparts, long part_stride, const int* loss_mask_pad, int S, int Spad, float* target_p_pad,
                           float* pod_scale_pad, float* tsum_pad, long long* ids_pad, int* pos_mask_pad, float* zmd_pad,
                           float* zinv_pad) {
    SF_SHARED float red[40];
    SF_SHARED int redi[16];
    const int r = (int)blockIdx.x;
    const int b = r / S, s = r - b * S;
    const long pr = (long)b * Spad + s;
    const T* x = z + (long)r * ldz;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const int V8 = Vz >> 3, D8 = Vd >> 3;
    // (m, d): online max / sum-exp over the row; dd: the sum-exp of the DRAFT columns relative to the same running maximum (the
    // terms are shared, so the draft pair costs no exponentials); md: the draft maximum (max only) -- the softmax reference
    float m = SF_NEG_BIG, d = 0.f, dd = 0.f, md = SF_NEG_BIG;
    ArgMax am{SF_NEG_BIG, 0x7fffffff};
    auto chunk = [&](int c, const SfRaw8<T>& raw) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = raw.at(i);
        float cm = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v[i]);
        if (cm >= am.v) {   // rare after the first chunks: a new maximum, or a tie to be settled by the original index
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v[i] == cm) {
                    const int o = perm[c * 8 + i];
                    if (cm > am.v || o < am.i) { am.v = cm; am.i = o; }
                }
        }
        if (cm > m) {       // (rare too) the running maximum moves: rescale both sums
            const float f = (m == SF_NEG_BIG) ? 0.f : sf_exp_fast(m - cm);
            d *= f;
            dd *= f;
            m = cm;
        }
        float s8 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s8 += sf_exp_fast(v[i] - m);
        d += s8;
        if (c < D8) {
            dd += s8;
            md = fmaxf(md, cm);
        }
    };
    constexpr int PU = 4;
    int c = tid;
    for (; c + (PU - 1) * nt < V8; c += PU * nt) {
        SfRaw8<T> raw[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) raw[u].ld(x + (c + u * nt) * 8);
#pragma unroll
        for (int u = 0; u < PU; ++u) chunk(c + u * nt, raw[u]);
    }
    for (; c < V8; c += nt) {
        SfRaw8<T> raw;
        raw.ld(x + c * 8);
        chunk(c, raw);
    }
    auto merge3 = [&](float m2, float d2, float dd2) {   // (m, d, dd) <- merged with another partial on the common maximum
        const float mn = fmaxf(m, m2);
        const float fa = (m == SF_NEG_BIG) ? 0.f : sf_exp(m - mn), fb = (m2 == SF_NEG_BIG) ? 0.f : sf_exp(m2 - mn);
        d = d * fa + d2 * fb;
        dd = dd * fa + dd2 * fb;
        m = mn;
    };
    for (int j = V8 * 8 + tid; j < Vz; j += nt) {   // (Vz % 8 != 0: never a draft column, Vd % 8 == 0 <= V8 * 8)
        const float v = SfElem<T>::ld(x + j);
        const int o = perm[j];
        if (v > am.v || (v == am.v && o < am.i)) { am.v = v; am.i = o; }
        merge3(v, 1.f, 0.f);
    }
    for (int q = tid; q < nparts; q += nt) {        // column blocks that exist only as the head GEMM's partials
        const sf_v4f pp = part[(long)r * part_stride + q];
        const int o = perm[(int)pp[2]];
        if (pp[0] > am.v || (pp[0] == am.v && o < am.i)) { am.v = pp[0]; am.i = o; }
        merge3(pp[0], pp[1], 0.f);
    }
    for (int k = 32; k >= 1; k >>= 1) {
        const float m2 = sf_shfl_xor(m, k), d2 = sf_shfl_xor(d, k), dd2 = sf_shfl_xor(dd, k);
        merge3(m2, d2, dd2);
        md = fmaxf(md, sf_shfl_xor(md, k));
    }
    am = am_wave(am);
    const int w = tid >> 6, nw = nt >> 6;
    if (sf_lane() == 0) { red[w] = m; red[8 + w] = d; red[16 + w] = am.v; redi[w] = am.i; red[24 + w] = md; red[32 + w] = dd; }
    sf_syncthreads();
    m = red[0]; d = red[8]; dd = red[32];
    am.v = red[16]; am.i = redi[0];
    md = red[24];
    for (int i = 1; i < nw; ++i) {
        merge3(red[i], red[8 + i], red[32 + i]);
        am = am_pick(am, ArgMax{red[16 + i], redi[i]});
        md = fmaxf(md, red[24 + i]);
    }
    const float lse_full = m + sf_log(d);
    sf_syncthreads();
    // sum exp(z_draft - md) = dd * exp(m - md); when the draft maximum lies so far below the row's that the shared terms lost
    // their precision (never with real logits), the sum is taken again relative to md (block-uniform branch)
    float sd;
    if (m - md < 60.f) {
        sd = dd * sf_exp(m - md);
    } else {
        float t = 0.f;
        for (int j = tid; j < D8; j += nt) {
            SfRaw8<T> raw;
            raw.ld(x + j * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) t += sf_exp_fast(raw.at(i) - md);
        }
        sd = sf_block_sum(t, red);
        sf_syncthreads();
    }
    const float inv = 1.0f / sd;
    // draft softmax (torch.softmax: exp(x - max) / sum): Vd contiguous logits, still in L2 from the pass above
    // (target_p_pad == NULL: the probabilities are not materialised -- sf_ce_fused_zt re-forms them from the stored logits with
    // (md, inv) below; their sum is still taken here, over the same values in the same per-lane order)
    float* tp = target_p_pad ? target_p_pad + pr * (long)Vd : nullptr;
    float ts = 0.f;
    constexpr int GU = 2;
    int j = tid;
    if (!tp) {
        // Nothing to write: the only other product of the pass below is the probabilities' sum, which is sd * inv = 1 up to one rounding
        // (round 5 re-summed the Vd exponentials here only to keep `tsum` bit-identical with the materialised form -- half of this
        // VALU-bound kernel's exponentials for the last ulp of a number the reference itself only knows to fp32 summation order)
        j = D8;
        ts = (tid == 0) ? sd * inv : 0.f;
    }
    for (; j + (GU - 1) * nt < D8; j += GU * nt) {
        SfRaw8<T> raw[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) raw[u].ld(x + (j + u * nt) * 8);
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { pv[i] = sf_exp_fast(raw[u].at(i) - md) * inv; ts += pv[i]; }
            if (tp) SfVec8<float>::st(tp + (j + u * nt) * 8, pv);
        }
    }
    for (; j < D8; j += nt) {
        SfRaw8<T> raw;
        raw.ld(x + j * 8);
        float pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { pv[i] = sf_exp_fast(raw.at(i) - md) * inv; ts += pv[i]; }
        if (tp) SfVec8<float>::st(tp + j * 8, pv);
    }
    ts = sf_block_sum(ts, red);
    if (tid == 0) {
        if (zmd_pad) { zmd_pad[pr] = md; zinv_pad[pr] = inv; }
        pod_scale_pad[pr] = sd * sf_exp(md - lse_full);
        tsum_pad[pr] = ts;
        ids_pad[pr] = (long long)am.i;
        pos_mask_pad[pr] = (t2d[am.i] ? 1 : 0) * loss_mask_pad[pr];
    }
}

}  // namespace

extern "C" int sf_ce_fused(void* logits, int dtype, long ld, int rows, int V, const float* target, int S, int Spad,
                           int off, const int* pos_mask_pad, const int* loss_mask_pad, const long long* tgt_ids_pad,
                           const float* pod_scale_pad, const float* tsum_pad, const long long* d2t, float grad_scale,
                           int write_grad, float* row_loss, float* row_correct, float* row_accept, int* row_pred,
                           const long long* row_map, void* stream) {
    SF_CHECK_ARG(rows >= 0 && V > 0 && S > 0 && Spad >= S && off >= 0 && off + S <= Spad, "sf_ce_fused: bad shape");
    SF_CHECK_ARG(dtype == SF_BF16 || dtype == SF_F32, "sf_ce_fused: dtype");
    SF_CHECK_ARG(ld % 8 == 0 && V % 8 == 0, "sf_ce_fused: ld and V must be multiples of 8");
    if (rows == 0) return 0;
    if (dtype == SF_BF16)
        SF_LAUNCH((ce_fused_kernel<sf_bf16>), dim3(rows), dim3(256), 0, stream, (sf_bf16*)logits, ld, V, target, S, Spad,
                  off, pos_mask_pad, loss_mask_pad, tgt_ids_pad, pod_scale_pad, tsum_pad, d2t, grad_scale, write_grad,
                  row_loss, row_correct, row_accept, row_pred, (const sf_bf16*)nullptr, 0L, (const float*)nullptr, (const float*)nullptr, row_map);
    else
        SF_LAUNCH((ce_fused_kernel<float>), dim3(rows), dim3(256), 0, stream, (float*)logits, ld, V, target, S, Spad,
                  off, pos_mask_pad, loss_mask_pad, tgt_ids_pad, pod_scale_pad, tsum_pad, d2t, grad_scale, write_grad,
                  row_loss, row_correct, row_accept, row_pred, (const sf_bf16*)nullptr, 0L, (const float*)nullptr, (const float*)nullptr, row_map);
    return sf_check_launch("sf_ce_fused");
}


extern "C" int sf_ce_lk_grad(void* logits, int dtype, long ld, int rows, int V, const float* target, int S, int Spad,
                             int off, const int* pos_mask_pad, const float* pod_scale_pad, const float* tsum_pad,
                             int lk_mode, float kl_scale, float kl_decay, float step_scale, float kl_row_scale,
                             const float* accept_sum, const float* mask_sum, void* stream) {
    SF_CHECK_ARG(rows >= 0 && V > 0 && S > 0 && Spad >= S && off >= 0 && off + S <= Spad, "sf_ce_lk_grad: bad shape");
    SF_CHECK_ARG(dtype == SF_BF16 || dtype == SF_F32, "sf_ce_lk_grad: dtype");
    SF_CHECK_ARG(lk_mode == 1 || lk_mode == 2, "sf_ce_lk_grad: lk_mode must be 1 (alpha) or 2 (lambda)");
    SF_CHECK_ARG(pos_mask_pad && pod_scale_pad && accept_sum && mask_sum, "sf_ce_lk_grad: missing input");
    if (rows == 0) return 0;
    if (dtype == SF_BF16)
        SF_LAUNCH((ce_lk_grad_kernel<sf_bf16>), dim3(rows), dim3(256), 0, stream, (sf_bf16*)logits, ld, V, target, S,
                  Spad, off, pos_mask_pad, pod_scale_pad, tsum_pad, lk_mode, kl_scale, kl_decay, step_scale,
                  kl_row_scale, accept_sum, mask_sum);
    else
        SF_LAUNCH((ce_lk_grad_kernel<float>), dim3(rows), dim3(256), 0, stream, (float*)logits, ld, V, target, S, Spad,
                  off, pos_mask_pad, pod_scale_pad, tsum_pad, lk_mode, kl_scale, kl_decay, step_scale, kl_row_scale,
                  accept_sum, mask_sum);
    return sf_check_launch("sf_ce_lk_grad");
}

extern "C" int sf_reduce_sum(const float* in, long n, int nsegments, float* out, float scale, void* stream) {
    SF_CHECK_ARG(n >= 0 && nsegments >= 1, "sf_reduce_sum: bad shape");
    // segment i sums in[i*n : (i+1)*n]
    SF_LAUNCH(reduce_sum_kernel, dim3((unsigned)nsegments), dim3(256), 0, stream, in, n, out, scale);
    return sf_check_launch("sf_reduce_sum");
}

extern "C" int sf_teacher_reduce(const void* z, int dtype, long ldz, int rows, int Vt, int Vd, const long long* d2t,
                                 const unsigned char* t2d, const int* loss_mask_pad, int S, int Spad,
                                 float* target_p_pad, float* pod_scale_pad, float* tsum_pad, long long* ids_pad,
                                 int* pos_mask_pad, void* stream) {
    SF_CHECK_ARG(rows >= 0 && Vt > 0 && Vd > 0 && S > 0 && Spad >= S, "sf_teacher_reduce: bad shape");
    SF_CHECK_ARG(ldz % 8 == 0, "sf_teacher_reduce: ldz must be a multiple of 8");
    if (rows == 0) return 0;
    if (dtype == SF_BF16)
        SF_LAUNCH((teacher_reduce_kernel<sf_bf16>), dim3(rows), dim3(256), 0, stream, (const sf_bf16*)z, ldz, Vt, Vd,
                  d2t, t2d, loss_mask_pad, S, Spad, target_p_pad, pod_scale_pad, tsum_pad, ids_pad, pos_mask_pad);
    else if (dtype == SF_F32)
        SF_LAUNCH((teacher_reduce_kernel<float>), dim3(rows), dim3(256), 0, stream, (const float*)z, ldz, Vt, Vd, d2t,
                  t2d, loss_mask_pad, S, Spad, target_p_pad, pod_scale_pad, tsum_pad, ids_pad, pos_mask_pad);
    else
        SF_CHECK_ARG(false, "sf_teacher_reduce: dtype");
    return sf_check_launch("sf_teacher_reduce");
}

extern "C" int sf_teacher_reduce_perm(const void* z, int dtype, long ldz, int rows, int Vz, int Vt, int Vd, const int* perm,
                                      const unsigned char* t2d, const float* part, int nparts, long part_stride,
                                      const int* loss_mask_pad, int S, int Spad, float* target_p_pad, float* pod_scale_pad,
                                      float* tsum_pad, long long* ids_pad, int* pos_mask_pad, float* zmd_pad, float* zinv_pad,
                                      void* stream) {
    SF_CHECK_ARG((zmd_pad != nullptr) == (zinv_pad != nullptr) && (target_p_pad || zmd_pad),
                 "sf_teacher_reduce_perm: zmd_pad / zinv_pad come together; without target_p_pad they are required");
    SF_CHECK_ARG(rows >= 0 && Vt > 0 && Vd > 0 && Vd % 8 == 0 && Vz >= Vd && Vz <= Vt && S > 0 && Spad >= S, "sf_teacher_reduce_perm: bad shape");
    SF_CHECK_ARG(ldz % 8 == 0 && perm && t2d, "sf_teacher_reduce_perm: ldz must be a multiple of 8; perm / t2d required");
    SF_CHECK_ARG(nparts >= 0 && (nparts == 0 || (part && part_stride >= nparts && ((size_t)part & 15) == 0)), "sf_teacher_reduce_perm: bad partials");
    SF_CHECK_ARG(nparts > 0 || Vz == Vt, "sf_teacher_reduce_perm: columns past Vz need partials");
    if (rows == 0) return 0;
    if (dtype == SF_BF16)
        SF_LAUNCH((teacher_reduce_perm_kernel<sf_bf16>), dim3(rows), dim3(256), 0, stream, (const sf_bf16*)z, ldz, Vz, Vd, perm, t2d,
                  (const sf_v4f*)part, nparts, part_stride, loss_mask_pad, S, Spad, target_p_pad, pod_scale_pad, tsum_pad, ids_pad, pos_mask_pad,
                  zmd_pad, zinv_pad);
    else if (dtype == SF_F32)
        SF_LAUNCH((teacher_reduce_perm_kernel<float>), dim3(rows), dim3(256), 0, stream, (const float*)z, ldz, Vz, Vd, perm, t2d,
                  (const sf_v4f*)part, nparts, part_stride, loss_mask_pad, S, Spad, target_p_pad, pod_scale_pad, tsum_pad, ids_pad, pos_mask_pad,
                  zmd_pad, zinv_pad);
    else
        SF_CHECK_ARG(false, "sf_teacher_reduce_perm: dtype");
    return sf_check_launch("sf_teacher_reduce_perm");
}

extern "C" int sf_eagle3_metrics(const float* met, const int* loss_mask_pad, const int* pos_mask_pad, int B, int S, int Spad, int T,
                                 float* out, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && T >= 1 && Spad >= S + T - 1 && met && loss_mask_pad && pos_mask_pad && out, "sf_eagle3_metrics: bad args");
    SF_LAUNCH(eagle3_metrics_kernel, dim3((unsigned)T), dim3(256), 0, stream, met, loss_mask_pad, pos_mask_pad, B, S, Spad, out);
    return sf_check_launch("sf_eagle3_metrics");
}

extern "C" int sf_ce_fused_zt(void* logits, int dtype, long ld, int rows, int V, const void* zt, long ldzt, const float* zmd_pad,
                              const float* zinv_pad, int S, int Spad, int off, const int* pos_mask_pad, const int* loss_mask_pad,
                              const long long* tgt_ids_pad, const float* pod_scale_pad, const float* tsum_pad, const long long* d2t,
                              float grad_scale, int write_grad, float* row_loss, float* row_correct, float* row_accept, int* row_pred,
                              const long long* row_map, void* stream) {
    SF_CHECK_ARG(rows >= 0 && V > 0 && S > 0 && Spad >= S && off >= 0 && off + S <= Spad && (row_map || rows % S == 0), "sf_ce_fused_zt: bad shape");
    SF_CHECK_ARG(dtype == SF_BF16 || dtype == SF_F32, "sf_ce_fused_zt: dtype");
    SF_CHECK_ARG(ld % 8 == 0 && V % 8 == 0 && ldzt % 8 == 0 && ldzt >= V, "sf_ce_fused_zt: ld, ldzt and V must be multiples of 8");
    SF_CHECK_ARG(zt && zmd_pad && zinv_pad && tsum_pad && pos_mask_pad, "sf_ce_fused_zt: teacher logits, their row scalars and tsum are required");
    if (rows == 0) return 0;
    if (dtype == SF_BF16)
        SF_LAUNCH((ce_fused_kernel<sf_bf16, 1>), dim3(rows), dim3(256), 0, stream, (sf_bf16*)logits, ld, V, (const float*)nullptr, S, Spad,
                  off, pos_mask_pad, loss_mask_pad, tgt_ids_pad, pod_scale_pad, tsum_pad, d2t, grad_scale, write_grad,
                  row_loss, row_correct, row_accept, row_pred, (const sf_bf16*)zt, ldzt, zmd_pad, zinv_pad, row_map);
    else
        SF_LAUNCH((ce_fused_kernel<float, 1>), dim3(rows), dim3(256), 0, stream, (float*)logits, ld, V, (const float*)nullptr, S, Spad,
                  off, pos_mask_pad, loss_mask_pad, tgt_ids_pad, pod_scale_pad, tsum_pad, d2t, grad_scale, write_grad,
                  row_loss, row_correct, row_accept, row_pred, (const sf_bf16*)zt, ldzt, zmd_pad, zinv_pad, row_map);
    return sf_check_launch("sf_ce_fused_zt");
}
