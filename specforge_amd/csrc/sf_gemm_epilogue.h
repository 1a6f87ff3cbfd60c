// Shared epilogue of the bf16 MFMA GEMM kernels (sf_gemm.hip, sf_gemm256.hip, sf_gemm256w4.hip).
// A lane owns C[m][n .. n+3]:
//   v = alpha * acc  (+ Cadd[rowmap(m)][n..])  (+ beta * C)  ; bf16 output with R: round first, then + R
// Cadd is an fp32 addend whose row for output row r = b*S + s is b*Spad + s + off: the precomputed
// embedding half of the TTT-step QKV projection (same token, shifted by the step index) joins the fp32
// accumulator before the single bf16 rounding, exactly where the reference's one GEMM over the concatenated
// [embed | hidden] input would have summed it (llama3_eagle.py:1625-1630, 661-700).
#pragma once
#include "sf_util.h"

struct SfGemmEpi {
    void* C; long ldc;
    const sf_bf16* R; long ldr;
    const float* Cadd; long ldadd;
    int add_S, add_Spad, add_off;
    int M, N;
    float alpha, beta;
    // fused d(SwiGLU) (4-wave kernel, interior bf16 tiles only; the launcher guarantees it): the GEMM's result is d(act)
    // [M, N = I]; instead of storing it, every 8-value row segment reads gate / up at the same position of gu [M, 2I] and
    // writes d(gate) / d(up) to dgu [M, 2I].  Null = plain store.
    const sf_bf16* sw_gu = nullptr; long sw_ldgu = 0;
    sf_bf16* sw_dgu = nullptr; long sw_lddgu = 0;
    // teacher-head form (4-wave kernel, sf_gemm_nt_teacher): tiles whose first column is >= red_n0 are REDUCED instead of
    // stored -- per row and 128-column block {max, sum exp(z - max), argmax column, 0} of the bf16-rounded logits goes to
    // red_part[(row * red_stride + block) * 4], block = (column - red_n0) / 128.  Null = plain store everywhere.
    float* red_part = nullptr; long red_stride = 0; int red_n0 = 0;
};

// ADD = 0 compiles the addend out (the 4-wave kernel's register allocation is sensitive to epilogue code, so its
// plain instantiation must not carry the extra branch)
template <int OUT_F32, int ADD = 1>
SF_DEVICE void sf_gemm_store4(const SfGemmEpi& p, int m, int n, float (&v)[4]) {
    if (m >= p.M || n >= p.N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
    const bool full = (n + 3 < p.N);
    if (ADD && p.Cadd) {
        const int bb = m / p.add_S;
        const float* a = p.Cadd + ((long)bb * p.add_Spad + (m - bb * p.add_S) + p.add_off) * p.ldadd + n;
        if (full) {
            const sf_v4f o = *reinterpret_cast<const sf_v4f*>(a);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
        } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] += a[r];
        }
    }
    if (OUT_F32) {
        float* c = (float*)p.C + (long)m * p.ldc + n;
        if (full) {
            if (p.beta != 0.f) {
                sf_v4f o = *reinterpret_cast<const sf_v4f*>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p.beta * o[r];
            }
            *reinterpret_cast<sf_v4f*>(c) = sf_v4f{v[0], v[1], v[2], v[3]};
        } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = v[r] + (p.beta != 0.f ? p.beta * c[r] : 0.f);
        }
    } else {
        sf_bf16* c = (sf_bf16*)p.C + (long)m * p.ldc + n;
        if (full) {
            if (p.beta != 0.f) {
                sf_v4s o = *reinterpret_cast<const sf_v4s*>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p.beta * sf_bf2f((sf_bf16)o[r]);
            }
            if (p.R) {  // round the projection first, then add the residual (bf16 + bf16)
                sf_v4s rr = *reinterpret_cast<const sf_v4s*>(p.R + (long)m * p.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = sf_round_bf(v[r]) + sf_bf2f((sf_bf16)rr[r]);
            }
            sf_v4s o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (short)sf_f2bf(v[r]);
            *reinterpret_cast<sf_v4s*>(c) = o;
        } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) {
                float t2 = v[r] + (p.beta != 0.f ? p.beta * sf_bf2f(c[r]) : 0.f);
                if (p.R) t2 = sf_round_bf(t2) + sf_bf2f(p.R[(long)m * p.ldr + n + r]);
                c[r] = sf_f2bf(t2);
            }
        }
    }
}
