// Shared device helpers: 8-wide vector loads/stores of bf16 / fp32 rows, block reductions.
#pragma once
#include "sf_platform.h"

#define SF_NEG_BIG (-3.0e38f)

template <typename T> struct SfVec8;
template <> struct SfVec8<sf_bf16> {
    // 16-byte load of 8 bf16
    static SF_DEVICE void ld(const sf_bf16* p, float (&o)[8]) {
        sf_v8s v = *reinterpret_cast<const sf_v8s*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = sf_bf2f((sf_bf16)v[i]);
    }
    static SF_DEVICE void st(sf_bf16* p, const float (&o)[8]) {
        sf_v8s v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (short)sf_f2bf(o[i]);
        *reinterpret_cast<sf_v8s*>(p) = v;
    }
};
template <> struct SfVec8<float> {
    static SF_DEVICE void ld(const float* p, float (&o)[8]) {
        sf_v4f a = *reinterpret_cast<const sf_v4f*>(p);
        sf_v4f b = *reinterpret_cast<const sf_v4f*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
    }
    static SF_DEVICE void st(float* p, const float (&o)[8]) {
        sf_v4f a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = o[i]; b[i] = o[4 + i]; }
        *reinterpret_cast<sf_v4f*>(p) = a;
        *reinterpret_cast<sf_v4f*>(p + 4) = b;
    }
};

// 8 elements as loaded (no conversion at the load site: a converted prefetch would make the compiler wait for the data
// right where the load is issued)
template <typename T> struct SfRaw8;
template <> struct SfRaw8<sf_bf16> {
    sf_v8s v;
    SF_DEVICE void ld(const sf_bf16* p) { v = *reinterpret_cast<const sf_v8s*>(p); }
    SF_DEVICE float at(int i) const { return sf_bf2f((sf_bf16)v[i]); }
};
template <> struct SfRaw8<float> {
    sf_v4f a, b;
    SF_DEVICE void ld(const float* p) { a = *reinterpret_cast<const sf_v4f*>(p); b = *reinterpret_cast<const sf_v4f*>(p + 4); }
    SF_DEVICE float at(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};

// d(SwiGLU): act = round_T(silu(g)) * u  ->  dg = da * u * silu'(g), du = da * round_T(silu(g)); one definition for the
// standalone kernel and the fused GEMM epilogue, so the two produce the same bits.  The sigmoid's exponential is one multiply +
// v_exp_f32 (relative error ~1e-6, far inside the bf16 rounding that follows); libm's expf made the fused epilogue VALU-bound
// act = round_T(silu(g)) * u (llama3_eagle.py:1518-1549: act_fn(gate_proj(x)) is a bf16 tensor before the product): one
// definition for swiglu_fwd_kernel and the fused gate|up GEMM epilogue (sf_gemm_nt_swiglu_fwd), so both give the same bits
template <typename T>
SF_DEVICE float sf_swiglu_fwd_elem(float g, float up) {
    const float sg = sf_rcp_fast(1.0f + sf_exp_fast(-g));
    return SfElem<T>::rnd(g * sg) * up;
}
template <typename T>
SF_DEVICE void sf_swiglu_bwd_elem(float g, float up, float da, float& dg, float& du) {
    const float sg = sf_rcp_fast(1.0f + sf_exp_fast(-g));
    const float silu = g * sg;
    dg = da * up * (sg * (1.0f + g * (1.0f - sg)));
    du = da * SfElem<T>::rnd(silu);
}

// Sum / max over a whole workgroup (blockDim.x multiple of 64, <= 1024).  `red` is LDS
// scratch of >= 16 floats owned by the caller; result is returned to every thread.
SF_DEVICE float sf_block_sum(float v, float* red) {
    v = sf_wave_sum(v);
    const int w = (int)threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
    sf_syncthreads();
    if (sf_lane() == 0) red[w] = v;
    sf_syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
SF_DEVICE float sf_block_max(float v, float* red) {
    v = sf_wave_max(v);
    const int w = (int)threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
    sf_syncthreads();
    if (sf_lane() == 0) red[w] = v;
    sf_syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}
