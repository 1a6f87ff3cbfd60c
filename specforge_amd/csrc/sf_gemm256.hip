// 256x256x64 "ping-pong" bf16 MFMA GEMM (NT form), the fast path of sf_gemm_nt for the big
// draft-layer shapes (K % 64 == 0).  C[M,N] = alpha * A[M,K].B[N,K]^T (+beta*C) (+R).
//
// Structure (CDNA4 guide: 256^2 tile, 8 waves as 2(M) x 4(N), BK = 64, LDS-DMA staging, counted
// vmcnt, raw s_barrier, two wave groups staggered by one barrier):
//   * 512 threads; wave (wr, wc) owns a 128 x 64 output block = 8 x 4 mfma_f32_16x16x32_bf16
//     tiles (128 accumulator VGPRs).  Waves w and w+4 share a SIMD and belong to different
//     groups (wr = w >> 2), so on every SIMD one wave is in its MFMA segment while the other
//     is in its LDS-read / DMA-issue segment.
//   * LDS = 2 K-tile buffers x {A,B} x 2 half-tiles (128 rows x 64 k, 16 KiB, XOR-swizzled
//     16-byte chunks) = 128 KiB.  Group wr reads only A half wr; wave wc reads B half wc>>1.
//   * A K-tile is consumed in 4 phases, one 64x32 output quadrant each (16 MFMAs):
//       p0: read A[m-half 0] (8 x ds_read_b128) + B[n-half 0] (4)   -> quadrant (0,0)
//       p1: read B[n-half 1] (4)                                    -> quadrant (0,1)
//       p2: read A[m-half 1] (8, reuses the A registers)            -> quadrant (1,1)
//       p3: no reads (B[n-half 0] was kept)                         -> quadrant (1,0)
//     phase = {reads, one half-tile DMA prefetch (2 x global_load_lds per wave), [p3: counted
//     vmcnt], lgkmcnt(0), s_barrier, 16 MFMA, s_barrier}.
//   * DMA schedule (t = current K-tile): p0 issues A-half0(t+1), p1 A-half1(t+1), p2
//     B-half0(t+2), p3 B-half1(t+2).  B halves of K-tile t are dead after p1, A halves after
//     p2, and every wave has drained its ds_reads (lgkmcnt(0)) before the barrier that
//     precedes the restaging phase -> no WAR.  At p3 `vmcnt(4)` retires everything except the
//     two newest half-tiles (B(t+2)), i.e. all of K-tile t+1; the readers pass two more
//     barriers before they touch it -> RAW safe for both (staggered) groups.
#include "sf_api_internal.h"
#include "sf_util.h"
#include "sf_gemm_epilogue.h"
#include <stdlib.h>

namespace {

#ifdef SF_EMU
static const sf_bf16 sf_zero16b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
__device__ const sf_bf16 sf_zero16b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kHalfBytes = 128 * TK * 2;       // 16 KiB
constexpr int kBufBytes = 4 * kHalfBytes;      // A0 A1 B0 B1 of one K-tile

struct Gemm256Args {
    const sf_bf16* A; long lda;
    const sf_bf16* B; long ldb;
    SfGemmEpi e;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;     // tile-rows per L2 group of the XCD-aware tile order
    int flags;  // tuning experiments: bit0 = no s_setprio around the MFMA segment
};

#ifdef SF_EMU
SF_DEVICE void raw_barrier() { sfemu::block_barrier(); }
SF_DEVICE void wait_lgkm0() {}
SF_DEVICE void wait_vm4() {}
SF_DEVICE void wait_vm8() {}
SF_DEVICE void sched_fence() {}
#else
SF_DEVICE void raw_barrier() { __builtin_amdgcn_s_barrier(); }
SF_DEVICE void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void wait_vm4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
SF_DEVICE void wait_vm8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
SF_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

SF_DEVICE void tile_coords256(int bid, int nblk, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int seq = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    const int per_group = GM * tiles_n;
    const int g = seq / per_group;
    const int first_m = g * GM;
    const int gsize = (tiles_m - first_m < GM) ? (tiles_m - first_m) : GM;
    const int in_g = seq - g * per_group;
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
}

template <int OUT_F32>
SF_GLOBAL void SF_LAUNCH_BOUNDS(512, 2) gemm_nt_256_kernel(Gemm256Args p) {
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    int tm, tn;
    tile_coords256((int)blockIdx.x, (int)gridDim.x, p.tiles_m, p.tiles_n, p.gm, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int nkt = p.K / TK;

    // ---- DMA sources: this wave stages pieces 2*wave, 2*wave+1 (8 rows x 128 B each) of every half-tile
    const int srow = lane >> 3;                       // row inside a piece
    const int slc = (lane & 7) ^ (srow & 7);          // logical 16-byte chunk fetched into physical chunk lane&7
    const sf_bf16* srcA[2][2];
    const sf_bf16* srcB[2][2];
    long incA[2][2], incB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = h * 128 + (2 * wave + j) * 8 + srow;
            const bool okA = m0 + r < p.M, okB = n0 + r < p.N;
            srcA[h][j] = okA ? p.A + (long)(m0 + r) * p.lda + slc * 8 : sf_zero16b;
            srcB[h][j] = okB ? p.B + (long)(n0 + r) * p.ldb + slc * 8 : sf_zero16b;
            incA[h][j] = okA ? TK : 0;
            incB[h][j] = okB ? TK : 0;
        }
    auto issue = [&](int op, int h, int kt) {  // op 0 = A, 1 = B; K-tile kt into buffer kt&1
        char* dst = smem + (kt & 1) * kBufBytes + (op * 2 + h) * kHalfBytes + (2 * wave) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const sf_bf16* s = op == 0 ? srcA[h][j] + (long)kt * incA[h][j] : srcB[h][j] + (long)kt * incB[h][j];
            sf_glds16(s, dst + j * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a half-tile); (row & 7) == (lane & 7) for every fragment row
    const int frow = lane & 15;
    const int swz0 = (((lane >> 4)) ^ (lane & 7)) << 4;
    const int swz1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;
    const int a_off = frow * 128;                                 // + (mh*64 + mt*16)*128
    const int b_off = ((wc & 1) * 64 + frow) * 128;               // + (nh*32 + nt*16)*128

    sf_v4f acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sf_v4f{0.f, 0.f, 0.f, 0.f};
    sf_v8s a[4][2], b0[2][2], b1[2][2];

    // ---- prologue: K-tile 0 complete, B halves of K-tile 1 in flight
    issue(0, 0, 0); issue(0, 1, 0); issue(1, 0, 0); issue(1, 1, 0);
    if (nkt > 1) { issue(1, 0, 1); issue(1, 1, 1); }
    sf_wait_vm0();
    raw_barrier();
    if (wr == 1) raw_barrier();  // stagger: group 1 runs one barrier behind group 0

    for (int t = 0; t < nkt; ++t) {
        const char* bufA = smem + (t & 1) * kBufBytes + wr * kHalfBytes;
        const char* bufB = smem + (t & 1) * kBufBytes + (2 + (wc >> 1)) * kHalfBytes;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            // ------------------------------------------------ load segment
            if (ph == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    b0[nt][0] = *reinterpret_cast<const sf_v8s*>(bufB + b_off + (nt * 16) * 128 + swz0);
                    b0[nt][1] = *reinterpret_cast<const sf_v8s*>(bufB + b_off + (nt * 16) * 128 + swz1);
                }
            }
            if (ph == 1) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    b1[nt][0] = *reinterpret_cast<const sf_v8s*>(bufB + b_off + (32 + nt * 16) * 128 + swz0);
                    b1[nt][1] = *reinterpret_cast<const sf_v8s*>(bufB + b_off + (32 + nt * 16) * 128 + swz1);
                }
            }
            if (ph == 0 || ph == 2) {
                const int mh = ph >> 1;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    a[mt][0] = *reinterpret_cast<const sf_v8s*>(bufA + a_off + (mh * 64 + mt * 16) * 128 + swz0);
                    a[mt][1] = *reinterpret_cast<const sf_v8s*>(bufA + a_off + (mh * 64 + mt * 16) * 128 + swz1);
                }
            }
            if (ph == 0 && t + 1 < nkt) issue(0, 0, t + 1);
            if (ph == 1 && t + 1 < nkt) issue(0, 1, t + 1);
            if (ph == 2 && t + 2 < nkt) issue(1, 0, t + 2);
            if (ph == 3) {
                if (t + 2 < nkt) {
                    issue(1, 1, t + 2);
                    wait_vm4();      // all of K-tile t+1 has landed (this wave's pieces)
                } else {
                    sf_wait_vm0();
                }
            }
            wait_lgkm0();            // my ds_reads are done before anyone may restage what I read
            raw_barrier();
            sched_fence();
            // ------------------------------------------------ compute segment: one 64x32 quadrant
            const int mh = (ph >= 2) ? 1 : 0;
            const int nh = (ph == 1 || ph == 2) ? 1 : 0;
            if (!(p.flags & 1)) sf_setprio_hi();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mh * 4 + mt][nh * 2 + nt] =
                            sf_mfma16(nh ? b1[nt][ks] : b0[nt][ks], a[mt][ks], acc[mh * 4 + mt][nh * 2 + nt]);
            if (!(p.flags & 1)) sf_setprio_lo();
            sched_fence();
            raw_barrier();
        }
    }
    if (wr == 0) raw_barrier();  // group 0 catches up (equal barrier counts)

    // ---- epilogue: lane owns C[m][n..n+3]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            sf_gemm_store4<OUT_F32>(p.e, m0 + wr * 128 + i * 16 + (lane & 15), n0 + wc * 64 + j * 16 + 4 * (lane >> 4), v);
        }
}


#ifdef SF_ABLATE
#include "../../tools/experiments/sf_gemm256_variants.inc"
#endif

}  // namespace

#ifdef SF_EMU
#define SF_ALLOW_SMEM2(kernel)
#else
#define SF_ALLOW_SMEM2(kernel)                                                                                   \
    do {                                                                                                         \
        static bool done_ = false;                                                                               \
        if (!done_) {                                                                                            \
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes); \
            (void)hipGetLastError();                                                                             \
            done_ = true;                                                                                        \
        }                                                                                                        \
    } while (0)
#endif

// launched by sf_gemm_nt (sf_gemm.hip) when the shape qualifies
int sf_gemm_nt_256_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype,
                          void* stream) {
    const int M = e.M, N = e.N;
    Gemm256Args p;
    p.A = (const sf_bf16*)A; p.lda = lda;
    p.B = (const sf_bf16*)B; p.ldb = ldb;
    p.e = e;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + TM - 1) / TM;
    p.tiles_n = (N + TN - 1) / TN;
    p.gm = sf_knob("SF_GEMM_GM", 4);
    if (p.gm < 1) p.gm = 1;
    p.flags = sf_knob("SF_GEMM_FLAGS", 0);
    const long nblk = (long)p.tiles_m * p.tiles_n;
#ifndef SF_EMU
    static bool attr_set = false;
    if (!attr_set) {  // 128 KiB of dynamic LDS needs the opt-in
        hipFuncSetAttribute((const void*)gemm_nt_256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes);
        hipFuncSetAttribute((const void*)gemm_nt_256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes);
#ifdef SF_ABLATE
        hipFuncSetAttribute((const void*)gemm_nt_256_mf32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes);
        hipFuncSetAttribute((const void*)gemm_nt_256_mf32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes);
#endif
        (void)hipGetLastError();
        attr_set = true;
    }
#endif
#ifdef SF_ABLATE   // A/B variants, tools only (tools/experiments/sf_gemm256_variants.inc)
    static const int variant = [] { const char* e = getenv("SF_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
    if (variant >= 1) {
#define SF_V2_CASE(V)                                                                                                    \
    case V:                                                                                                              \
        SF_ALLOW_SMEM2((gemm_nt_256v2_kernel<0, V - 1>));                                                                \
        SF_ALLOW_SMEM2((gemm_nt_256v2_kernel<1, V - 1>));                                                                \
        if (c_dtype == SF_F32) SF_LAUNCH((gemm_nt_256v2_kernel<1, V - 1>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p); \
        else SF_LAUNCH((gemm_nt_256v2_kernel<0, V - 1>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);     \
        break;
        switch (variant) {
            SF_V2_CASE(1) SF_V2_CASE(2) SF_V2_CASE(3) SF_V2_CASE(5) SF_V2_CASE(7) SF_V2_CASE(9) SF_V2_CASE(13) SF_V2_CASE(15)
            default: SF_CHECK_ARG(false, "unknown SF_GEMM_VARIANT");
        }
#undef SF_V2_CASE
        return sf_check_launch("sf_gemm_nt(256v2)");
    }
    static const bool mf32 = [] { const char* e = getenv("SF_GEMM_MFMA"); return e ? atoi(e) == 32 : false; }();
    if (mf32) {
        if (c_dtype == SF_F32)
            SF_LAUNCH((gemm_nt_256_mf32_kernel<1>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);
        else
            SF_LAUNCH((gemm_nt_256_mf32_kernel<0>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);
        return sf_check_launch("sf_gemm_nt(256 mf32)");
    }
#endif
    if (c_dtype == SF_F32)
        SF_LAUNCH((gemm_nt_256_kernel<1>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);
    else
        SF_LAUNCH((gemm_nt_256_kernel<0>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);
    return sf_check_launch("sf_gemm_nt(256)");
}
