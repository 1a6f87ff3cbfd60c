// 256x256x64 bf16 MFMA GEMM (NT form), 4 waves x (128 x 128) -- one wave per SIMD, software-pipelined
// in ONE instruction stream per wave instead of the two barrier-staggered wave groups of sf_gemm256.hip.
//
// Why: the ping-pong kernel needs 8 workgroup barriers per K-tile to hand the matrix pipe from one wave
// group to the other; measured on MI355X an MFMA+barrier-only ablation of that loop tops out at ~80 % of
// the sustained MFMA rate (each barrier costs ~45 cycles against 256 cycles of MFMA), and the load
// segments only partially hide under the partner's MFMAs.  Here every SIMD runs one wave that owns the
// pipe for the whole K loop; ds_reads and LDS-DMA issues are threaded between its MFMAs (one per 4 MFMAs),
// and there is ONE barrier per K-tile.
//
//   * 256 threads; wave (wr, wc) of a 2 x 2 grid owns a 128 x 128 output block = 8 x 8
//     mfma_f32_16x16x32_bf16 tiles (256 accumulator registers; the kernel runs at one wave per SIMD with
//     the full 512-entry unified VGPR/AGPR file).
//   * LDS = 2 K-tile buffers x {A 256 rows, B 256 rows} x 128 B, XOR-swizzled 16-byte chunks = 128 KiB.
//   * a K-tile is two half-steps (k = 0..31, 32..63) of 64 MFMAs; fragments are double-buffered in
//     registers: half-step h computes from set h&1 while the 16 ds_read_b128 of half-step h+1 fill the
//     other set.
//   * per K-tile t:   half-step 2t   : 64 MFMA | 16 ds_read (tile t, k-half 1), front-loaded
//                                      vmcnt(0) lgkmcnt(0) s_barrier      <- tile t+1 visible, buffer t&1 free
//                     half-step 2t+1 : 64 MFMA | 16 ds_read (tile t+1, k-half 0) | 16 LDS-DMA (tile t+2)
//     RAW: a tile's DMA is issued one full half-step (>= 1000 cycles) before the vmcnt(0)+barrier that
//     publishes it.  WAR: every ds_read of buffer t&1 has returned (lgkmcnt(0)) before the barrier that
//     precedes its restaging.
#include "sf_api_internal.h"
#include "sf_util.h"
#include "sf_gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

#define SF_INLINE_LAMBDA __attribute__((always_inline))

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kOpBytes = 256 * TK * 2;       // 32 KiB: one operand's K-tile
constexpr int kBufBytes = 2 * kOpBytes;      // A + B

struct GemmW4Args {
    const sf_bf16* A; long lda;
    const sf_bf16* B; long ldb;
    SfGemmEpi e;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;
};

#ifdef SF_EMU
SF_DEVICE void w4_barrier() { sfemu::block_barrier(); }
SF_DEVICE void w4_wait_all() {}
SF_DEVICE void w4_wait_lgkm() {}
SF_DEVICE void w4_wait_vm16() {}
SF_DEVICE void w4_wait_vm0() {}
SF_DEVICE void w4_fence() {}
#else
SF_DEVICE void w4_barrier() { __builtin_amdgcn_s_barrier(); }
SF_DEVICE void w4_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void w4_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void w4_wait_vm16() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }   // all but the newest 16 LDS-DMA pieces
SF_DEVICE void w4_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
SF_DEVICE void w4_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

SF_DEVICE void w4_tile_coords(int bid, int nblk, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
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

template <int OUT_F32, int ADD = 1>
SF_DEVICE void w4_store4(const GemmW4Args& p, int m, int n, sf_v4f acc) {
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    sf_gemm_store4<OUT_F32, ADD>(p.e, m, n, v);
}

// ABL (timing ablations only, results are wrong): bit0 = no ds_reads after the first tile, bit1 = no DMA in the loop
template <int OUT_F32, int ABL = 0, int BUF = 0, int ADD = 0, int SCHED = 1>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) gemm_nt_256w4_kernel(GemmW4Args p) {
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id();
    const int wr = wave >> 1, wc = wave & 1;
    int tm, tn;
    w4_tile_coords((int)blockIdx.x, (int)gridDim.x, p.tiles_m, p.tiles_n, p.gm, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int nkt = p.K / TK;

    // ---- DMA sources: this wave stages pieces 8*wave .. 8*wave+7 (8 rows x 128 B each) of A and of B.
    // Rows past the matrix edge re-read the last valid row: they only feed accumulators that are never stored.
    // (Register allocation of this kernel is fragile: running 64-bit pointers + a peeled tail allocate cleanly --
    // 188 VGPR + 256 AGPR, no copies in the loop; a uniform loop body with clamped tile indices made the
    // compiler shuffle accumulators through VGPRs and lost 30 %.)
    const int srow = lane >> 3;
    const int slc = (lane & 7) ^ (srow & 7);  // logical 16-byte chunk fetched into physical chunk lane&7
    const sf_bf16* src[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int ra = m0 + (8 * wave + j) * 8 + srow, rb = n0 + (8 * wave + j) * 8 + srow;
        ra = ra < p.M ? ra : p.M - 1;
        rb = rb < p.N ? rb : p.N - 1;
        src[j] = p.A + (long)ra * p.lda + slc * 8;
        src[8 + j] = p.B + (long)rb * p.ldb + slc * 8;
    }
    // BUF: the same pieces through raw buffer descriptors rooted at the tile origin -- one 32-bit voffset per piece
    // and lane, the K advance in soffset (scalar), no per-DMA vector arithmetic
    const SfBuf bufA = sf_make_buf(p.A + (long)m0 * p.lda, 0x7fffffffu);
    const SfBuf bufB = sf_make_buf(p.B + (long)n0 * p.ldb, 0x7fffffffu);
    unsigned voff[16];
    if (BUF) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int ra = (8 * wave + j) * 8 + srow, rb = ra;
            ra = m0 + ra < p.M ? ra : p.M - 1 - m0;
            rb = n0 + rb < p.N ? rb : p.N - 1 - n0;
            voff[j] = (unsigned)(((long)ra * p.lda + slc * 8) * 2);
            voff[8 + j] = (unsigned)(((long)rb * p.ldb + slc * 8) * 2);
        }
    }
    auto dma = [&](int g, int kt) {  // piece g (0..7 A, 8..15 B) of the next un-issued K-tile into buffer kt&1
        char* dst = smem + (kt & 1) * kBufBytes + (g >> 3) * kOpBytes + (8 * wave + (g & 7)) * 1024;
        if (BUF) {
            sf_buf_glds16(g < 8 ? bufA : bufB, voff[g], (unsigned)kt * (TK * 2), dst);
        } else {
            if (SCHED == 1) sf_glds16_opaque(src[g], dst); else sf_glds16(src[g], dst);
            src[g] += TK;
        }
    };

    // ---- fragment read offsets; (row & 7) == (lane & 7) for every fragment row
    const int frow = lane & 15;
    int swz[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) swz[ks] = ((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
    const int a_off = (wr * 128 + frow) * 128;
    const int b_off = kOpBytes + (wc * 128 + frow) * 128;

    sf_v4f acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = sf_v4f{0.f, 0.f, 0.f, 0.f};
    if (ADD) {
        // row-mapped fp32 addend: START the accumulators from it (alpha == 1 is enforced by the launcher), so the
        // K loop and the epilogue are exactly the plain kernel's -- the loads overlap the staging of K-tiles 0 and 1
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            if (m < p.M) {
                const int bb = m / p.e.add_S;
                const float* a = p.e.Cadd + ((long)bb * p.e.add_Spad + (m - bb * p.e.add_S) + p.e.add_off) * p.e.ldadd;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + wc * 128 + j * 16 + 4 * (lane >> 4);
                    if (n + 3 < p.N) acc[i][j] = *reinterpret_cast<const sf_v4f*>(a + n);
                    else
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) acc[i][j][r] = a[n + r];
                }
            }
        }
    }
    sf_v8s f[2][16];  // [set][0..7 = B n-tiles, 8..15 = A m-tiles]

    auto read_frag = [&](int set, int g, const char* buf, int ks) {
        if (g < 8) f[set][g] = *reinterpret_cast<const sf_v8s*>(buf + b_off + g * 2048 + swz[ks]);
        else f[set][g] = *reinterpret_cast<const sf_v8s*>(buf + a_off + (g - 8) * 2048 + swz[ks]);
    };

    // ---- prologue: K-tiles 0 and 1 staged, fragments of half-step 0 in registers
#pragma unroll
    for (int g = 0; g < 16; ++g) dma(g, 0);
    if (nkt > 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g) dma(g, 1);
    }
    w4_wait_all();
    w4_barrier();
#pragma unroll
    for (int g = 0; g < 16; ++g) read_frag(0, g, smem, 0);

    // one K-tile: READ_NEXT = tile t+1 exists, DO_DMA = tile t+2 exists
    auto tile = [&](auto READ_NEXT, auto DO_DMA, int t) {
        const char* cur = smem + (t & 1) * kBufBytes;
        const char* nxt = smem + ((t + 1) & 1) * kBufBytes;
        // ---- half-step 2t: compute set 0; fragments of k-half 1 -> set 1 (two reads per group, front-loaded)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = g * 4 + q, mt = idx >> 3, nt = idx & 7;
                acc[mt][nt] = sf_mfma16(f[0][nt], f[0][8 + mt], acc[mt][nt]);
            }
            w4_fence();   // MFMAs first, then this group's loads (the compiler otherwise hoists the loads and waits on them)
            if (g < 8 && !((ABL & 1) && t > 0)) { read_frag(1, 2 * g, cur, 1); read_frag(1, 2 * g + 1, cur, 1); }
            w4_fence();
        }
        if constexpr (ABL & 8) w4_wait_lgkm(); else
        w4_wait_all();   // my pieces of tile t+1 have landed; my reads of buffer t&1 have returned
        if constexpr (!(ABL & 16))
        w4_barrier();    // -> tile t+1 visible to everyone, buffer t&1 free for tile t+2
        // ---- half-step 2t+1: compute set 1; fragments of (t+1, k-half 0) -> set 0; stage tile t+2
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = g * 4 + q, mt = idx >> 3, nt = idx & 7;
                acc[mt][nt] = sf_mfma16(f[1][nt], f[1][8 + mt], acc[mt][nt]);
            }
            w4_fence();
            // reads front-loaded (groups 0..11) so that the next half-step's first MFMA never waits on them
            if constexpr (decltype(READ_NEXT)::value && !(ABL & 1)) {
                if (g < 4) { read_frag(0, 2 * g, nxt, 0); read_frag(0, 2 * g + 1, nxt, 0); }
                else if (g < 12) read_frag(0, g + 4, nxt, 0);
            }
            if constexpr (decltype(DO_DMA)::value && !(ABL & 2)) {
                if constexpr (ABL & 4) { if (g < 8) { dma(2 * g, t + 2); dma(2 * g + 1, t + 2); } }
                else dma(g, t + 2);
            }
            w4_fence();
        }
    };

    // SCHED 1 ("early release"): all 16 fragment reads of tile t's second k-half are issued in the first 32 MFMAs, so the
    // tile's LDS buffer is handed back after ~1/3 of the iteration (barrier 1) and the 16 DMA pieces of tile t+2 go out
    // over MFMAs 40..104 -- a FULL iteration more lead than issuing them in the second half-step; they are waited for
    // with a COUNTED vmcnt at MFMA ~104 of the NEXT iteration (this iteration's own 16 pieces stay in flight), followed by
    // barrier 2 and the first-half reads of tile t+1.  DMA lead: 1.0 .. 1.5 iterations (2200 .. 3300 cycles) instead of
    // 0.5 .. 1.0 -- HBM latency under a full chip of streaming GEMM tiles is above the shorter lead.
    //   RAW: tile t+1 is read only after (own pieces landed: vmcnt) + barrier 2.
    //   WAR: buffer t&1 is re-staged only after (own reads returned: lgkmcnt(0)) + barrier 1; its first-half fragments
    //        were read at the end of iteration t-1.
    auto tile1 = [&](auto READ_NEXT, auto DO_DMA, int t) {
        const char* cur = smem + (t & 1) * kBufBytes;
        const char* nxt = smem + ((t + 1) & 1) * kBufBytes;
#pragma unroll
        for (int g = 0; g < 32; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = (g & 15) * 4 + q, mt = idx >> 3, nt = idx & 7;
                if (g < 16) acc[mt][nt] = sf_mfma16(f[0][nt], f[0][8 + mt], acc[mt][nt]);
                else acc[mt][nt] = sf_mfma16(f[1][nt], f[1][8 + mt], acc[mt][nt]);
            }
            w4_fence();
            if (g < 8) { read_frag(1, 2 * g, cur, 1); read_frag(1, 2 * g + 1, cur, 1); }
            if (g == 9) {
                w4_wait_lgkm();      // my reads of buffer t&1 have returned
                w4_barrier();        // -> buffer t&1 is free for tile t+2
            }
            if constexpr (decltype(DO_DMA)::value) {
                if (g >= 10 && g < 26) dma(g - 10, t + 2);
            }
            if (g == 25) {
                if constexpr (decltype(DO_DMA)::value) w4_wait_vm16(); else w4_wait_vm0();   // my pieces of tile t+1 landed
                w4_barrier();        // -> tile t+1 visible to everyone
            }
            if constexpr (decltype(READ_NEXT)::value) {
                if (g >= 26 && g < 31) { read_frag(0, 3 * (g - 26), nxt, 0); read_frag(0, 3 * (g - 26) + 1, nxt, 0); read_frag(0, 3 * (g - 26) + 2, nxt, 0); }
                if (g == 31) read_frag(0, 15, nxt, 0);
            }
            w4_fence();
        }
    };

    int t = 0;
    if constexpr (SCHED == 1) {
        for (; t + 2 < nkt; ++t) tile1(std::true_type{}, std::true_type{}, t);
        if (t + 1 < nkt) { tile1(std::true_type{}, std::false_type{}, t); ++t; }
        tile1(std::false_type{}, std::false_type{}, t);
    } else {
        for (; t + 2 < nkt; ++t) tile(std::true_type{}, std::true_type{}, t);
        if (t + 1 < nkt) { tile(std::true_type{}, std::false_type{}, t); ++t; }
        tile(std::false_type{}, std::false_type{}, t);
    }

    // ---- epilogue: lane owns C[m][n..n+3]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            w4_store4<OUT_F32, 0>(p, m0 + wr * 128 + i * 16 + (lane & 15), n0 + wc * 128 + j * 16 + 4 * (lane >> 4), acc[i][j]);
}


#ifdef SF_ABLATE
#include "../../tools/experiments/sf_gemm256w4_variants.inc"
#endif

}  // namespace

#ifdef SF_EMU
#define SF_W4_SMEM(kernel)
#else
#define SF_W4_SMEM(kernel)                                                                                       \
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
int sf_gemm_nt_256w4_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype,
                            void* stream) {
    const int M = e.M, N = e.N;
    GemmW4Args p;
    p.A = (const sf_bf16*)A; p.lda = lda;
    p.B = (const sf_bf16*)B; p.ldb = ldb;
    p.e = e;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + TM - 1) / TM;
    p.tiles_n = (N + TN - 1) / TN;
    p.gm = sf_knob("SF_GEMM_GM", 4);
    if (p.gm < 1) p.gm = 1;
    const long nblk = (long)p.tiles_m * p.tiles_n;
#ifdef SF_ABLATE   // A/B variants, tools only (tools/experiments/sf_gemm256w4_variants.inc)
    static const bool m32 = [] { const char* e = getenv("SF_GEMM_MFMA"); return e ? atoi(e) == 32 : false; }();
    if (m32) {
        if (c_dtype == SF_F32) {
            SF_W4_SMEM((gemm_nt_256w4m32_kernel<1>));
            SF_LAUNCH((gemm_nt_256w4m32_kernel<1>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        } else {
            SF_W4_SMEM((gemm_nt_256w4m32_kernel<0>));
            SF_LAUNCH((gemm_nt_256w4m32_kernel<0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        }
        return sf_check_launch("sf_gemm_nt(256w4m32)");
    }
    static const bool soft = [] { const char* e = getenv("SF_GEMM_SOFT"); return e ? atoi(e) == 1 : false; }();
    if (soft && !p.e.Cadd) {
        const int smem_bytes = 2 * kBufBytes + 64;
#ifndef SF_EMU
        static bool attr_soft = false;
        if (!attr_soft) {
            hipFuncSetAttribute((const void*)gemm_nt_256w4s_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
            hipFuncSetAttribute((const void*)gemm_nt_256w4s_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
            (void)hipGetLastError();
            attr_soft = true;
        }
#endif
        static const bool hard = [] { const char* e = getenv("SF_GEMM_SOFT_HARD"); return e ? atoi(e) == 1 : false; }();
        if (hard) {
            static bool attr_h = false;
#ifndef SF_EMU
            if (!attr_h) {
                hipFuncSetAttribute((const void*)gemm_nt_256w4s_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
                (void)hipGetLastError();
                attr_h = true;
            }
#endif
            (void)attr_h;
            SF_CHECK_ARG(c_dtype != SF_F32, "SF_GEMM_SOFT_HARD: bf16 output only (A/B knob)");
            SF_LAUNCH((gemm_nt_256w4s_kernel<0, 0>), dim3((unsigned)nblk), dim3(256), smem_bytes, stream, p);
            return sf_check_launch("sf_gemm_nt(256w4 asm)");
        }
        if (c_dtype == SF_F32) SF_LAUNCH((gemm_nt_256w4s_kernel<1>), dim3((unsigned)nblk), dim3(256), smem_bytes, stream, p);
        else SF_LAUNCH((gemm_nt_256w4s_kernel<0>), dim3((unsigned)nblk), dim3(256), smem_bytes, stream, p);
        return sf_check_launch("sf_gemm_nt(256w4 soft)");
    }
    static const bool w4r = [] { const char* e = getenv("SF_GEMM_W4R"); return e ? atoi(e) == 1 : false; }();
    if (w4r && !p.e.Cadd) {
        if (c_dtype == SF_F32) {
            SF_W4_SMEM((gemm_nt_256w4r_kernel<1>));
            SF_LAUNCH((gemm_nt_256w4r_kernel<1>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        } else {
            SF_W4_SMEM((gemm_nt_256w4r_kernel<0>));
            SF_LAUNCH((gemm_nt_256w4r_kernel<0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        }
        return sf_check_launch("sf_gemm_nt(256w4r)");
    }
    static const bool w8 = [] { const char* e = getenv("SF_GEMM_W8"); return e ? atoi(e) == 1 : false; }();
    if (w8) {
        if (c_dtype == SF_F32) {
            SF_W4_SMEM((gemm_nt_256w8_kernel<1>));
            SF_LAUNCH((gemm_nt_256w8_kernel<1>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);
        } else {
            SF_W4_SMEM((gemm_nt_256w8_kernel<0>));
            SF_LAUNCH((gemm_nt_256w8_kernel<0>), dim3((unsigned)nblk), dim3(512), 2 * kBufBytes, stream, p);
        }
        return sf_check_launch("sf_gemm_nt(256w8)");
    }
    static const int abl = [] { const char* e = getenv("SF_GEMM_ABL"); return e ? atoi(e) : 0; }();
#define SF_ABL_CASE(V) \
    if (abl == V) { SF_W4_SMEM((gemm_nt_256w4_kernel<0, V, 0, 0, 0>)); SF_LAUNCH((gemm_nt_256w4_kernel<0, V, 0, 0, 0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p); return sf_check_launch("abl"); }
    SF_ABL_CASE(1) SF_ABL_CASE(2) SF_ABL_CASE(3) SF_ABL_CASE(4) SF_ABL_CASE(8) SF_ABL_CASE(9) SF_ABL_CASE(12) SF_ABL_CASE(16) SF_ABL_CASE(24)
#undef SF_ABL_CASE
    const int sched = sf_knob("SF_GEMM_SCHED", 1);   // 0 = the round-1 schedule (DMA issued in the second half-step, vmcnt(0))
    if (sched == 0) {
        SF_CHECK_ARG(c_dtype != SF_F32, "SF_GEMM_SCHED=0: bf16 output only (A/B knob)");
        if (p.e.Cadd) {
            SF_W4_SMEM((gemm_nt_256w4_kernel<0, 0, 0, 1, 0>));
            SF_LAUNCH((gemm_nt_256w4_kernel<0, 0, 0, 1, 0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        } else {
            SF_W4_SMEM((gemm_nt_256w4_kernel<0, 0, 0, 0, 0>));
            SF_LAUNCH((gemm_nt_256w4_kernel<0, 0, 0, 0, 0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        }
        return sf_check_launch("sf_gemm_nt(256w4 sched0)");
    }
    static const bool bufdma = [] { const char* e = getenv("SF_GEMM_BUF"); return e ? atoi(e) == 1 : false; }();
    if (bufdma) {
        if (c_dtype == SF_F32) {
            SF_W4_SMEM((gemm_nt_256w4_kernel<1, 0, 1, 0, 0>));
            SF_LAUNCH((gemm_nt_256w4_kernel<1, 0, 1, 0, 0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        } else {
            SF_W4_SMEM((gemm_nt_256w4_kernel<0, 0, 1, 0, 0>));
            SF_LAUNCH((gemm_nt_256w4_kernel<0, 0, 1, 0, 0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        }
        return sf_check_launch("sf_gemm_nt(256w4 buf)");
    }
#endif
    if (p.e.Cadd) {
        SF_CHECK_ARG(p.e.alpha == 1.0f, "sf_gemm_nt_rowadd: the 4-wave kernel needs alpha == 1");
        if (c_dtype == SF_F32) {
            SF_W4_SMEM((gemm_nt_256w4_kernel<1, 0, 0, 1>));
            SF_LAUNCH((gemm_nt_256w4_kernel<1, 0, 0, 1>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        } else {
            SF_W4_SMEM((gemm_nt_256w4_kernel<0, 0, 0, 1>));
            SF_LAUNCH((gemm_nt_256w4_kernel<0, 0, 0, 1>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
        }
        return sf_check_launch("sf_gemm_nt(256w4 rowadd)");
    }
    if (c_dtype == SF_F32) {
        SF_W4_SMEM((gemm_nt_256w4_kernel<1>));
        SF_LAUNCH((gemm_nt_256w4_kernel<1>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
    } else {
        SF_W4_SMEM((gemm_nt_256w4_kernel<0>));
        SF_LAUNCH((gemm_nt_256w4_kernel<0>), dim3((unsigned)nblk), dim3(256), 2 * kBufBytes, stream, p);
    }
    return sf_check_launch("sf_gemm_nt(256w4)");
}
