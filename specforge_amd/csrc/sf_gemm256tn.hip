// "TN" form of the 4-wave 256x256x64 bf16 MFMA GEMM:   C[M,N] = alpha * A^T . B (+ beta * C)
// with A stored [K, M] and B stored [K, N] (row-major, the contraction index outermost).
//
// This is the weight-gradient form of a linear layer: dW[out, in] = dY^T . X with dY [tokens, out] and
// X [tokens, in] exactly as the forward / backward sweep produced them (the reference gets it from autograd's
// `grad_output.t() @ input`, e.g. for llama3_eagle.py:555-566,1513-1515,1674-1693).  Feeding those natural layouts
// straight to the GEMM removes every operand transpose (and the transposed stash copies) that the NT-only
// kernels needed: 70 transposes / 48 GB of traffic per training step at cfg 2.
//
// Same pipeline as sf_gemm256w4.hip (4 waves x 128x128, one wave per SIMD, fragments double-buffered in
// registers, LDS-DMA threaded between the MFMAs, one barrier per K-tile).  What differs is the LDS image and the
// fragment reads:
//   * a K-tile of an operand is staged as two half-tiles of 128 columns: [64 k][256 B], i.e. rows of the
//     SOURCE matrix land as rows of LDS (an LDS-DMA instruction moves 4 k-rows x 256 B);
//   * an MFMA fragment (16 columns x 32 k, 8 consecutive k per lane) is therefore a TRANSPOSED read: two
//     ds_read_b64_tr_b16 per fragment (each returns 4 k-values of the lane's column from a 4 x 16 block);
//   * bank conflicts of those reads are broken by permuting the 16-byte chunks of k-row k with
//     c -> c ^ (2*((k & 3) | ((k >> 3) & 1) << 2)), applied to the global source address (the LDS image must stay
//     lane-linear for LDS-DMA) and again to the read address.
// Shapes: K % 64 == 0 (the caller pads its token stash with zero rows), M, N >= 8 and multiples of 8; edge tiles
// re-read the last valid 8 columns (they only feed accumulators that are never stored).
#include "sf_api_internal.h"
#include "sf_util.h"
#include "sf_gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kHalfBytes = 128 * TK * 2;     // 16 KiB: [64 k][256 B]
constexpr int kBufBytes = 4 * kHalfBytes;    // A0 A1 B0 B1
constexpr long kTnSyncFloats = 4096;         // 16 KiB of pace counters at the tail of the caller's workspace

struct GemmTnArgs {
    const sf_bf16* A; long lda;   // [K, M]
    const sf_bf16* B; long ldb;   // [K, N]
    SfGemmEpi e;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;
    int ksplit;       // > 1: blockIdx.y contracts K-range [y*K/ksplit, (y+1)*K/ksplit) into fp32 partial y of `ws`
    float* ws;        // [ksplit][M][N] fp32 partials (split-K only)
    // pace-keeping of the workgroups that share operand panels in one XCD's L2 (null = off): one zero-initialised
    // counter per group of 32 consecutive tiles of an XCD, bumped every `sync_every` K-tiles
    unsigned* sync;
    int sync_every;   // power of two
};

#ifdef SF_EMU
SF_DEVICE void tn_barrier() { sfemu::block_barrier(); }
SF_DEVICE void tn_wait_all() {}
SF_DEVICE void tn_fence() {}
#else
SF_DEVICE void tn_barrier() { __builtin_amdgcn_s_barrier(); }
SF_DEVICE void tn_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void tn_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

SF_DEVICE void tn_tile_coords(int bid, int nblk, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
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

#ifdef SF_ABLATE   // round-1 schedule (one barrier per K-tile, vmcnt(0)): tools build only, SF_GEMM_TN_PLAN=-1
template <int OUT_F32, int SPREAD = 0>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) gemm_tn_256w4_kernel(GemmTnArgs p) {
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id();
    const int wr = wave >> 1, wc = wave & 1;
    int tm, tn;
    tn_tile_coords((int)blockIdx.x, (int)gridDim.x, p.tiles_m, p.tiles_n, p.gm, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int nkt = p.K / TK / p.ksplit;
    const long k0 = (long)blockIdx.y * nkt * TK;      // first contraction row of this split
    if (p.ksplit > 1) {                               // partial sums go to the fp32 workspace, plain store
        p.e.C = p.ws + (long)blockIdx.y * p.M * p.N;
        p.e.ldc = p.N;
        p.e.alpha = 1.f;
        p.e.beta = 0.f;
    }

    // ---- DMA sources: half hh (0,1 = A columns m0+0.., m0+128..; 2,3 = B) is 16 pieces of 4 k-rows x 256 B; this wave
    // stages pieces 4*wave .. 4*wave+3 of every half.  Lane: k-row skr = lane>>4 of the piece, physical chunk lane&15.
    const int skr = lane >> 4;
    const sf_bf16* src[16];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int krow = (4 * wave + j) * 4 + skr;
            const int hk = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int lc = (lane & 15) ^ (hk << 1);            // logical 16-byte chunk fetched into physical chunk lane&15
            const bool isA = hh < 2;
            const int dim = isA ? p.M : p.N;
            int col = (isA ? m0 : n0) + (hh & 1) * 128 + lc * 8;
            col = col + 8 <= dim ? col : dim - 8;
            src[hh * 4 + j] = (isA ? p.A + (k0 + krow) * p.lda : p.B + (k0 + krow) * p.ldb) + col;
        }
    const long incA = (long)TK * p.lda, incB = (long)TK * p.ldb;
    auto dma = [&](int g, int kt) {  // g = hh*4 + j: piece 4*wave+j of half hh of the next un-issued K-tile
        char* dst = smem + (kt & 1) * kBufBytes + (g >> 2) * kHalfBytes + (4 * wave + (g & 3)) * 1024;
        sf_glds16_opaque(src[g], dst);
        src[g] += (g < 8) ? incA : incB;
    };

    // ---- fragment reads (transposed): lane constants of ds_read_b64_tr_b16 into the [64 k][256 B] image
    const int fi = lane & 15, fg = lane >> 4;
    const int fh = (fi >> 2) | ((fg & 1) << 2);
    const int frag_lane = (8 * fg + (fi >> 2)) * 256 + ((fi >> 1) & 1) * 16 + (fi & 1) * 8;
    const int a_half = wr * kHalfBytes, b_half = (2 + wc) * kHalfBytes;

    sf_v4f acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = sf_v4f{0.f, 0.f, 0.f, 0.f};
    sf_v8s f[2][16];  // [set][0..7 = B n-tiles, 8..15 = A m-tiles]

    auto read_frag = [&](int set, int g, const char* buf, int ks) {
        const char* a = buf + (g < 8 ? b_half : a_half) + ks * (32 * 256) + frag_lane + ((((g & 7)) ^ fh) << 5);
        const sf_v4s lo = sf_ds_read_tr16(a);
        const sf_v4s up = sf_ds_read_tr16(a + 4 * 256);
        f[set][g] = sf_v8s{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    };

    // ---- prologue
#pragma unroll
    for (int g = 0; g < 16; ++g) dma(g, 0);
    if (nkt > 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g) dma(g, 1);
    }
    tn_wait_all();
    tn_barrier();
#pragma unroll
    for (int g = 0; g < 16; ++g) read_frag(0, g, smem, 0);

    auto tile = [&](auto READ_NEXT, auto DO_DMA, int t) {
        const char* cur = smem + (t & 1) * kBufBytes;
        const char* nxt = smem + ((t + 1) & 1) * kBufBytes;
        // ---- half-step 2t: compute set 0; fragments of k-half 1 -> set 1 (front-loaded)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = g * 4 + q, mt = idx >> 3, nt = idx & 7;
                sf_mfma16_acc(f[0][nt], f[0][8 + mt], acc[mt][nt]);
            }
            tn_fence();
            if (SPREAD) read_frag(1, g, cur, 1);
            else if (g < 8) { read_frag(1, 2 * g, cur, 1); read_frag(1, 2 * g + 1, cur, 1); }
            tn_fence();
        }
        tn_wait_all();   // my pieces of tile t+1 have landed; my reads of buffer t&1 have returned
        tn_barrier();    // -> tile t+1 visible to everyone, buffer t&1 free for tile t+2
        // ---- half-step 2t+1: compute set 1; fragments of (t+1, k-half 0) -> set 0; stage tile t+2
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = g * 4 + q, mt = idx >> 3, nt = idx & 7;
                sf_mfma16_acc(f[1][nt], f[1][8 + mt], acc[mt][nt]);
            }
            tn_fence();
            if constexpr (decltype(READ_NEXT)::value) {
                if (SPREAD) read_frag(0, g, nxt, 0);
                else if (g < 4) { read_frag(0, 2 * g, nxt, 0); read_frag(0, 2 * g + 1, nxt, 0); }
                else if (g < 12) read_frag(0, g + 4, nxt, 0);
            }
            if constexpr (decltype(DO_DMA)::value) dma(g, t + 2);
            tn_fence();
        }
    };

    int t = 0;
    for (; t + 2 < nkt; ++t) tile(std::true_type{}, std::true_type{}, t);
    if (t + 1 < nkt) { tile(std::true_type{}, std::false_type{}, t); ++t; }
    tile(std::false_type{}, std::false_type{}, t);

    // ---- epilogue: lane owns C[m][n..n+3]
    sf_mfma_drain();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sf_acc_touch(acc[i][j]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            sf_gemm_store4<OUT_F32, 0>(p.e, m0 + wr * 128 + i * 16 + (lane & 15), n0 + wc * 128 + j * 16 + 4 * (lane >> 4), v);
        }
}

#endif

// ---------------------------------------------------------------------------------------------------------------
// Round 2: the same kernel under the plan-scheduled main loop of sf_gemm256w4_kernel.h -- every non-MFMA instruction in
// its own MFMA slot, the two operands' LDS halves released separately (3 barriers), counted vmcnt(16), DMA lead 1.1-1.6
// iterations, buffer-descriptor LDS-DMA from inline asm with the K advance on the descriptor base (a scalar 64-bit add
// per operand and K-tile; the per-lane offsets are loop constants).  A K-tile here carries 64 transposed fragment reads
// (two ds_read_b64_tr_b16 per fragment), so the plan is denser: reads take one slot each.
//   rd1(r), rd0(r): r = 2*fragment + half (fragments 0..7 = B n-tiles, 8..15 = A m-tiles; half 0 = k 0..3, 1 = k 4..7 of
//   each 8-k group)
// Pace group of a workgroup: the (up to) 32 workgroups with consecutive sequence numbers on one XCD run concurrently on
// that XCD's 32 CUs and share 4 A panels and 8 B panels through its L2 -- as long as they stay within the ~10 K-tiles
// the 4 MiB hold.  Nothing keeps them there: the DMA lead of this kernel hides exactly the latency differences that
// would otherwise slow a runaway tile down, and over 1792 K-tiles (K = 114 688) the members drift apart until every
// tile streams its own copy of the panels (measured: 39.6 GB of L2 misses per launch against 17.4 GB with perfect
// sharing).  So every `sync_every` K-tiles the members meet at a counter in global memory: wave 0 ARRIVES 8 iterations
// early (fire-and-forget atomic) and polls at the rendezvous iteration; the other waves are held by the loop's own
// barriers.  No data passes through it -- it only aligns timing -- so there is no ordering requirement and a bounded
// spin (then pacing is switched off for the rest of the tile) makes it deadlock-free whatever the placement.
struct TnPace { unsigned* ctr; unsigned members; };
SF_DEVICE TnPace tn_pace_group(unsigned* base, int bid, int y, int nblk, int ksplit) {
    // groups follow the LINEAR dispatch order (y-major), 32 per XCD at a time = one "generation" of co-resident
    // workgroups; with split-K the launcher enables pacing only when nblk % 8 == 0 (then block (x, y) sits on XCD x % 8)
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7;
    const int cnt = ksplit > 1 ? ksplit * q : q + (xcd < rem ? 1 : 0);
    const int idx = y * q + (bid >> 3);
    const int per_xcd = (ksplit * (q + 1) + 31) >> 5;      // groups per XCD (upper bound)
    const int gl = idx >> 5;
    const int left = cnt - (gl << 5);
    TnPace g;
    g.ctr = base + (xcd * per_xcd + gl);
    g.members = (unsigned)(left < 32 ? left : 32);
    return g;
}
#ifdef SF_EMU
SF_DEVICE void tn_pace_arrive(TnPace) {}
SF_DEVICE bool tn_pace_wait(TnPace, unsigned) { return true; }
#else
SF_DEVICE void tn_pace_arrive(TnPace g) { __hip_atomic_fetch_add(g.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SF_DEVICE bool tn_pace_wait(TnPace g, unsigned target) {   // false = gave up
    for (int spin = 0; spin < 4096; ++spin) {              // <= ~1 ms
        if (__hip_atomic_load(g.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}
#endif

template <int AFIRST>
struct TnPlan {
    static constexpr int barB = 21, barA = 51, bar2 = 92, vm = 16;
    static constexpr int first(int r) { return r; }                                   // 0 .. 15: first operand, k-half 1
    static constexpr int second(int r) { return 23 + r + r / 2; }                     // 23,24,26,27,...,44,45 (DMA in between)
    static constexpr int rd1(int r) {
        const bool isA = r >= 16;
        return (isA == (AFIRST != 0)) ? first(r & 15) : second(r & 15);
    }
    static constexpr int dma(int g) {      // pieces 0..7 = A halves, 8..15 = B halves
        constexpr int early[8] = {22, 25, 28, 31, 34, 37, 40, 43};
        constexpr int late[8] = {53, 58, 63, 68, 73, 78, 83, 88};
        const bool isA = g < 8;
        return (isA == (AFIRST != 0)) ? early[g & 7] : late[g & 7];
    }
    static constexpr int rd0(int r) { return 94 + r; }                                // 94 .. 125, one read per slot
};

// slot -> filler lookups, evaluated by the constant evaluator (no template instantiation per candidate)
template <class P> constexpr int tn_rd1_at(int i) { for (int r = 0; r < 32; ++r) if (P::rd1(r) == i) return r; return -1; }
template <class P> constexpr int tn_rd0_at(int i) { for (int r = 0; r < 32; ++r) if (P::rd0(r) == i) return r; return -1; }
template <class P> constexpr int tn_dma_at(int i) { for (int g = 0; g < 16; ++g) if (P::dma(g) == i) return g; return -1; }

// M0 (LDS destination) of piece g is written right after the PREVIOUS piece's DMA, in that piece's slot (the early group
// has no empty slot between its DMAs); the first piece of an iteration gets the free slot 16.  The DMA itself is then one
// instruction: no M0 write + s_nop in front of it (see sf_gemm256w4_kernel.h, same reasoning).
template <class P> constexpr int tn_dma_prev_slot(int g) {
    int best = -1;
    for (int h = 0; h < 16; ++h) if (P::dma(h) < P::dma(g) && P::dma(h) > best) best = P::dma(h);
    return best;
}
template <class P> constexpr int tn_m0_at(int i) {
    for (int g = 0; g < 16; ++g) {
        const int prev = tn_dma_prev_slot<P>(g);
        if ((prev < 0 ? 16 : prev) == i) return g;
    }
    return -1;
}

// The fragment read addresses (one register per fragment: the XOR swizzle term differs per fragment, so they cannot
// share a base + immediate) are moved to the other K-tile buffer ONCE per iteration, in slots of their own between the
// last read of tile t (slot 45) and the first read of tile t+1 (slot 94); re-deriving them next to every pair of reads
// (what the compiler does with a loop-invariant base + per-fragment offset) put a VALU add into 32 read slots.
template <class P> constexpr bool tn_mid_busy(int i) { return tn_dma_at<P>(i) >= 0 || i == P::barA || i == P::barB || i == P::bar2; }
template <class P> constexpr int tn_tog_at(int i) {
    int g = 0;
    for (int s = 46; s < 94 && g < 16; ++s) {
        if (tn_mid_busy<P>(s)) continue;
        if (s == i) return g;
        ++g;
    }
    return -1;
}

template <int I, int N, class F>
SF_DEVICE void tn_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        tn_static_for<I + 1, N>(f);
    }
}
#define SF_TN_LAMBDA __attribute__((always_inline))
#ifdef SF_EMU
SF_DEVICE void tn_wait_lgkm() {}
SF_DEVICE void tn_wait_vm16() {}
SF_DEVICE void tn_wait_vm0() {}
#else
SF_DEVICE void tn_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void tn_wait_vm16() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
SF_DEVICE void tn_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

template <int OUT_F32, int AFIRST>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) gemm_tn_256w4p_kernel(GemmTnArgs p) {
    using P = TnPlan<AFIRST>;
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id();
    const int wr = wave >> 1, wc = wave & 1;
    int tm, tn;
    tn_tile_coords((int)blockIdx.x, (int)gridDim.x, p.tiles_m, p.tiles_n, p.gm, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int nkt = p.K / TK / p.ksplit;
    const long k0 = (long)blockIdx.y * nkt * TK;      // first contraction row of this split
    if (p.ksplit > 1) {                               // partial sums go to the fp32 workspace, plain store
        p.e.C = p.ws + (long)blockIdx.y * p.M * p.N;
        p.e.ldc = p.N;
        p.e.alpha = 1.f;
        p.e.beta = 0.f;
    }

    // ---- DMA sources: half hh (0,1 = A columns m0+0.., m0+128..; 2,3 = B) is 16 pieces of 4 k-rows x 256 B; this wave
    // stages pieces 4*wave .. 4*wave+3 of every half.  Lane: k-row skr = lane>>4 of the piece, physical chunk lane&15.
    // Per-lane byte offset from the K-tile's first row of the operand (loop constant); the descriptor base advances.
    const int skr = lane >> 4;
    unsigned voff[16];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int krow = (4 * wave + j) * 4 + skr;
            const int hk = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int lc = (lane & 15) ^ (hk << 1);            // logical 16-byte chunk fetched into physical chunk lane&15
            const bool isA = hh < 2;
            const int dim = isA ? p.M : p.N;
            int col = (isA ? m0 : n0) + (hh & 1) * 128 + lc * 8;
            col = col + 8 <= dim ? col : dim - 8;
            voff[hh * 4 + j] = (unsigned)(((long)krow * (isA ? p.lda : p.ldb) + col) * 2);
        }
    const sf_bf16* baseA = p.A + k0 * p.lda;
    const sf_bf16* baseB = p.B + k0 * p.ldb;
    const long incA = (long)TK * p.lda, incB = (long)TK * p.ldb;
    auto dma_dst = [&](int g, int kt) -> char* {
        return smem + (kt & 1) * kBufBytes + (g >> 2) * kHalfBytes + (4 * wave + (g & 3)) * 1024;
    };
    auto dma = [&](int g, int kt) {  // g = hh*4 + j: piece 4*wave+j of half hh of K-tile kt
        const SfBufRaw b = sf_make_buf_raw(g < 8 ? baseA + (long)kt * incA : baseB + (long)kt * incB);
        sf_buf_glds16_opaque(b, voff[g], 0u, dma_dst(g, kt));
    };
#ifndef SF_EMU
    // the loop's DMA: descriptors of K-tile t+2, advanced once per iteration (64-bit base: K * ld * 2 can pass 4 GiB, so
    // the K advance cannot live in the 32-bit scalar offset), M0 written one slot ahead
    SfBufRaw dscA = sf_make_buf_raw(baseA + 2 * incA), dscB = sf_make_buf_raw(baseB + 2 * incB);
    auto dsc_advance = [&](SfBufRaw& d, long inc_elems) {
        const unsigned long long a = (((unsigned long long)(unsigned)d.w[1] << 32) | (unsigned)d.w[0]) + (unsigned long long)inc_elems * 2;
        d.w[0] = (int)(unsigned)a;
        d.w[1] = (int)(unsigned)(a >> 32);   // bits 63:48 (stride / swizzle) stay zero: addresses are < 2^48
    };
#endif

    // ---- fragment reads (transposed): lane constants of ds_read_b64_tr_b16 into the [64 k][256 B] image
    const int fi = lane & 15, fg = lane >> 4;
    const int fh = (fi >> 2) | ((fg & 1) << 2);
    const int frag_lane = (8 * fg + (fi >> 2)) * 256 + ((fi >> 1) & 1) * 16 + (fi & 1) * 8;
    const int a_half = wr * kHalfBytes, b_half = (2 + wc) * kHalfBytes;

    sf_v4f acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = sf_v4f{0.f, 0.f, 0.f, 0.f};
    sf_v4s flo[2][16], fup[2][16];  // [set][0..7 = B n-tiles, 8..15 = A m-tiles]: k 0..3 / k 4..7 of each lane's 8-k group

    // ra[g]: address of fragment g's piece in the buffer the NEXT reads use: buffer t&1 at the start of
    // iteration t (k-half 1 of tile t), moved to buffer (t+1)&1 in the middle of the iteration (k-half 0 of tile t+1)
    const char* ra[16];   // (pointers, not offsets: the dynamic-LDS base is a link-time constant the compiler re-adds per use)
#pragma unroll
    for (int g = 0; g < 16; ++g) ra[g] = smem + (g < 8 ? b_half : a_half) + frag_lane + ((((g & 7)) ^ fh) << 5);
    auto read_half = [&](int set, int g, int ks, int half) {
        const char* a = ra[g] + ks * (32 * 256) + half * (4 * 256);
        if (half == 0) flo[set][g] = sf_ds_read_tr16(a);
        else fup[set][g] = sf_ds_read_tr16(a);
    };
    auto frag = [&](int set, int g) {
        const sf_v4s lo = flo[set][g], up = fup[set][g];
        return sf_v8s{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    };

    // ---- prologue
#pragma unroll
    for (int g = 0; g < 16; ++g) dma(g, 0);
    if (nkt > 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g) dma(g, 1);
    }
    tn_wait_all();
    tn_barrier();
#pragma unroll
    for (int g = 0; g < 16; ++g) { read_half(0, g, 0, 0); read_half(0, g, 0, 1); }

    auto tilep = [&](auto READ_NEXT, auto DO_DMA, int t) {
        const int flip = (t & 1) ? -kBufBytes : kBufBytes;
        tn_static_for<0, 128>([&](auto I) SF_TN_LAMBDA {
            constexpr int i = decltype(I)::value, idx = i & 63, mt = idx >> 3, nt = idx & 7, set = i >> 6;
            sf_mfma16_acc(frag(set, nt), frag(set, 8 + mt), acc[mt][nt]);
            tn_fence();
            constexpr int r1 = tn_rd1_at<P>(i), r0 = tn_rd0_at<P>(i), gd = tn_dma_at<P>(i);
            if constexpr (r1 >= 0) read_half(1, r1 >> 1, 1, r1 & 1);
            constexpr int gt = tn_tog_at<P>(i);
            if constexpr (gt >= 0) ra[gt] += flip;
            if constexpr (P::barB == i || P::barA == i) { tn_wait_lgkm(); tn_barrier(); }
#ifdef SF_EMU
            if constexpr (decltype(DO_DMA)::value && gd >= 0) dma(gd, t + 2);
#else
            if constexpr (decltype(DO_DMA)::value && gd >= 0) sf_buf_glds16_m0(gd < 8 ? dscA : dscB, voff[gd], 0u);
            constexpr int gm0 = tn_m0_at<P>(i);
            if constexpr (decltype(DO_DMA)::value && gm0 >= 0) sf_m0_set(dma_dst(gm0, t + 2));
#endif
            if constexpr (P::bar2 == i) {
                if constexpr (decltype(DO_DMA)::value) tn_wait_vm16(); else tn_wait_vm0();
                tn_barrier();
            }
            if constexpr (decltype(READ_NEXT)::value && r0 >= 0) read_half(0, r0 >> 1, 0, r0 & 1);
            tn_fence();
        });
#ifndef SF_EMU
        if constexpr (decltype(DO_DMA)::value) { dsc_advance(dscA, incA); dsc_advance(dscB, incB); }
#endif
    };

    bool pace = p.sync != nullptr && wave == 0;
    const TnPace pg = tn_pace_group(p.sync, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, p.ksplit);
    const int pmask = p.sync_every - 1;
    int t = 0;
    for (; t + 2 < nkt; ++t) {
        if (pace) {
            if ((t & pmask) == pmask - 8 && lane == 0) tn_pace_arrive(pg);
            if ((t & pmask) == pmask) {
                bool ok = true;
                if (lane == 0) ok = tn_pace_wait(pg, pg.members * (unsigned)((t >> __builtin_ctz((unsigned)p.sync_every)) + 1));
                pace = sf_all(ok);
            }
        }
        tilep(std::true_type{}, std::true_type{}, t);
    }
    if (t + 1 < nkt) { tilep(std::true_type{}, std::false_type{}, t); ++t; }
    tilep(std::false_type{}, std::false_type{}, t);

    // ---- epilogue: lane owns C[m][n..n+3]
    sf_mfma_drain();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sf_acc_touch(acc[i][j]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            sf_gemm_store4<OUT_F32, 0>(p.e, m0 + wr * 128 + i * 16 + (lane & 15), n0 + wc * 128 + j * 16 + 4 * (lane >> 4), v);
        }
}

// C = alpha * sum_y ws[y] (+ beta * C): deterministic fixed-order reduction of the split-K partials
template <int OUT_F32>
SF_GLOBAL void tn_splitk_reduce_kernel(const float* ws, int ksplit, void* C, long ldc, int M, int N, float alpha, float beta) {
    const long n4 = N / 4, total = (long)M * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        sf_v4f a = *reinterpret_cast<const sf_v4f*>(ws + m * N + n);
        for (int y = 1; y < ksplit; ++y) {
            const sf_v4f b = *reinterpret_cast<const sf_v4f*>(ws + (long)y * M * N + m * N + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += b[r];
        }
        if (OUT_F32) {
            float* c = (float*)C + m * ldc + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = alpha * a[r] + (beta != 0.f ? beta * c[r] : 0.f);
        } else {
            sf_bf16* c = (sf_bf16*)C + m * ldc + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = sf_f2bf(alpha * a[r] + (beta != 0.f ? beta * sf_bf2f(c[r]) : 0.f));
        }
    }
}

}  // namespace

#ifdef SF_EMU
#define SF_TN_SMEM(kernel)
#else
#define SF_TN_SMEM(kernel)                                                                                       \
    do {                                                                                                         \
        static bool done_ = false;                                                                               \
        if (!done_) {                                                                                            \
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes); \
            (void)hipGetLastError();                                                                             \
            done_ = true;                                                                                        \
        }                                                                                                        \
    } while (0)
#endif

extern "C" int sf_gemm_tn(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N,
                          int K, float alpha, float beta, float* workspace, long workspace_floats, int ksplit,
                          void* stream) {
    SF_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "sf_gemm_tn: negative shape");
    SF_CHECK_ARG(K % 64 == 0, "sf_gemm_tn: K must be a multiple of 64 (pad the contraction with zero rows)");
    SF_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "sf_gemm_tn: M, N, lda, ldb must be multiples of 8");
    SF_CHECK_ARG((M == 0 || M >= 8) && (N == 0 || N >= 8), "sf_gemm_tn: M, N >= 8");
    SF_CHECK_ARG(ldc % 4 == 0, "sf_gemm_tn: ldc must be a multiple of 4");
    SF_CHECK_ARG(c_dtype == SF_BF16 || c_dtype == SF_F32, "sf_gemm_tn: c_dtype");
    if (M == 0 || N == 0) return 0;
    if (K == 0) {
        SF_CHECK_ARG(false, "sf_gemm_tn: K == 0 is not supported");
    }
    GemmTnArgs p;
    p.A = (const sf_bf16*)A; p.lda = lda;
    p.B = (const sf_bf16*)B; p.ldb = ldb;
    p.e.C = C; p.e.ldc = ldc; p.e.R = nullptr; p.e.ldr = 0;
    p.e.Cadd = nullptr; p.e.ldadd = 0; p.e.add_S = 1; p.e.add_Spad = 1; p.e.add_off = 0;
    p.e.M = M; p.e.N = N; p.e.alpha = alpha; p.e.beta = beta;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + TM - 1) / TM;
    p.tiles_n = (N + TN - 1) / TN;
    p.gm = sf_knob("SF_GEMM_GM", 4);
    if (p.gm < 1) p.gm = 1;
    SF_CHECK_ARG(ksplit == 0 || ksplit == 1 || ksplit == 2, "sf_gemm_tn: ksplit must be 0 (automatic), 1 (never) or 2 (two-way)");
    const long nblk = (long)p.tiles_m * p.tiles_n;
    SF_CHECK_ARG(nblk < (1L << 31), "sf_gemm_tn: grid too large");
    // Split-K by 2 when the tile count leaves the last round of the 256 CUs at most half full (e.g. 384 or 896 tiles:
    // 1.5 / 3.5 rounds -> 3 / 7 full rounds of half-length blocks) and the caller provided room for the fp32 partials.
    p.ksplit = 1;
    p.ws = nullptr;
    {
        const long rem = nblk % 256;
        const bool helps = rem > 0 && rem <= 128 && nblk < 2048 && ((2 * nblk) % 256 == 0 || (2 * nblk) % 256 > 192);
        const bool fits = workspace && workspace_floats >= 2L * M * N && (K / TK) % 2 == 0 && K >= 4096 && N % 4 == 0;
        SF_CHECK_ARG(ksplit != 2 || fits, "sf_gemm_tn: ksplit = 2 needs a workspace of 2*M*N floats, K >= 4096 with an even "
                                          "number of 64-row K-tiles, and N % 4 == 0");
        if (fits && ksplit != 1 && (helps || ksplit == 2)) { p.ksplit = 2; p.ws = workspace; }
    }
    // pace-keeping counters: the last kTnSyncFloats floats of the workspace (when it is large enough), zeroed per launch
    p.sync = nullptr;
    p.sync_every = sf_knob("SF_GEMM_TN_SYNC", 128);
#ifndef SF_EMU
    {
        const long need = (p.ksplit > 1 ? 2L * M * N : 0L) + kTnSyncFloats;
        const bool pow2 = p.sync_every >= 16 && (p.sync_every & (p.sync_every - 1)) == 0;
        // counters used = 8 XCDs x groups per XCD (tn_pace_group): they must fit the tail (a grid of > ~130 k tiles would not)
        const long groups_per_xcd = ((long)p.ksplit * (nblk / 8 + 1) + 31) / 32;
        if (workspace && workspace_floats >= need && pow2 && K / TK / p.ksplit >= 2 * p.sync_every && nblk >= 64 &&
            (p.ksplit == 1 || nblk % 8 == 0) && 8 * groups_per_xcd <= kTnSyncFloats) {
            p.sync = reinterpret_cast<unsigned*>(workspace + workspace_floats - kTnSyncFloats);
            if (hipMemsetAsync(p.sync, 0, kTnSyncFloats * sizeof(float), (hipStream_t)stream) != hipSuccess) p.sync = nullptr;
        }
    }
#endif
    // plan: which operand's LDS halves are released / re-staged first (0 = B, 1 = A).  Measured on the step's shapes
    // (profiles/old/r2_gemm_tn_ab.jsonl): A first wins for wide X (down-proj, N = 14336), B first elsewhere; both beat the
    // round-1 schedule (tools build: SF_GEMM_TN_PLAN=-1) by 3-8 %.
    const int plan = sf_knob("SF_GEMM_TN_PLAN", N > 8192 ? 1 : 0);
    const bool f32_main = p.ksplit > 1 || c_dtype == SF_F32;
    const dim3 grid((unsigned)nblk, (unsigned)p.ksplit);
#define SF_TN_CASE(F32, AF)                                                                                           \
    if (f32_main == (F32 != 0) && plan == AF) {                                                                      \
        SF_TN_SMEM((gemm_tn_256w4p_kernel<F32, AF>));                                                                \
        SF_LAUNCH((gemm_tn_256w4p_kernel<F32, AF>), grid, dim3(256), 2 * kBufBytes, stream, p);                      \
    } else
    SF_TN_CASE(0, 0) SF_TN_CASE(1, 0) SF_TN_CASE(0, 1) SF_TN_CASE(1, 1)
#undef SF_TN_CASE
    {
#ifdef SF_ABLATE
        if (f32_main) {
            SF_TN_SMEM((gemm_tn_256w4_kernel<1>));
            SF_LAUNCH((gemm_tn_256w4_kernel<1>), grid, dim3(256), 2 * kBufBytes, stream, p);
        } else {
            SF_TN_SMEM((gemm_tn_256w4_kernel<0>));
            SF_LAUNCH((gemm_tn_256w4_kernel<0>), grid, dim3(256), 2 * kBufBytes, stream, p);
        }
#else
        SF_CHECK_ARG(false, "sf_gemm_tn: no kernel for this configuration");
#endif
    }
    if (p.ksplit > 1) {
        const long units = (long)M * (N / 4);
        const int rgrid = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
        if (c_dtype == SF_F32)
            SF_LAUNCH((tn_splitk_reduce_kernel<1>), dim3(rgrid), dim3(256), 0, stream, (const float*)p.ws, p.ksplit, C, ldc, M, N, alpha, beta);
        else
            SF_LAUNCH((tn_splitk_reduce_kernel<0>), dim3(rgrid), dim3(256), 0, stream, (const float*)p.ws, p.ksplit, C, ldc, M, N, alpha, beta);
        return sf_check_launch("sf_gemm_tn(split-K)");
    }
    return sf_check_launch("sf_gemm_tn");
}
