// 256x256x64 bf16 MFMA GEMM (NT form), 4 waves x (128 x 128) -- one wave per SIMD, software-pipelined in ONE
// instruction stream per wave, every non-MFMA instruction placed in its own MFMA slot by a compile-time plan.
//
//   * 256 threads; wave (wr, wc) of a 2 x 2 grid owns a 128 x 128 output block = 8 x 8 mfma_f32_16x16x32_bf16 tiles
//     (256 accumulator registers pinned to AGPRs; one wave per SIMD with the full 512-entry register file).
//   * LDS = 2 K-tile buffers x {A 256 rows, B 256 rows} x 128 B, XOR-swizzled 16-byte chunks = 128 KiB, filled by
//     LDS-DMA (buffer_load ... lds through raw buffer descriptors: the K advance is a scalar offset, no vector
//     arithmetic in the loop), + 17 KiB staging for the bf16 epilogue (the tile leaves as whole 128-byte lines).
//   * persistent over tiles: one workgroup per CU walks tiles w, w + grid, ...; the next tile's first two K-tiles are
//     put in flight before the finished tile's epilogue (see the hand-over section of the kernel).
//   * a K-tile = 128 MFMAs = two k-halves of 64; fragments are double-buffered in registers (set 0 = k-half 0,
//     set 1 = k-half 1).
//   * Schedule of iteration t (round 2; measured against hipBLASLt's hand-scheduled kernel of the same geometry, whose
//     main loop issues exactly one filler per MFMA gap and is ~99 % matrix-pipe-bound in cycles):
//       slots   0..45  : the 16 fragment reads of (t, k-half 1), one operand at a time, one read per >= 2 MFMAs
//       slot   21 / 51 : lgkmcnt(0) + s_barrier -> that OPERAND's half of buffer t&1 is free (released separately,
//                        so its re-staging starts after 1/6 of the iteration instead of 1/2)
//       slots  22..95  : the 16 LDS-DMA pieces of tile t+2, never two fillers in one slot (an LDS-DMA issue costs
//                        60-180 cycles when bunched with ds_reads; a burst of one per 2 MFMAs measured -7 %); the
//                        LDS destination (M0) of each piece is written in an EARLIER free slot, so the piece itself
//                        is one instruction (s_mov m0 + s_nop + buffer_load in one gap: ~10 cycles per piece)
//       slot  108      : s_waitcnt vmcnt(16) + s_barrier -> tile t+1 (issued during iteration t-1) is visible; this
//                        iteration's 16 pieces stay in flight.  DMA lead: 1.1 .. 1.7 iterations (2400 .. 3700 cycles).
//       slots 110..126 : the 16 fragment reads of (t+1, k-half 0)
//     Two plans differ in which operand goes first: B first is faster when N is narrow (<= 8192: +4..6 % over A
//     first), A first when N is wide (+2..4 %) -- measured, profiles/old/r2_gemm_ab.jsonl; the launcher picks by N.
//     RAW: a tile is read only after (own pieces landed: counted vmcnt) + barrier.  WAR: an operand half is re-staged
//     only after (own reads returned: lgkmcnt(0)) + barrier; its k-half-0 fragments were read in iteration t-1.
//   * round-1 schedule (one barrier per K-tile, groups of 4 MFMAs + 2 reads, DMA in the second half-step, vmcnt(0)):
//     30 % more cycles than hipBLASLt on the same shape; this one 10 % (SQ_WAVE_CYCLES, profiles/old/r2_gemm_pmc.txt).
//     Cycles per K-tile read inside the kernel (tools build, SF_GEMM_CYC): MFMAs alone 2083, this loop 2251
//     (profiles/old/r2_gemm_cycles.jsonl).
//     It and the intermediate plans live in tools/experiments/sf_gemm256w4_sched.inc (tools build only).
#pragma once
#include "sf_api_internal.h"
#include "sf_util.h"
#include "sf_gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

#define SF_INLINE_LAMBDA __attribute__((always_inline))
#ifndef SF_W4_FAST_EPI
#define SF_W4_FAST_EPI 1
#endif

#ifdef SF_EMU
#define SF_W4_SMEM(kernel, ...)
#else
#define SF_W4_SMEM(kernel, ...)                                                                                  \
    do {                                                                                                         \
        static bool done_ = false;                                                                               \
        if (!done_) {                                                                                            \
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (kW4SmemBytes, ##__VA_ARGS__)); \
            (void)hipGetLastError();                                                                             \
            done_ = true;                                                                                        \
        }                                                                                                        \
    } while (0)
#endif


// (has linkage: the per-instantiation launch functions take it across translation units)
struct GemmW4Args {
    const sf_bf16* A; long lda;
    const sf_bf16* B; long ldb;
    SfGemmEpi e;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;
    // split-K (round 4; fp32 plain form only): work unit u = ks * tiles + tile computes K-range [ks * K, (ks + 1) * K) of the operands
    // (K here = the chunk length; ks_a / ks_b = element offsets of a chunk in A / B rows) into partial ks of C (ks_c floats apart)
    int ksplit = 1;
    long ks_a = 0, ks_b = 0, ks_c = 0;
    int k_total = 0;      // split-K: the whole contraction length (the last chunk may be shorter than K); 0 = every chunk is K long
    // ragged M (round 4; real batches): 1 = the last row tile starts at row M - 256 instead of tiles_m * 256 - 256 -- it overlaps its
    // neighbour, every tile is whole (fast / fused epilogues, no bounds), and the rows computed twice get the same bits twice (a row's
    // accumulation order does not depend on the tile it is in).  Set by the launcher for M >= 256, beta == 0, no in-place residual.
    int mshift = 0;
    // round 6: 1 = whole tiles of the bf16 residual form transpose in REGISTERS (v_permlane16_swap of packed column-block pairs: a lane ends
    // up with 8 consecutive columns = 16 bytes of residual in, 16 bytes out) instead of taking the general store; 0 (tools build A/B) = as before
    int epi_direct = 0;
#ifdef SF_ABLATE
    int stagger;
    int cyc;   // tools build: wave 0 of every workgroup overwrites C[m0][n0..n0+1] with its K-loop cycle count (fp32 bits)
#endif
};

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kOpBytes = 256 * TK * 2;       // 32 KiB: one operand's K-tile
constexpr int kBufBytes = 2 * kOpBytes;      // A + B
constexpr int kStageRow = 272;               // epilogue staging: 256 B of a 128-column bf16 row + 16 B pad (bank spread)
constexpr int kStageBytes = 4 * 16 * kStageRow;  // 4 waves x 16 rows
constexpr int kW4SmemBytes = 2 * kBufBytes + kStageBytes;   // two K-tile buffers + the staging area = 145 KiB of the CU's 160
// fused SwiGLU forward (ADD = 3): a staged row holds the wave's 64 gate, 64 up and 64 act values (3 x 128 B) + 16 B pad
constexpr int kStageRow3 = 400;
constexpr int kW4SmemBytes3 = 2 * kBufBytes + 4 * 16 * kStageRow3;   // 153 KiB
template <int ADD> constexpr int w4_smem_bytes() { return ADD == 3 ? kW4SmemBytes3 : kW4SmemBytes; }


#ifdef SF_EMU
SF_DEVICE void w4_barrier() { sfemu::block_barrier(); }
SF_DEVICE void w4_wait_all() {}
SF_DEVICE void w4_wait_lgkm() {}
SF_DEVICE void w4_wait_vm16() {}
SF_DEVICE void w4_wait_vm13() {}
SF_DEVICE void w4_wait_vm32() {}
SF_DEVICE void w4_wait_vm63() {}
SF_DEVICE void w4_wait_vm0() {}
SF_DEVICE void w4_fence() {}
#else
SF_DEVICE void w4_barrier() { __builtin_amdgcn_s_barrier(); }
SF_DEVICE void w4_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void w4_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
SF_DEVICE void w4_wait_vm16() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }   // all but the newest 16 LDS-DMA pieces
SF_DEVICE void w4_wait_vm13() { asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }
SF_DEVICE void w4_wait_vm32() { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
SF_DEVICE void w4_wait_vm63() { asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); }
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

// ---- slot plans of the fine-grained schedules: which of a K-tile's 128 MFMA slots each non-MFMA instruction follows.
// A slot holds at most one filler (the MI355X notes: a 16x16x32 MFMA leaves room for ~2 single-issue instructions in
// its shadow; two ds_read_b128 or a DMA with its address arithmetic in ONE gap overflow it and idle the matrix pipe).
//   rd1[r] : read of fragment r of (tile t, k-half 1)      -> register set 1   (set 1 is consumed by MFMAs 64..127)
//   dma[g] : LDS-DMA piece g of tile t+2                    -> buffer t&1
//   rd0[r] : read of fragment r of (tile t+1, k-half 0)     -> register set 0   (set 0 is consumed by MFMAs 0..63)
//   bar1   : lgkmcnt(0) + s_barrier after this slot  (every wave's reads of buffer t&1 returned: it may be re-staged)
//   bar2   : vmcnt + s_barrier after this slot       (every wave's pieces of tile t+1 landed: it may be read)
// compile-time loop: the body receives std::integral_constant<int, I> (a 128-trip `#pragma unroll` loop with nested
// conditionals is not reliably unrolled, and runtime-indexed fragment / accumulator arrays would go to scratch)
// slot -> filler lookups, evaluated by the constant evaluator (no template instantiation per candidate)
template <class P> constexpr int w4_rd1_at(int i) { for (int r = 0; r < 16; ++r) if (P::rd1(r) == i) return r; return -1; }
template <class P> constexpr int w4_rd0_at(int i) { for (int r = 0; r < 16; ++r) if (P::rd0(r) == i) return r; return -1; }
template <class P> constexpr int w4_dma_at(int i) { for (int g = 0; g < 16; ++g) if (P::dma(g) == i) return g; return -1; }

template <int I, int N, class F>
SF_DEVICE void w4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w4_static_for<I + 1, N>(f);
    }
}

template <class P, class = void> struct w4_has_split : std::false_type {};
template <class P> struct w4_has_split<P, std::void_t<decltype(P::barB)>> : std::true_type {
    static constexpr bool barB(int i) { return P::barB == i; }
    static constexpr bool barA(int i) { return P::barA == i; }
};

// M0 (the LDS destination of a DMA piece) is written in a slot of its own, after the previous piece's slot and before the
// piece's own: the first such slot that holds no other filler; when there is none it shares the previous DMA's slot
// (issued after it).  hipBLASLt's loop does the same (buffer_load ... lds ; next gap: s_add m0): a DMA slot that also
// carries the M0 write and the s_nop its hazard needs overflows the MFMA shadow (measured: ~10 cycles per piece).
template <class P> constexpr bool w4_slot_busy(int i) {
    if (w4_rd1_at<P>(i) >= 0 || w4_rd0_at<P>(i) >= 0 || w4_dma_at<P>(i) >= 0 || i == P::bar1 || i == P::bar2) return true;
    if constexpr (w4_has_split<P>::value) { if (i == P::barB || i == P::barA) return true; }
    return false;
}
template <class P> constexpr int w4_dma_prev_slot(int g) {
    int best = -1;
    for (int h = 0; h < 16; ++h) if (P::dma(h) < P::dma(g) && P::dma(h) > best) best = P::dma(h);
    return best;
}
template <class P> constexpr int w4_m0_slot(int g) {
    const int lo = w4_dma_prev_slot<P>(g);
    for (int s = lo + 1; s < P::dma(g); ++s) if (!w4_slot_busy<P>(s)) return s;
    return lo >= 0 ? lo : 0;
}
template <class P> constexpr int w4_m0_at(int i) { for (int g = 0; g < 16; ++g) if (w4_m0_slot<P>(g) == i) return g; return -1; }

// Plan interface: the MFMA slot (0..127) after which each filler of iteration t is issued.
//   rd1(r) : read of fragment r (0..7 = B n-tiles, 8..15 = A m-tiles) of (tile t, k-half 1)  -> register set 1
//   dma(g) : LDS-DMA piece g (0..7 = A rows, 8..15 = B rows) of tile t+2                     -> buffer t&1
//   rd0(r) : read of fragment r of (tile t+1, k-half 0)                                       -> register set 0
//   barB / barA : lgkmcnt(0) + s_barrier after these slots (first / second operand half released)
//   bar2   : counted vmcnt + s_barrier after this slot (tile t+1 published); vm = pieces of THIS iteration in flight then
struct W4PlanBFirst {   // B (weights) released and re-staged first: faster for narrow N
    static constexpr int barB = 21, barA = 51, bar1 = -1, bar2 = 108, vm = 16;
    static constexpr int rd1(int r) {
        constexpr int s[16] = {0, 2, 4, 6, 8, 10, 12, 14, 24, 27, 30, 33, 36, 39, 42, 45};
        return s[r];
    }
    static constexpr int dma(int g) {
        constexpr int b[8] = {22, 26, 29, 32, 35, 38, 41, 44};
        constexpr int a[8] = {53, 59, 65, 71, 77, 83, 89, 95};
        return g < 8 ? a[g] : b[g - 8];
    }
    static constexpr int rd0(int r) { return 110 + r + (r >= 8 ? 1 : 0); }   // 110..117, 119..126
};
struct W4PlanAFirst {   // A (activations) first: faster for wide N
    static constexpr int barB = 21, barA = 51, bar1 = -1, bar2 = 108, vm = 16;
    static constexpr int rd1(int r) {
        constexpr int s[16] = {24, 27, 30, 33, 36, 39, 42, 45, 0, 2, 4, 6, 8, 10, 12, 14};
        return s[r];
    }
    static constexpr int dma(int g) {
        constexpr int first[8] = {22, 26, 29, 32, 35, 38, 41, 44};
        constexpr int second[8] = {53, 59, 65, 71, 77, 83, 89, 95};
        return g < 8 ? first[g] : second[g - 8];
    }
    static constexpr int rd0(int r) { return 110 + r + (r >= 8 ? 1 : 0); }
};
#ifdef SF_ABLATE
#include "../../tools/experiments/sf_gemm256w4_sched.inc"
#else
template <int SCHED> using W4PlanFor = std::conditional_t<SCHED == 13, W4PlanAFirst, W4PlanBFirst>;
#endif

// SCHED: 12 = W4PlanBFirst, 13 = W4PlanAFirst (product); 0 and 2..11 exist in the tools build only.
// ABL (tools build, timing ablations of the round-1 schedule only, results are wrong): bit0 = no ds_reads after the first
// tile, bit1 = no DMA in the loop, ...
// ADD: 0 = plain, 1 = + fp32 row-mapped addend in the epilogue, before the single bf16 rounding (sf_gemm_nt_rowadd), 2 = d(SwiGLU) in the bf16 epilogue
// (sf_gemm_nt_swiglu_bwd; whole tiles only -- its launcher guarantees it), 3 = SwiGLU forward in the bf16 epilogue of the fused
// gate|up projection (sf_gemm_nt_swiglu_fwd; whole tiles only), 4 = teacher head: column tiles from red_n0 on are reduced (row max,
// sum-exp, argmax per 128-column block) instead of stored (sf_gemm_nt_teacher); 3 in detail: B = [gate rows ; up rows] of the fused weight, N = 2 I; tile tn takes
// the gate AND the up rows of act columns tn*128 .. +127, interleaved 16 / 16 along its 256 B rows, so that a lane's accumulators
// [i][2 jj] / [i][2 jj + 1] are gate / up of the SAME 4 columns: gate|up leave in their natural [M, 2I] layout (the backward reads
// them), act = round(silu(gate)) * up leaves beside them, and the separate pass over gate|up (0.94 GB read per call) is gone
template <int OUT_F32, int ADD = 0, int SCHED = 12, int ABL = 0>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) gemm_nt_256w4_kernel(GemmW4Args p) {
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id();
    const int wr = wave >> 1, wc = wave & 1;
    // PERSISTENT over tiles: workgroup w computes tiles w, w + gridDim.x, ... (the launcher starts one workgroup per CU
    // when there are more tiles than CUs; gridDim.x is a multiple of 8 then, so a tile keeps its XCD).  The next tile's
    // first two K-tiles are put in flight BEFORE the epilogue of the finished one: its DMA latency, the per-tile set-up and
    // the workgroup hand-over (~3 us of a ~98 us tile at K = 4096) disappear behind the epilogue's stores.
    const int nblk = p.tiles_m * p.tiles_n;
    const int nunits = nblk * p.ksplit;           // work units of the walk (= tiles unless split-K)
    int tile = (int)blockIdx.x;
    int m0 = 0, n0 = 0, ks0 = 0;
    int nkt = p.K / TK;                           // K-tiles of the current work unit (split-K: the last chunk may be shorter)

    // ---- DMA sources: this wave stages pieces 8*wave .. 8*wave+7 (8 rows x 128 B each) of A and of B through raw
    // buffer descriptors rooted at the tile origin: one 32-bit byte offset per piece and lane (constant over the K
    // loop), the K advance in the scalar offset.  Rows past the matrix edge re-read the last valid row: they only feed
    // accumulators that are never stored.
    const int srow = lane >> 3;
    const int slc = (lane & 7) ^ (srow & 7);  // logical 16-byte chunk fetched into physical chunk lane&7
    SfBuf bufA, bufB;
#ifndef SF_EMU
    SfBufRaw rawA, rawB;   // the same descriptors for the asm (compiler-opaque) DMA forms
#endif
    unsigned voff[16];
    auto setup_tile = [&](int t) {   // tile origin, descriptors and per-lane source offsets of work unit t
        int tm, tn;
        ks0 = 0;
        if (p.ksplit > 1) {                                             // (split-K: the unit's K chunk)
            ks0 = t / nblk;
            t -= ks0 * nblk;
            if (p.k_total) {
                const int left = p.k_total - ks0 * p.K;
                nkt = (left < p.K ? left : p.K) / TK;
            }
        }
        w4_tile_coords(t, nblk, p.tiles_m, p.tiles_n, p.gm, tm, tn);
        m0 = tm * TM;
        if (p.mshift && m0 + TM > p.M) m0 = p.M - TM;
        n0 = tn * TN;
        const sf_bf16* Ab = p.A + (long)m0 * p.lda + ks0 * p.ks_a;
        const long brow0 = ADD == 3 ? (long)tn * (TN / 2) : (long)n0;   // (ADD = 3: the tile's first gate row)
        const sf_bf16* Bb = p.B + brow0 * p.ldb + ks0 * p.ks_b;
        bufA = sf_make_buf(Ab, 0x7fffffffu);
        bufB = sf_make_buf(Bb, 0x7fffffffu);
#ifndef SF_EMU
        rawA = sf_make_buf_raw(Ab);
        rawB = sf_make_buf_raw(Bb);
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int ra = (8 * wave + j) * 8 + srow, rb = ra;
            ra = m0 + ra < p.M ? ra : p.M - 1 - m0;
            if constexpr (ADD == 3) {
                // tile row rb = wave column block (rb >> 7) | pair (2 bits) | up? (1 bit) | column in the 16-group (4 bits)
                const int w7 = rb & 127;
                rb = ((w7 >> 4) & 1) * (p.N >> 1) + (rb >> 7) * 64 + (w7 >> 5) * 16 + (w7 & 15);
            } else {
                rb = n0 + rb < p.N ? rb : p.N - 1 - n0;
            }
            voff[j] = (unsigned)(((long)ra * p.lda + slc * 8) * 2);
            voff[8 + j] = (unsigned)(((long)rb * p.ldb + slc * 8) * 2);
        }
    };
    setup_tile(tile);
    auto dma_dst = [&](int g, int kt) -> char* {
        return smem + (kt & 1) * kBufBytes + (g >> 3) * kOpBytes + (8 * wave + (g & 7)) * 1024;
    };
    auto dma = [&](int g, int kt) {  // piece g (0..7 A, 8..15 B) of K-tile kt into buffer kt&1
        sf_buf_glds16(g < 8 ? bufA : bufB, voff[g], (unsigned)kt * (TK * 2), dma_dst(g, kt));
    };
    // between tiles: invisible to the compiler like the loop's DMAs (a builtin DMA would make it wait for vmcnt(0) before
    // the epilogue's staging reads, which it cannot tell apart from the K-tile buffers)
    auto dma_next = [&](int g, int kt) {
#ifdef SF_EMU
        dma(g, kt);
#else
        sf_buf_glds16_opaque(g < 8 ? rawA : rawB, voff[g], (unsigned)kt * (TK * 2), dma_dst(g, kt));
#endif
    };

    // ---- fragment read offsets; (row & 7) == (lane & 7) for every fragment row
    const int frow = lane & 15;
    int swz[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) swz[ks] = ((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
    const int a_off = (wr * 128 + frow) * 128;
    const int b_off = kOpBytes + (wc * 128 + frow) * 128;

    sf_v4f acc[8][8];
    auto acc_init = [&]() SF_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = sf_v4f{0.f, 0.f, 0.f, 0.f};
    };
    acc_init();
    sf_v8s f[2][16];  // [set][0..7 = B n-tiles, 8..15 = A m-tiles]
#ifdef SF_ABLATE
    float cyc_loop = 0.f;
#ifndef SF_EMU
    unsigned rt[4] = {(unsigned)__builtin_amdgcn_s_memrealtime(), 0u, 0u, 0u};   // 100 MHz: entry, loop begin, loop end, exit
    if (p.stagger && blockIdx.x < 256u) {   // first-round workgroups start late by (rank in the stagger order) * stagger * 10 ns
        const unsigned rank = p.stagger > 0 ? (blockIdx.x & 7u) : (blockIdx.x >> 3);   // > 0: by XCD; < 0: by slot within the XCD
        const unsigned wait = rank * (unsigned)(p.stagger > 0 ? p.stagger : -p.stagger);
        while ((unsigned)__builtin_amdgcn_s_memrealtime() - rt[0] < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
#endif

    auto read_frag = [&](int set, int g, const char* buf, int ks) {
        if (g < 8) f[set][g] = *reinterpret_cast<const sf_v8s*>(buf + b_off + g * 2048 + swz[ks]);
        else f[set][g] = *reinterpret_cast<const sf_v8s*>(buf + a_off + (g - 8) * 2048 + swz[ks]);
    };

    // ---- prologue: K-tiles 0 and 1 staged, fragments of k-half 0 of tile 0 in registers
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

    // ---- one K-tile by plan P: READ_NEXT = tile t+1 exists, DO_DMA = tile t+2 exists
    auto tilep = [&](auto PLAN, auto READ_NEXT, auto DO_DMA, int t) {
        using P = decltype(PLAN);
        const char* cur = smem + (t & 1) * kBufBytes;
        const char* nxt = smem + ((t + 1) & 1) * kBufBytes;
        w4_static_for<0, 128>([&](auto I) SF_INLINE_LAMBDA {
            constexpr int i = decltype(I)::value, idx = i & 63, mt = idx >> 3, nt = idx & 7;
            // accumulators pinned to AGPRs (asm form): the builtin's allocation does not survive this interleave
            if constexpr (i < 64) sf_mfma16_acc(f[0][nt], f[0][8 + mt], acc[mt][nt]);
            else sf_mfma16_acc(f[1][nt], f[1][8 + mt], acc[mt][nt]);
            w4_fence();
            constexpr int r1 = w4_rd1_at<P>(i), r0 = w4_rd0_at<P>(i), gd = w4_dma_at<P>(i);
            // ABL (tools build, timing only, results are wrong): 1 = no fragment reads, 2 = no DMA, 4 = no barriers, 8 = no waits
            constexpr bool kRd = !(ABL & 1), kDma = !(ABL & 2), kBar = !(ABL & 4), kWait = !(ABL & 8);
            if constexpr (kRd && r1 >= 0) read_frag(1, r1, cur, 1);
            if constexpr (P::bar1 == i && P::bar1 != P::bar2) {
                if constexpr (kWait) w4_wait_lgkm();
                if constexpr (kBar) w4_barrier();
            }
            if constexpr (w4_has_split<P>::value) {
                if constexpr (w4_has_split<P>::barB(i) || w4_has_split<P>::barA(i)) {
                    if constexpr (kWait) w4_wait_lgkm();
                    if constexpr (kBar) w4_barrier();
                }
            }
#ifdef SF_EMU
            if constexpr (kDma && decltype(DO_DMA)::value && gd >= 0) dma(gd, t + 2);
#else
            if constexpr (kDma && decltype(DO_DMA)::value && gd >= 0)
                sf_buf_glds16_m0(gd < 8 ? rawA : rawB, voff[gd], (unsigned)(t + 2) * (TK * 2));
            constexpr int gm0 = w4_m0_at<P>(i);
            if constexpr (kDma && decltype(DO_DMA)::value && gm0 >= 0) sf_m0_set(dma_dst(gm0, t + 2));
#endif
            if constexpr (P::bar2 == i) {
                if constexpr (kWait) {
                    if constexpr (P::bar1 == P::bar2) w4_wait_all();
                    else if constexpr (decltype(DO_DMA)::value && P::vm == 16) w4_wait_vm16();
                    else w4_wait_vm0();
                }
                if constexpr (kBar) w4_barrier();
            }
            if constexpr (kRd && decltype(READ_NEXT)::value && r0 >= 0) read_frag(0, r0, nxt, 0);
            w4_fence();
        });
    };

    for (;;) {   // ---- tiles of this workgroup
#ifdef SF_ABLATE
    if constexpr (SCHED == 0) {
#include "../../tools/experiments/sf_gemm256w4_sched0.inc"
    } else
#endif
    {
        using P = W4PlanFor<SCHED>;
        int t = 0;
#if defined(SF_ABLATE) && !defined(SF_EMU)
        const unsigned long long cyc0 = __builtin_readcyclecounter();
        rt[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
        for (; t + 2 < nkt; ++t) tilep(P{}, std::true_type{}, std::true_type{}, t);
#if defined(SF_ABLATE) && !defined(SF_EMU)
        cyc_loop = (float)(__builtin_readcyclecounter() - cyc0);
        rt[2] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
        if (t + 1 < nkt) { tilep(P{}, std::true_type{}, std::false_type{}, t); ++t; }
        tilep(P{}, std::false_type{}, std::false_type{}, t);
        sf_mfma_drain();   // asm MFMAs are invisible to the hazard recogniser: let the last ones retire before acc is read
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sf_acc_touch(acc[i][j]);
    }

    // ---- hand-over: the finished tile's coordinates, then the next tile's first two K-tiles go in flight
    const int mc = m0, nc = n0;
    const long coff = ks0 * p.ks_c;              // (split-K: this unit's partial of C; 0 otherwise)
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < nunits;
    // the interior-tile fast path of the epilogue (workgroup-uniform; 16-byte row segments need 16-byte aligned rows)
    // (split-K partials live in a workspace laid out in WHOLE row tiles: an edge tile stores all 256 rows there -- the rows past M come
    //  from re-read operand rows and are never read back)
    const bool direct_r = !OUT_F32 && ADD == 0 && p.epi_direct && p.e.R && (p.e.ldr & 7) == 0 && ((size_t)p.e.R & 15) == 0;   // residual: direct form only
    const bool fast = SF_W4_FAST_EPI && (mc + TM <= p.M || p.ksplit > 1) && nc + TN <= p.N && p.e.beta == 0.f && (!p.e.R || direct_r) && (p.e.ldc & 7) == 0 &&
                      ((size_t)p.e.C & 15) == 0;
    // Row-addend form (round 3): the fp32 addend joins the accumulator in the EPILOGUE, before the single bf16 rounding.  Round 2
    // started the accumulators from it: 256 KiB of loads per tile in front of the first MFMA, every workgroup of a round at
    // once (64 MiB per round at the fabric's speed: ~0.1 ms of a 0.67 ms launch, 0.82 of hipBLASLt), and no persistent walk
    // (a second call site of that 256-register load made the compiler spill).  Here the rows of m-tile i + 2 are loaded while
    // m-tile i is packed and stored -- the epilogue is bound by the CU's store path, the loads ride under it -- and the first
    // two m-tiles' rows are requested BEFORE the next tile's DMA pieces (vmcnt is in-order: a wait for an addend row younger
    // than the DMAs would drain them).  The fragment registers are dead here, so the 64 registers are free.
    sf_v4f ad[2][8];
    const int arow_l = mc + wr * 128 + (lane & 15), acol = nc + wc * 128 + 4 * (lane >> 4);
    auto addend_rows = [&](int i, sf_v4f (&dst)[8]) SF_INLINE_LAMBDA {
        const int m = arow_l + i * 16;
        const int bb = m / p.e.add_S;
        const float* a = p.e.Cadd + ((long)bb * p.e.add_Spad + (m - bb * p.e.add_S) + p.e.add_off) * p.e.ldadd + acol;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const sf_v4f*>(a + j * 16);
    };
    if constexpr (ADD == 1) {
        if (fast) { addend_rows(0, ad[0]); addend_rows(1, ad[1]); }
    }
    w4_wait_lgkm();
    w4_barrier();                 // every wave's last fragment reads returned: both K-tile buffers are free
    if (has_next) {
        setup_tile(next);
#pragma unroll
        for (int g = 0; g < 16; ++g) dma_next(g, 0);
        if (nkt > 1) {
#pragma unroll
            for (int g = 0; g < 16; ++g) dma_next(g, 1);
        }
    }

    // ---- epilogue of tile (mc, nc): lane owns C[m][n..n+3]
    // Interior tiles without beta / residual (every launch of the training step) take a straight path; the general store
    // below re-derives address, bounds and the beta / residual cases per 4 values (~7500 instructions).
    int newer = 0;   // vector-memory instructions this wave issues AFTER the next tile's DMAs (lower bound; 0 = unknown)
    bool reduced = false;
    if constexpr (ADD == 4) {
        // Teacher head (eagle3/model.py:487-501 needs, of the [rows, Vt] logits, the row maximum, its sum-exp, the argmax and the
        // draft sub-vocabulary's logits): the columns from red_n0 on are only REDUCED.  A lane holds, per 16-row m-tile, 32 logits of
        // ONE row (8 n-tiles x 4 columns); the 4 lanes of a row (lane >> 4) are merged with two xor steps.  The logits are rounded
        // to bf16 first: TargetHead is a bf16 module, its argmax near-ties are reference semantics.  The first column wins ties
        // (the permuted head keeps original order inside this range; across blocks sf_teacher_reduce_perm compares original ids).
        if (nc >= p.e.red_n0) {
            reduced = true;
            const int r = lane & 15, q = lane >> 4;
            const int cw = nc + wc * 128;                         // first column of this wave's block
            const int blk = (cw - p.e.red_n0) >> 7;
            if (cw < p.N) {
                auto reduce_block = [&](auto EDGE) SF_INLINE_LAMBDA {
                    constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v[8][4];
                        float lm = -__builtin_inff();
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float t = sf_round_bf(acc[i][j][e]);
                                if constexpr (decltype(EDGE)::value) {     // the block reaches past the last column (Vt % 128 != 0)
                                    if (cw + j * 16 + 4 * q + e >= p.N) t = -__builtin_inff();
                                }
                                v[j][e] = t;
                                lm = fmaxf(lm, t);
                            }
                        int lc = 0;
#pragma unroll
                        for (int j = 7; j >= 0; --j)
#pragma unroll
                            for (int e = 3; e >= 0; --e) lc = (v[j][e] == lm) ? j * 16 + e : lc;
                        lc += cw + 4 * q;
                        auto merge = [&](auto MK) SF_INLINE_LAMBDA {   // both lanes of a pair see the same (a, b): same winner
                            float ma, mb;
                            int ca, cb;
                            sf_xor_pair<decltype(MK)::value>(lm, ma, mb);
                            sf_xor_pair<decltype(MK)::value>(lc, ca, cb);
                            const bool tb = (mb > ma) | ((mb == ma) & (cb < ca));    // (no short-circuit: branch-free)
                            lm = tb ? mb : ma;
                            lc = tb ? cb : ca;
                        };
                        merge(std::integral_constant<int, 16>{});
                        merge(std::integral_constant<int, 32>{});
                        const float nb = -lm * kLog2e;      // exp(v - max) = exp2(v * log2e - max * log2e): one fma + v_exp_f32
                        float se = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) se += sf_exp2_raw(__builtin_fmaf(v[j][e], kLog2e, nb));
                        {
                            float sa, sb;
                            sf_xor_pair<16>(se, sa, sb);
                            se = sa + sb;
                            sf_xor_pair<32>(se, sa, sb);
                            se = sa + sb;
                        }
                        const int row = mc + wr * 128 + i * 16 + r;
                        if (q == 0 && row < p.M)
                            *reinterpret_cast<sf_v4f*>(p.e.red_part + ((long)row * p.e.red_stride + blk) * 4) = sf_v4f{lm, se, (float)lc, 0.f};
                    }
                };
                if (cw + 128 > p.N) reduce_block(std::true_type{});
                else reduce_block(std::false_type{});
            }
            newer = 0;
        }
    }
    if (reduced) {
    } else
#ifdef SF_ABLATE
    if ((p.cyc & 4) || ((p.cyc & 8) && (blockIdx.x & 1))) {   // timing experiments: no stores at all / only every other workgroup stores
    } else
#endif
    if (fast) {
        const float alpha = p.e.alpha;
        if constexpr (OUT_F32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float* crow = (float*)p.e.C + coff + (long)(mc + wr * 128 + i * 16 + (lane & 15)) * p.e.ldc + nc + wc * 128 + 4 * (lane >> 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (ADD == 1) *reinterpret_cast<sf_v4f*>(crow + j * 16) = acc[i][j] + ad[i & 1][j];   // (alpha == 1)
                    else *reinterpret_cast<sf_v4f*>(crow + j * 16) = acc[i][j] * alpha;
                }
                if constexpr (ADD == 1) { if (i + 2 < 8) addend_rows(i + 2, ad[i & 1]); }
            }
            newer = 64;
        } else {
            // bf16: a lane's 4 values are 8 bytes, so a direct store instruction touches 16 rows x 32 B = 16 L2 requests; the
            // CU issues about one request per 4 cycles and the 256 stores of a tile took ~10 us (measured per workgroup,
            // independent of how many other workgroups were storing) -- 10 % of a K = 4096 tile.  The tile is transposed
            // through LDS instead (a staging area behind the K-tile buffers, 16 rows per wave at a time): 8-byte writes in
            // the accumulator layout, 16-byte reads in row-major order, so every store instruction writes 4 rows x 256 B =
            // 8 full 128-byte lines (~5 us: the CU's 16 B/clk store path).
            char* st = smem + 2 * kBufBytes + wave * (16 * (ADD == 3 ? kStageRow3 : kStageRow));     // wave-private
            const int r = lane & 15, q = lane >> 4;
            sf_bf16* cbase = (sf_bf16*)p.e.C + (long)(mc + wr * 128 + q) * p.e.ldc + nc + wc * 128 + r * 8;
            if constexpr (ADD == 3) {
                // gate | up | act of the wave's 64 act columns: [row][3][128 B]; out as 8 rows x 128 B per store instruction
                const int srow3 = lane >> 3, sch = lane & 7;
                const int I = p.N >> 1, acol0 = (nc >> 1) + wc * 64 + sch * 8;
                sf_bf16* gub = (sf_bf16*)p.e.C + (long)(mc + wr * 128 + srow3) * p.e.ldc + acol0;
                sf_bf16* actb = p.e.sw_dgu + (long)(mc + wr * 128 + srow3) * p.e.sw_lddgu + acol0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        // (pairs packed by one v_cvt_pk_bf16_f32 each; the activation from the ROUNDED gate / up, like swiglu_fwd_kernel)
                        const sf_v4f ga = acc[i][2 * jj], ua = acc[i][2 * jj + 1];
                        sf_v2u og, ou, oa;
                        float av[4];
                        og[0] = sf_pack2_bf16(ga[0], ga[1]); og[1] = sf_pack2_bf16(ga[2], ga[3]);
                        ou[0] = sf_pack2_bf16(ua[0], ua[1]); ou[1] = sf_pack2_bf16(ua[2], ua[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = __builtin_bit_cast(float, (e & 1) ? (og[e >> 1] & 0xffff0000u) : (og[e >> 1] << 16));
                            const float u = __builtin_bit_cast(float, (e & 1) ? (ou[e >> 1] & 0xffff0000u) : (ou[e >> 1] << 16));
                            av[e] = sf_swiglu_fwd_elem<sf_bf16>(g, u);
                        }
                        oa[0] = sf_pack2_bf16(av[0], av[1]); oa[1] = sf_pack2_bf16(av[2], av[3]);
                        *reinterpret_cast<sf_v2u*>(st + r * kStageRow3 + jj * 32 + q * 8) = og;
                        *reinterpret_cast<sf_v2u*>(st + r * kStageRow3 + 128 + jj * 32 + q * 8) = ou;
                        *reinterpret_cast<sf_v2u*>(st + r * kStageRow3 + 256 + jj * 32 + q * 8) = oa;
                    }
                    sf_wave_lockstep();
#pragma unroll
                    for (int h = 0; h < 2; ++h) {          // rows 8 h + srow3 of this m-tile
                        const char* sp = st + (8 * h + srow3) * kStageRow3 + sch * 16;
                        const long ro = (long)(i * 16 + 8 * h);
                        *reinterpret_cast<sf_v8s*>(gub + ro * p.e.ldc) = *reinterpret_cast<const sf_v8s*>(sp);
                        *reinterpret_cast<sf_v8s*>(gub + ro * p.e.ldc + I) = *reinterpret_cast<const sf_v8s*>(sp + 128);
                        *reinterpret_cast<sf_v8s*>(actb + ro * p.e.sw_lddgu) = *reinterpret_cast<const sf_v8s*>(sp + 256);
                    }
                    sf_wave_lockstep();
                }
            } else if (direct_r) {
            // Residual form (round 6; o_proj and down_proj of the forward: 14 launches per step, which took the general store below until
            // then -- 8-byte residual loads and 8-byte stores, 16 row fragments of 32 B per instruction).  No LDS staging: a lane holds columns
            // 4 q .. 4 q + 3 of each 16-column block j; two v_permlane16_swap per block PAIR (2 jp, 2 jp + 1) exchange the packed halves
            // between lane rows q and q ^ 1, after which lane (q, r) holds 8 consecutive columns of row r -- block 2 jp + (q & 1), columns
            // 8 (q >> 1) .. + 7: the residual comes in and the sum leaves as 16 bytes per lane, 64 contiguous bytes per row and instruction.
            // round(round(acc) + residual), as sf_gemm_store4 does.  Measured against the general store: 0.459 -> 0.409 ms at 16384 x 4096 x
            // 4096, 1.339 -> 1.300 at K = 14336.  The same register transposition for the forms WITHOUT a second operand is a wash against
            // the staged whole lines (+-1 %), and 4 - 7 % slower for d(SwiGLU): profiles/r6_epi_direct_ab.jsonl -- they keep the staging.
            const int colq = (q & 1) * 16 + (q >> 1) * 8;
            const long drow = mc + wr * 128 + r;
            const int dcol = nc + wc * 128 + colq;
            sf_bf16* cdir = (sf_bf16*)p.e.C + drow * p.e.ldc + dcol;
            const sf_bf16* rdir = p.e.R + drow * p.e.ldr + dcol;
            sf_v8s rq[2][4];                // residual of the lane's 8 columns: m-tile i + 1 requested while m-tile i is converted
            auto res_load = [&](int i, int bufi) SF_INLINE_LAMBDA {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) rq[bufi][jp] = *reinterpret_cast<const sf_v8s*>(rdir + (long)(i * 16) * p.e.ldr + jp * 32);
            };
            res_load(0, 0);
            auto tile_out_res = [&](auto UNIT) SF_INLINE_LAMBDA {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + 1 < 8) res_load(i + 1, (i + 1) & 1);
                w4_fence();
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    sf_v4f v0 = acc[i][2 * jp], v1 = acc[i][2 * jp + 1];
                    if constexpr (!decltype(UNIT)::value) { v0 = v0 * alpha; v1 = v1 * alpha; }
                    unsigned d8[4] = {sf_pack2_bf16(v0[0], v0[1]), sf_pack2_bf16(v0[2], v0[3]), sf_pack2_bf16(v1[0], v1[1]), sf_pack2_bf16(v1[2], v1[3])};
                    sf_swap_rows16(d8[0], d8[2]);
                    sf_swap_rows16(d8[1], d8[3]);
                    sf_v4i o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __builtin_bit_cast(float, d8[e] << 16) + sf_bf2f((sf_bf16)rq[i & 1][jp][2 * e]);
                        const float hi = __builtin_bit_cast(float, d8[e] & 0xffff0000u) + sf_bf2f((sf_bf16)rq[i & 1][jp][2 * e + 1]);
                        o[e] = (int)sf_pack2_bf16(lo, hi);
                    }
                    *reinterpret_cast<sf_v4i*>(cdir + (long)(i * 16) * p.e.ldc + jp * 32) = o;
                }
            }
            };
            if (p.e.alpha != 1.0f) tile_out_res(std::false_type{});
            else tile_out_res(std::true_type{});
            } else {
            // (UNIT: alpha == 1, every GEMM of the training step -- its own copy of the loop: a per-value select costs more than the multiply)
            // d(SwiGLU) form: gate / up of m-tile i + 1 are requested before m-tile i is staged and converted -- with the lean staging
            // code the loads of an m-tile issued at its top were not back when its d(act) segment was (1.56 -> 1.76 ms per launch)
            sf_v8s gq[2][4], uq[2][4];
            auto gu_load = [&](int i, int bufi) SF_INLINE_LAMBDA {
                const long row0 = mc + wr * 128 + i * 16 + q;
                const int col = nc + wc * 128 + r * 8;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const sf_bf16* gp = p.e.sw_gu + (row0 + 4 * s4) * p.e.sw_ldgu + col;
                    gq[bufi][s4] = *reinterpret_cast<const sf_v8s*>(gp);
                    uq[bufi][s4] = *reinterpret_cast<const sf_v8s*>(gp + p.N);
                }
            };
            if constexpr (ADD == 2) gu_load(0, 0);
            auto tile_out = [&](auto UNIT) SF_INLINE_LAMBDA {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (ADD == 2) { if (i + 1 < 8) gu_load(i + 1, (i + 1) & 1); }
                // keeps the AGPR -> VGPR copies of m-tile i + 1 out of m-tile i: hoisted, all 256 of them fill the VGPR file and the four
                // staging reads below share one register quad (read, wait, store, four times).  Not in the d(SwiGLU) form: its m-tiles
                // load gate / up, and a fence keeps the next m-tile's loads from being issued under this one's stores (+0.2 ms per launch)
                if constexpr (ADD != 2) w4_fence();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sf_v4f v;
                    if constexpr (ADD == 1) v = acc[i][j] + ad[i & 1][j];   // (alpha == 1 is enforced by the launcher)
                    else if constexpr (decltype(UNIT)::value) v = acc[i][j];
                    else v = acc[i][j] * alpha;
                    sf_v2u o;
                    o[0] = sf_pack2_bf16(v[0], v[1]);
                    o[1] = sf_pack2_bf16(v[2], v[3]);
                    *reinterpret_cast<sf_v2u*>(st + r * kStageRow + j * 32 + q * 8) = o;
                }
                if constexpr (ADD == 1) { if (i + 2 < 8) addend_rows(i + 2, ad[i & 1]); }
                sf_wave_lockstep();
                if constexpr (ADD == 2) {   // its own instantiation: a different operator with its own line in a kernel trace
                    // fused d(SwiGLU): this row segment of d(act) never goes to memory.  gate / up of the same 8 positions
                    // come in (the loads of all four segments first), d(gate) / d(up) go out -- the arithmetic and its
                    // bf16 roundings are those of swiglu_bwd_kernel on the stored d(act)
                    const long row0 = mc + wr * 128 + i * 16 + q;
                    const int col = nc + wc * 128 + r * 8;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const sf_v8s d = *reinterpret_cast<const sf_v8s*>(st + (4 * s4 + q) * kStageRow + r * 16);
                        float dg[8], du[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            sf_swiglu_bwd_elem<sf_bf16>(sf_bf2f((sf_bf16)gq[i & 1][s4][e]), sf_bf2f((sf_bf16)uq[i & 1][s4][e]),
                                                        sf_bf2f((sf_bf16)d[e]), dg[e], du[e]);
                        sf_v4i og, ou;                       // (pairs by one v_cvt_pk_bf16_f32 each)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            og[e] = (int)sf_pack2_bf16(dg[2 * e], dg[2 * e + 1]);
                            ou[e] = (int)sf_pack2_bf16(du[2 * e], du[2 * e + 1]);
                        }
                        sf_bf16* op = p.e.sw_dgu + (row0 + 4 * s4) * p.e.sw_lddgu + col;
                        *reinterpret_cast<sf_v4i*>(op) = og;
                        *reinterpret_cast<sf_v4i*>(op + p.N) = ou;
                    }
                } else {
                    sf_v8s d[4];                             // rows 4*s4 + q of this m-tile, 16 bytes at column 8*r: four reads in
#pragma unroll                                                 // flight, then the four stores (one read-wait-store chain per row serialises)
                    for (int s4 = 0; s4 < 4; ++s4) d[s4] = *reinterpret_cast<const sf_v8s*>(st + (4 * s4 + q) * kStageRow + r * 16);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) *reinterpret_cast<sf_v8s*>(cbase + (long)(i * 16 + 4 * s4) * p.e.ldc) = d[s4];
                }
                sf_wave_lockstep();
            }
            };
            if (ADD == 0 && p.e.alpha != 1.0f) tile_out(std::false_type{});
            else tile_out(std::true_type{});
            }
            newer = 32;   // (the fused forms issue 48 stores / 64 stores + 64 loads: more, which is the safe direction)
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w4_store4<OUT_F32, ADD == 1 ? 1 : 0>(p, mc + wr * 128 + i * 16 + (lane & 15), nc + wc * 128 + j * 16 + 4 * (lane >> 4), acc[i][j]);
    }
    if (!has_next) break;

    // ---- the next tile: accumulators, then its K-tiles 0 and 1 (in flight since before the epilogue) must have landed.
    // vmcnt retires in order, so "all but the `newer` youngest" covers the DMAs without waiting for the epilogue's stores.
    tile = next;
    acc_init();
    if (newer >= 63) w4_wait_vm63();
    else if (newer >= 32) w4_wait_vm32();
    else w4_wait_all();
    w4_barrier();
#pragma unroll
    for (int g = 0; g < 16; ++g) read_frag(0, g, smem, 0);
    }   // tiles
#ifdef SF_ABLATE
    if ((p.cyc & 3) == 1 && tid == 0) *reinterpret_cast<float*>((char*)p.e.C + ((long)m0 * p.e.ldc + n0) * (OUT_F32 ? 4 : 2)) = cyc_loop;
#ifndef SF_EMU
    if ((p.cyc & 3) == 2 && tid == 0) {   // timeline of this workgroup + where it ran: 6 words at the tile origin
        rt[3] = (unsigned)__builtin_amdgcn_s_memrealtime();   // stores issued, not yet complete
        unsigned* o = reinterpret_cast<unsigned*>((char*)p.e.C + ((long)m0 * p.e.ldc + n0) * (OUT_F32 ? 4 : 2));
        o[0] = rt[0]; o[1] = rt[1]; o[2] = rt[2]; o[3] = rt[3];
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
#endif
#endif
}

#ifdef SF_ABLATE
#include "../../tools/experiments/sf_gemm256w4_variants.inc"
#endif

}  // namespace

// Grid of the persistent kernel: one workgroup per CU once there are more tiles than CUs (256 on MI355X; a multiple of 8,
// so tile t and tile t + grid sit on the same XCD).  SF_GEMM_PERSIST=0 (tools build) gives one workgroup per tile.
static inline unsigned sf_w4_grid(long nblk, int add = 0) {
    static const int persist = sf_knob("SF_GEMM_PERSIST", 1);
    static const int cus = [] {
        int dev = 0, n = 0;
#ifndef SF_EMU
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
#endif
#ifdef SF_EMU
        return 8 + 0 * (dev + n);   // interpreter: 8 "CUs", so the multi-tile path runs at test sizes
#else
        return n >= 8 ? n / 8 * 8 : 256;
#endif
    }();
    (void)add;   // (the row-addend form is persistent too since its addend moved to the epilogue)
    return (unsigned)((persist && nblk > cus) ? cus : nblk);
}

// one launch function per instantiation; each lives in its own translation unit (sf_gemm256w4_i*.hip) so that the
// eight 128-slot kernels compile in parallel (~25 s each; together in one unit they took > 10 minutes)
#define SF_W4_DECLARE(F32, ADD, SCHED) int sf_w4_launch_##F32##_##ADD##_##SCHED(const GemmW4Args& p, long nblk, void* stream)
#define SF_W4_DEFINE(F32, ADD, SCHED)                                                                                  \
    SF_W4_DECLARE(F32, ADD, SCHED) {                                                                                   \
        SF_W4_SMEM((gemm_nt_256w4_kernel<F32, ADD, SCHED>), w4_smem_bytes<ADD>());                                     \
        SF_LAUNCH((gemm_nt_256w4_kernel<F32, ADD, SCHED>), dim3(sf_w4_grid(nblk, ADD == 1)), dim3(256), w4_smem_bytes<ADD>(), stream, p); \
        return sf_check_launch("sf_gemm_nt(256w4)");                                                                   \
    }
SF_W4_DECLARE(0, 0, 12); SF_W4_DECLARE(0, 0, 13); SF_W4_DECLARE(1, 0, 12); SF_W4_DECLARE(1, 0, 13);
SF_W4_DECLARE(0, 1, 12); SF_W4_DECLARE(0, 1, 13); SF_W4_DECLARE(1, 1, 12); SF_W4_DECLARE(1, 1, 13);
SF_W4_DECLARE(0, 2, 12); SF_W4_DECLARE(0, 2, 13);
SF_W4_DECLARE(0, 3, 12); SF_W4_DECLARE(0, 3, 13);
SF_W4_DECLARE(0, 4, 12); SF_W4_DECLARE(0, 4, 13);
