// Shared pieces of the TTT attention kernels (sf_attn.hip: forward, backward preprocess, dQ; sf_attn_dkv.hip: dK / dV):
// constants, the L2-aware work order, the LDS tile layout (swizzle, LDS-DMA staging, MFMA fragment addressing).
#pragma once
#include "sf_api_internal.h"
#include "sf_util.h"
#include <stdlib.h>
#include <type_traits>

namespace sfattn {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -1.0e30f;
constexpr int kMaxDiag = 32;   // diagonal branches per launch = earlier TTT steps (ttt_length <= 33)
// fragment reads in flight ahead of their MFMA in the slot-planned kernels (sf_attn_dkv.hip, sf_attn_w1*.hip); 8 slots = 256 matrix-pipe
// cycles of lead (a build-time constant so that another depth can be A/B-ed as a second library: tools/attn_bench.py --lib=...)
#ifndef SF_ATTN_KAHEAD
#define SF_ATTN_KAHEAD 8
#endif

// ---- L2-aware work order ----------------------------------------------------
// Workgroup ids are handed to the 8 XCDs round-robin (block b runs on XCD b % 8: observed, relied on for speed only) and
// each XCD has its own 4 MiB L2.  The tiles a workgroup streams (K/V of one (batch, kv head) for fwd / dQ; Q and dO of
// the group's query heads for dK/dV) are shared by every workgroup of that (batch, kv head) "pair" -- 1 MiB of K + V at
// S = 2048, hd = 128.  Dispatching the heaviest blocks of ALL pairs first spread 32 different pairs over the 64
// workgroups resident on one XCD, so nearly every tile load missed L2: rocprofv3 showed 3.3 GB of fabric reads per
// forward launch against 0.34 GB of algorithmic bytes, i.e. the kernel ran at the fabric's 6.6 TB/s, not at the MFMA
// rate.  The work list is therefore ordered pair-major (heaviest block first INSIDE a pair) and cut into 8 contiguous
// ranges, one per XCD: the workgroups resident on an XCD at any time belong to one or two pairs.
SF_DEVICE int attn_work_index(int bid, int total, int l2_map) {
    if (!l2_map) return bid;
    const int per_xcd = (total + 7) >> 3;
    return (bid & 7) * per_xcd + (bid >> 3);   // >= total: no work for this workgroup
}
static inline unsigned attn_grid(long total, int l2_map) { return (unsigned)(l2_map ? 8 * ((total + 7) / 8) : total); }

// The same for a list of P pairs x W items, heaviest item first INSIDE a pair (forward / dQ: the causal key range grows with the query
// block).  Cutting that list into 8 contiguous ranges is balanced only when every XCD gets whole pairs.  With FEWER pairs than XCDs -- the
// batch-1 recipes: 1 x 4 kv heads, 1 x 2 -- a pair is spread over X = 8 / P XCDs and contiguous ranges hand the first of them the heavy half
// of the pair and the last the light one (P = 2, S 4096: 114 vs 18 key tiles per CU; the launch lasts as long as the heaviest XCD, measured
// 1.45x the batch-8 time per flop).  Then the X XCDs of a pair take its items IN TURN instead: every XCD of the pair walks the same
// heaviest-first profile, the pair's K / V are still shared by X L2s only.  -> index into the pair-major list, or -1 (no work)
SF_DEVICE int attn_pair_major_index(int bid, int W, int P) {
    const int xcd = bid & 7, pos = bid >> 3;
    if (P < 8 && 8 % P == 0) {
        const int X = 8 / P, w = pos * X + xcd % X;
        return w < W ? (xcd / X) * W + w : -1;
    }
    const int total = W * P, v = xcd * ((total + 7) >> 3) + pos;
    return v < total ? v : -1;
}
static inline unsigned attn_pair_major_grid(long W, long P) {
    if (P < 8 && 8 % P == 0) return (unsigned)(8 * ((W + 8 / P - 1) / (8 / P)));
    return (unsigned)(8 * ((W * P + 7) / 8));
}

struct AttnFwdArgs {
    const sf_bf16* q; long ldq;        // [B*S, nh*hd] view, row stride ldq
    const sf_bf16* k0; long ldk;       // step-0 keys [B*S, nkv*hd] view
    const sf_bf16* v0;                 // step-0 values [B*S, nkv*hd] view (row stride ldk)
    const sf_bf16* kd[kMaxDiag];       // diagonal-branch keys of steps 1..ndiag (views, stride ldk)
    const sf_bf16* vd[kMaxDiag];       // diagonal-branch values
    int ndiag;
    const int* kv_len;                 // [B] number of valid (non-padding) keys
    sf_bf16* o; long ldo;              // [B*S, nh*hd]
    float* lse;                        // [B, nh, S] natural-log lse over all S+k columns
    int B, S, nh, nkv;
    float scale;
    int l2_map;  // pair-major work order (attn_work_index); 0 only in the tools build's A/B
    int dbg;  // profiling experiments only (SF_ATTN_DBG): 1 = stage tile 0 only, 2 = skip the MFMA/softmax work
};

struct AttnBwdArgs {
    const sf_bf16* q; long ldq;      // natural
    const sf_bf16* dout; long lddo;  // natural
    const sf_bf16* k0; long ldk;     // natural
    const sf_bf16* v0; long ldv;     // natural
    const int* kv_len;
    const float* lse;                // [B, nh, S]
    const float* delta;              // [B, nh, S]
    const float* dq_init;            // fp32 [B*S, nh*hd] or null
    sf_bf16* dq; long lddq;          // out (dq kernel)
    float* dk; float* dv; long lddk; // fp32 accumulators (+=) [B*S, nkv*hd]  (dkv kernel)
    int B, S, nh, nkv;
    float scale;
    int l2_map;
    // dK/dV kernel, head split (small B * nkv): the query heads of a kv group are divided over `hsplit` workgroups, each writing
    // (=, not +=) its partial gradients to part_k / part_v [hsplit][B*S, nkv*hd] fp32; attn_dkv_reduce_kernel adds them to dk / dv
    int hsplit;
    float* part_k; float* part_v; long part_stride;
};

// ---- LDS tile swizzle -------------------------------------------------------
// 16-byte chunk c of tile row r is stored at chunk c ^ swz<HD>(r).  HD = 128 (16 chunks per 256-byte row):
// swz = ((r & 3) << 2) ^ ((r >> 2) & 3) is a bijection of r mod 16 onto 0..15, so a ds_read_b128 of 16
// consecutive rows at one logical chunk is conflict-free, and the 8 (row, column-half) pieces of a 32-lane
// ds_read_b64_tr_b16 pass land in 8 distinct 32-byte bank slots.  HD = 64 (8 chunks per row): r & 7.
// HD = 256 (32 chunks per 512-byte row = two 256-byte bank spans): the same 4-bit value XORed into the LOW four bits of the
// chunk index -- a chunk stays in its half of the row, and its position inside the bank span is permuted exactly as at HD 128.
template <int HD>
SF_DEVICE int swz(int r) {
    return HD >= 128 ? (((r & 3) << 2) ^ ((r >> 2) & 3)) : (r & 7);
}
// Workgroups per CU of the MFMA attention kernels (fwd, dQ): two 4-wave workgroups (2 waves per SIMD, 256 registers each) up to
// head_dim 128; at head_dim 256 the output accumulators alone are 128 registers per wave and the K / V ring is 128 KiB of LDS:
// one workgroup per CU with the whole register file.
template <int HD> struct AttnOcc { static constexpr int kWgPerCu = HD > 128 ? 1 : 2; };

// ---- LDS tile staging -------------------------------------------------------
// A tile of R rows x HD (row-major in LDS, 16-byte chunks XOR-swizzled by swz<HD>(row)) arrives as R*HD*2/1024 pieces
// of 1 KiB, one LDS-DMA wave-instruction each, through a BOUNDED buffer descriptor: rows at or past the end of the
// sequence read as zeros without a select or a branch.  `off[t]` = byte offset of this lane's 16-byte chunk of piece t
// relative to the tile's first row; a tile is staged with one add + one DMA per piece (the pointer form this replaces
// compiled to ~10 instructions per piece, exec-masked).
// Nothing in a tile loop may be a compiler-visible VMEM load: vmcnt is ONE in-order counter, so any wait the compiler
// inserts for a load of its own also drains the DMA prefetch of the next tile (which it cannot see).  Round 2's kernels
// had exactly that: the Q / K fragments loaded ahead of the loop were waited for at their first use INSIDE the loop
// (vmcnt(7)..vmcnt(0) in front of the QK^T MFMAs, every iteration), and the dK/dV kernel staged lse / delta through
// registers (global_load; vmcnt(0); ds_write) right behind the DMA issue -- the prefetch never overlapped anything.
template <int HD, int ROWS, int NW>
struct TileStage {
    static constexpr int CPR = HD / 8, RPI = 64 / CPR, NP = ROWS / RPI, NI = (NP + NW - 1) / NW;
    unsigned off[NI];
    int piece0;
    SF_DEVICE void init(long ld, int wave, int lane) {
        piece0 = wave * NI;
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int rr = (piece0 + t) * RPI + lane / CPR;
            const int lc = (lane % CPR) ^ swz<HD>(rr);
            off[t] = (unsigned)(((long)rr * ld + lc * 8) * 2);
        }
    }
    // `row_bytes` = first row of the tile * ld * 2 (wave-uniform); `lds` = tile base (wave-uniform)
    SF_DEVICE void issue(SfBufB buf, unsigned row_bytes, char* lds) const {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            if (NP % NW != 0 && piece0 + t >= NP) break;  // wave-uniform
            sf_bufb_glds16(buf, off[t] + row_bytes, lds + (piece0 + t) * 1024);
        }
    }
};
// dK/dV kernel: wait until at most ONE tile's worth of this wave's DMA pieces (Q + dO pieces + 1 bookkeeping piece) is outstanding
template <int HD>
SF_DEVICE void sf_wait_vm_tile() { sf_wait_vmcnt<2 * TileStage<HD, 64, 4>::NI + 1>(); }
// descriptor over the rows [0, S) of one (batch, head) slice: base = first row, row stride ld elements
template <int HD>
SF_DEVICE SfBufB rows_buf(const sf_bf16* base, long ld, int S) {
    return sf_make_bufb(base, (unsigned)((((long)S - 1) * ld + HD) * 2));
}
// Per-lane LDS byte offsets of the MFMA fragments, computed once per kernel so the tile loops issue
// ds_reads with (register + immediate) addresses only.  Tile row blocks start at multiples of 32
// rows, so (row & 7) == (lane & 7) for every fragment row.
template <int HD>
struct FragOff {
    // chunk indices at or above 16 (HD = 256) differ from those below only in bit 4, which the swizzle leaves alone: entry
    // i + NR of either table is entry i + 256 bytes -- an immediate offset of the read instead of another register
    static constexpr int NR = HD / 16 > 8 ? 8 : HD / 16, NT = HD / 32 > 4 ? 4 : HD / 32;
    int rows[NR];     // natural tile, k-step ks: (lane&31)*rowbytes + swizzled chunk (2ks + hi)
    int tr[NT][2];    // transpose-read of a natural tile, 32-column block db, rows r0+.. / r0+8+..: this lane's piece
    SF_DEVICE void init(int lane) {
        const int c = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < NR; ++ks) rows[ks] = c * (HD * 2) + (((2 * ks + hi) ^ swz<HD>(c)) << 4);
        // ds_read_b64_tr_b16: 16-lane group (lane>>4) covers tile rows r0 + 4*hi + 0..3 and columns
        // db*32 + 16*((lane>>4)&1) + 0..15; lane i of the group supplies piece i = (row i/4, cols 4*(i%4)..+3)
        const int i = lane & 15, t = 2 * ((lane >> 4) & 1) + ((i >> 1) & 1);
#pragma unroll
        for (int db = 0; db < NT; ++db)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec) {
                const int qx = 8 * sec + 4 * hi + (i >> 2);  // tile row (mod 16; r0 is a multiple of 16)
                tr[db][sec] = qx * (HD * 2) + ((((4 * db + t) ^ swz<HD>(qx))) << 4) + (i & 1) * 8;
            }
    }
    SF_DEVICE int row_off(int ks) const { return rows[ks % NR] + (ks / NR) * 256; }
    SF_DEVICE int tr_off(int db, int sec) const { return tr[db % NT][sec] + (db / NT) * 256; }
};
// A fragment (32 rows x 16 k) from a natural tile: row = r0 + (lane&31), k = 16*ks + 8*(lane>>5)
template <int HD>
SF_DEVICE sf_v8s frag_rows(const char* lds, int r0, int ks, const FragOff<HD>& fo) {
    return *reinterpret_cast<const sf_v8s*>(lds + r0 * (HD * 2) + fo.row_off(ks));
}
// A fragment for the "C-layout as B operand" contraction, taken from a NATURAL tile X[row][d] with the
// hardware transpose read: MFMA row = column d = db*32 + (lane&31) of the tile, k-slots = tile rows
// {r0 + 4*hi + 0..3} and {r0 + 8 + 4*hi + 0..3}  (r0 multiple of 16)
template <int HD>
SF_DEVICE sf_v8s frag_tr(const char* lds, int db, int r0, const FragOff<HD>& fo) {
    const sf_v4s lo = sf_ds_read_tr16(lds + r0 * (HD * 2) + fo.tr_off(db, 0));
    const sf_v4s up = sf_ds_read_tr16(lds + r0 * (HD * 2) + fo.tr_off(db, 1));
    return sf_v8s{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
}
SF_DEVICE sf_v8s pack_bf16x8(const sf_v16f& p, int r0) {      // four v_cvt_pk_bf16_f32 (scalar casts rely on the SLP vectoriser to pair up:
    typedef unsigned sf_v4u_ __attribute__((ext_vector_type(4)));   // without it they are 8 conversions + 4 v_perm)
    sf_v4u_ o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = sf_pack2_bf16(p[r0 + 2 * i], p[r0 + 2 * i + 1]);
    return __builtin_bit_cast(sf_v8s, o);
}
// Store one 32-column block of a row-per-lane 32x32 MFMA result as bf16: lane (c, hi) holds, for j = 0..3, the 4 columns 8 j + 4 hi .. + 3
// of its row -- natural stores are four 8-byte pieces per lane.  Two half-wave exchanges per column-group pair (v_permlane32_swap) give
// the lower lanes columns 16 p .. + 7 and the upper lanes 16 p + 8 .. + 15: TWO 16-byte stores per lane instead of four 8-byte ones, same
// bytes, same addresses (the epilogue is store-ISSUE-bound: MI355X_MICROARCH.md "attention epilogue store tail", guide T21).
// `v(i)` = value i (0..15) of the block in the accumulator's register order.  Every lane of the wave must call it (lane exchange);
// `live` masks the stores.
template <class F>
SF_DEVICE void store_row32_bf16(sf_bf16* dst, int hi, bool live, F&& v) {
    unsigned w[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w[j][0] = sf_pack2_bf16(v(4 * j), v(4 * j + 1));
        w[j][1] = sf_pack2_bf16(v(4 * j + 2), v(4 * j + 3));
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        sf_swap_halves(w[2 * pr][0], w[2 * pr + 1][0]);
        sf_swap_halves(w[2 * pr][1], w[2 * pr + 1][1]);
        if (live)
            *reinterpret_cast<sf_v4i*>(dst + 16 * pr + 8 * hi) = sf_v4i{(int)w[2 * pr][0], (int)w[2 * pr][1], (int)w[2 * pr + 1][0], (int)w[2 * pr + 1][1]};
    }
}
// row index inside a 32x32 MFMA result tile held by this lane in register r
SF_DEVICE int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

SF_DEVICE float dot8(sf_v8s a, sf_v8s b) {      // 8 bf16 products summed in fp32: four v_dot2c_f32_bf16
    typedef unsigned sf_v4u_ __attribute__((ext_vector_type(4)));
    const sf_v4u_ x = __builtin_bit_cast(sf_v4u_, a), y = __builtin_bit_cast(sf_v4u_, b);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s = sf_dot2_bf16(x[i], y[i], s);
    return s;
}


// compile-time loop: the body receives std::integral_constant<int, I> (asm "i" operands need constants)
template <int I, int N, class F>
SF_DEVICE void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


#ifdef SF_EMU
#define SF_SCHED_FENCE()
#else
#define SF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define SF_LAMBDA_INLINE __attribute__((always_inline))

// ---- a wave's asm-owned register bank --------------------------------------------------------------------------------
// The kernels that run ONE wave per SIMD (512 registers) keep their long-lived MFMA state in AGPRs the compiler never
// sees: NACC accumulator tiles of 16 registers at a[16 i], then NBF B-operand fragments of 4 registers.  Every MFMA that
// touches the bank is an asm statement naming those registers; the compiler allocates only what VALU touches and the
// A fragments in flight.  (Left to the register allocator, 128+ accumulator registers per wave ended up in architectural
// VGPRs, fragment addresses were spilled to AGPRs, ~290 registers per tile went through v_accvgpr_read / _write, and the
// pressure-bound scheduler serialised every fragment read.)  Nothing the compiler emits may touch a[0 : kEnd):
// tests/test_isa_invariants.py audits the ISA.  Interpreter build: plain arrays.
template <int NACC, int NBF>
struct AgprBank {
    static constexpr int kBf = 16 * NACC, kEnd = 16 * NACC + 4 * NBF;
    static_assert(kEnd == 96 || kEnd == 128 || kEnd == 192 || kEnd == 256, "add the clobber name for this bank size");
#ifdef SF_EMU
    sf_v16f acc[NACC];
    sf_v8s bf[NBF];
    SF_DEVICE void init() {
        for (int d = 0; d < NACC; ++d)
            for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    }
    template <int I> SF_DEVICE void set_b(sf_v8s v) { bf[I] = v; }
    template <int I> SF_DEVICE sf_v8s get_b() { return bf[I]; }
    template <int I, bool FIRST> SF_DEVICE void mfma_vb(sf_v16f& c, sf_v8s a) {   // c (+)= a . bf[I]   (c compiler-owned)
        if (FIRST) for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = sf_mfma32(a, bf[I], c);
    }
    template <int A> SF_DEVICE void mfma_acc(sf_v8s a, sf_v8s b) { acc[A] = sf_mfma32(a, b, acc[A]); }   // acc[A] += a . b
    template <int A> SF_DEVICE sf_v16f get() { return acc[A]; }
    template <int A> SF_DEVICE void scale(float f) { for (int r = 0; r < 16; ++r) acc[A][r] *= f; }
    template <int A> SF_DEVICE void axpy(float f, const sf_v16f& x) { for (int r = 0; r < 16; ++r) acc[A][r] = acc[A][r] * f + x[r]; }
    SF_DEVICE void drain() {}
#else
    // (the clobber makes the kernel descriptor allocate the bank)
    SF_DEVICE void init() {
        if constexpr (kEnd == 256) asm volatile("" ::: "a255");
        else if constexpr (kEnd == 192) asm volatile("" ::: "a191");
        else if constexpr (kEnd == 128) asm volatile("" ::: "a127");
        else asm volatile("" ::: "a95");
        static_for<0, 16 * NACC>([&](auto I) SF_LAMBDA_INLINE { asm volatile("v_accvgpr_write_b32 a[%c0], 0" : : "i"(decltype(I)::value)); });
    }
    template <int I> SF_DEVICE void set_b(sf_v8s v) {   // (the asm reads the value: the compiler waits for its load HERE)
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const v4i_ w = __builtin_bit_cast(v4i_, v);
        constexpr int B = kBf + 4 * I;
        asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\t"
                     "v_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                     : : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "i"(B), "i"(B + 1), "i"(B + 2), "i"(B + 3));
    }
    template <int I> SF_DEVICE sf_v8s get_b() {          // fragment I back out of the bank (epilogues: 4 reads instead of 4 live registers)
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        constexpr int B = kBf + 4 * I;
        int w0, w1, w2, w3;
        asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\t"
                     "v_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                     : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "i"(B), "i"(B + 1), "i"(B + 2), "i"(B + 3));
        return __builtin_bit_cast(sf_v8s, v4i_{w0, w1, w2, w3});
    }
    template <int I, bool FIRST> SF_DEVICE void mfma_vb(sf_v16f& c, sf_v8s a) {
        constexpr int B = kBf + 4 * I;
        if (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(c) : "v"(a), "i"(B), "i"(B + 3));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(c) : "v"(a), "i"(B), "i"(B + 3));
    }
    template <int A> SF_DEVICE void mfma_acc(sf_v8s a, sf_v8s b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * A), "i"(16 * A + 15));
    }
    template <int A> SF_DEVICE sf_v16f get() {   // only behind drain()
        sf_v16f r;
        static_for<0, 16>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int i = decltype(I)::value;
            float v;
            asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(16 * A + i));
            r[i] = v;
        });
        return r;
    }
    // acc[A] *= f, lane-wise (online-softmax rescale): read, multiply, write back; the caller keeps it clear of MFMAs in
    // flight on acc[A] (>= 3 MFMA slots after the last one that wrote it) and of the next one that reads it (s_nop inside)
    template <int A> SF_DEVICE void scale(float f) {
        static_for<0, 16>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int i = decltype(I)::value;
            float v;
            asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(16 * A + i));
            v *= f;
            asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(v), "i"(16 * A + i));
        });
    }
    // acc[A] = acc[A] * f + x, lane-wise, in place (the forward's diagonal-branch epilogue: the accumulators never leave the bank); same
    // distance rules as scale()
    template <int A> SF_DEVICE void axpy(float f, const sf_v16f& x) {
        static_for<0, 16>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int i = decltype(I)::value;
            float v;
            asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(16 * A + i));
            v = fmaf(v, f, x[i]);
            asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(v), "i"(16 * A + i));
        });
    }
    SF_DEVICE void drain() { sf_mfma_drain(); }   // MFMA results in the bank are read only behind this
#endif
};


// ---- TOOLS BUILD ONLY: cycle stamps inside the slot-planned tile loops ---------------------------------------------
// SfProf<PROF> accumulates s_memtime deltas between marks into 16 per-wave sums and flushes them to a device buffer
// (sf_tool_attn_prof); PROF = 0 (always, in the product library) compiles to nothing.  The streams are pinned by
// SF_SCHED_FENCE already, so the stamps (one SMEM instruction + two scalar adds each) barely move them.
#if defined(SF_ABLATE) && !defined(SF_EMU)
extern __device__ unsigned long long* g_attn_prof;
template <int PROF> struct SfProf {
    unsigned t[12], last, rt0;   // 32-bit sums: the whole lifetime of a workgroup is < 2^32 ticks (and SGPRs are scarce)
    SF_DEVICE void start() {
        if (PROF) { for (int i = 0; i < 12; ++i) t[i] = 0; __builtin_amdgcn_sched_barrier(0); rt0 = (unsigned)__builtin_amdgcn_s_memrealtime(); last = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    }
    template <int I> SF_DEVICE void mark() {
        if (PROF) { __builtin_amdgcn_sched_barrier(0); const unsigned x = (unsigned)__builtin_amdgcn_s_memtime(); t[I] += x - last; last = x; __builtin_amdgcn_sched_barrier(0); }
    }
    SF_DEVICE void flush(long slot, int lane) {
        if (PROF) {
            const unsigned span = (unsigned)__builtin_amdgcn_s_memrealtime() - rt0;
            if (g_attn_prof && lane == 0) { for (int i = 0; i < 12; ++i) g_attn_prof[slot * 16 + i] = t[i]; g_attn_prof[slot * 16 + 15] = span; }
        }
    }
};
#else
template <int PROF> struct SfProf {
    SF_DEVICE void start() {}
    template <int I> SF_DEVICE void mark() {}
    SF_DEVICE void flush(long, int) {}
};
#endif

}  // namespace sfattn

#ifdef SF_EMU
#define SF_ALLOW_SMEM(kernel, bytes)
#else
// > 64 KiB of dynamic LDS needs the opt-in (once per kernel instantiation)
#define SF_ALLOW_SMEM(kernel, bytes)                                                                        \
    do {                                                                                                    \
        static bool done_ = false;                                                                          \
        if (!done_) {                                                                                       \
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));  \
            (void)hipGetLastError();                                                                        \
            done_ = true;                                                                                   \
        }                                                                                                   \
    } while (0)
#endif

namespace sfattn {
// sf_attn_w1.hip: the one-wave-per-SIMD, slot-planned forward / dQ kernels (head_dim 256)
int attn_fwd_w1_launch(const AttnFwdArgs& p, int hd, void* stream);
int attn_bwd_dq_w1_launch(const AttnBwdArgs& p, int hd, void* stream);
// sf_attn_w1_dkv.hip: dK / dV at head_dim 256, a pair of waves per 32 keys splitting the score products (p.hsplit etc. set by the caller)
int attn_bwd_dkv_w1_launch(const AttnBwdArgs& p, int hd, void* stream);
}  // namespace sfattn

#define SF_HD_DISPATCH(hd, CALL)                                  \
    do {                                                          \
        if ((hd) == 128) { constexpr int HD = 128; CALL; }        \
        else if ((hd) == 64) { constexpr int HD = 64; CALL; }     \
        else if ((hd) == 256) { constexpr int HD = 256; CALL; }   \
        else SF_CHECK_ARG(false, "head_dim must be 64, 128 or 256"); \
    } while (0)

// (tools-build variants that exist for head_dim 64 / 128 only)
#define SF_HD_DISPATCH_128(hd, CALL)                              \
    do {                                                          \
        if ((hd) == 128) { constexpr int HD = 128; CALL; }        \
        else if ((hd) == 64) { constexpr int HD = 64; CALL; }     \
        else SF_CHECK_ARG(false, "this variant exists for head_dim 64 and 128 only"); \
    } while (0)
