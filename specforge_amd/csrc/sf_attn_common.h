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
constexpr int kMaxDiag = 8;

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
};

// ---- LDS tile swizzle -------------------------------------------------------
// 16-byte chunk c of tile row r is stored at chunk c ^ swz<HD>(r).  HD = 128 (16 chunks per 256-byte row):
// swz = ((r & 3) << 2) ^ ((r >> 2) & 3) is a bijection of r mod 16 onto 0..15, so a ds_read_b128 of 16
// consecutive rows at one logical chunk is conflict-free, and the 8 (row, column-half) pieces of a 32-lane
// ds_read_b64_tr_b16 pass land in 8 distinct 32-byte bank slots.  HD = 64 (8 chunks per row): r & 7.
template <int HD>
SF_DEVICE int swz(int r) {
    return HD == 128 ? (((r & 3) << 2) ^ ((r >> 2) & 3)) : (r & 7);
}

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
    int rows[HD / 16];  // natural tile, k-step ks: (lane&31)*rowbytes + swizzled chunk (2ks + hi)
    int tr[HD / 32][2]; // transpose-read of a natural tile, 32-column block db, rows r0+.. / r0+8+..: this lane's piece
    SF_DEVICE void init(int lane) {
        const int c = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) rows[ks] = c * (HD * 2) + (((2 * ks + hi) ^ swz<HD>(c)) << 4);
        // ds_read_b64_tr_b16: 16-lane group (lane>>4) covers tile rows r0 + 4*hi + 0..3 and columns
        // db*32 + 16*((lane>>4)&1) + 0..15; lane i of the group supplies piece i = (row i/4, cols 4*(i%4)..+3)
        const int i = lane & 15, t = 2 * ((lane >> 4) & 1) + ((i >> 1) & 1);
#pragma unroll
        for (int db = 0; db < HD / 32; ++db)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec) {
                const int qx = 8 * sec + 4 * hi + (i >> 2);  // tile row (mod 16; r0 is a multiple of 16)
                tr[db][sec] = qx * (HD * 2) + ((((4 * db + t) ^ swz<HD>(qx))) << 4) + (i & 1) * 8;
            }
    }
};
// A fragment (32 rows x 16 k) from a natural tile: row = r0 + (lane&31), k = 16*ks + 8*(lane>>5)
template <int HD>
SF_DEVICE sf_v8s frag_rows(const char* lds, int r0, int ks, const FragOff<HD>& fo) {
    return *reinterpret_cast<const sf_v8s*>(lds + r0 * (HD * 2) + fo.rows[ks]);
}
// A fragment for the "C-layout as B operand" contraction, taken from a NATURAL tile X[row][d] with the
// hardware transpose read: MFMA row = column d = db*32 + (lane&31) of the tile, k-slots = tile rows
// {r0 + 4*hi + 0..3} and {r0 + 8 + 4*hi + 0..3}  (r0 multiple of 16)
template <int HD>
SF_DEVICE sf_v8s frag_tr(const char* lds, int db, int r0, const FragOff<HD>& fo) {
    const sf_v4s lo = sf_ds_read_tr16(lds + r0 * (HD * 2) + fo.tr[db][0]);
    const sf_v4s up = sf_ds_read_tr16(lds + r0 * (HD * 2) + fo.tr[db][1]);
    return sf_v8s{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
}
SF_DEVICE sf_v8s pack_bf16x8(const sf_v16f& p, int r0) {
    sf_v8s o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (short)sf_f2bf(p[r0 + i]);
    return o;
}
// row index inside a 32x32 MFMA result tile held by this lane in register r
SF_DEVICE int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

SF_DEVICE float dot8(sf_v8s a, sf_v8s b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += sf_bf2f((sf_bf16)a[i]) * sf_bf2f((sf_bf16)b[i]);
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

#define SF_HD_DISPATCH(hd, CALL)                                  \
    do {                                                          \
        if ((hd) == 128) { constexpr int HD = 128; CALL; }        \
        else if ((hd) == 64) { constexpr int HD = 64; CALL; }     \
        else SF_CHECK_ARG(false, "head_dim must be 64 or 128");   \
    } while (0)
