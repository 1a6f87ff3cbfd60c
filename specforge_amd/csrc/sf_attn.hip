// TTT (training-time-test) attention of the EAGLE3 draft layer, forward and backward.
//
// Semantics = the reference's cache branch (specforge/modeling/draft/llama3_eagle.py:745-778,
// blueprint for the lse-merge: _FlashCachedMergeFunc 1024-1151): at unroll step k a query at
// position t attends to the step-0 keys 0..t (causal, right-padding excluded) plus ONE key per
// later step i=1..k -- the key at its own position t ("diagonal" branch terms, never masked);
// softmax in fp32 over those S+k columns with scale 1/sqrt(hd); GQA without repeat_kv.
//
// Kernels (all MFMA 32x32x16 bf16, 4 waves per workgroup, one wave owns 32 queries / keys):
//   attn_fwd    : flash loop over 64-key tiles of block 0 computing S^T = K.Q^T so that the
//                 softmax row of a query is lane-local, O^T += V^T.P^T with V^T read from a
//                 pre-transposed copy; the diagonal terms are folded in as k extra online-
//                 softmax updates in registers in the epilogue (no second pass over O).
//   attn_bwd_pre: delta = rowsum(dO*O) and the gradients of the diagonal terms (dq_init,
//                 dK_i, dV_i accumulated over the GQA group) -- HBM-bound row kernel.
//   attn_bwd_dq : per 128-query block, dQ^T += K^T.dS^T over the causal key tiles.
//   attn_bwd_dkv: per 128-key block of one kv head, loops the group's query heads and the
//                 query tiles at/after the diagonal; dK^T += Q^T.dS, dV^T += dO^T.P; owns its
//                 keys exclusively, so dK/dV accumulate into fp32 buffers without atomics.
// Tiles are staged HBM -> LDS by 16-byte LDS-DMA with an XOR chunk swizzle on the source
// address (same scheme as sf_gemm.hip).
#include "sf_api_internal.h"
#include "sf_util.h"
#include <stdlib.h>

namespace {

#ifdef SF_ABLATE   // profiling experiments of the tools build: 1 = stage tile 0 only, 2 = skip the MFMA / softmax work
#define SF_ATTN_DBG(p, bit) ((p).dbg & (bit))
#else
#define SF_ATTN_DBG(p, bit) false
#endif

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

// ------------------------------------------------------------------ forward
// -inf where the tile-relative key index `c` is past `rel` (= last visible key - first key of the block - 4 * hi)
SF_DEVICE void mask_scores(sf_v16f& s, int rel) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) > rel) s[r] = -INFINITY;
}

template <int HD, int NW>
SF_GLOBAL void SF_LAUNCH_BOUNDS(NW * 64, 2) attn_fwd_kernel(AttnFwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, QB = NW * 32, TILE = 128 * HD * 2;
    SF_DYN_SMEM(smem);  // 2 x { K [64][HD], V [64][HD] }
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    // 1-D grid, heaviest work first: the causal key range grows with the query block, so the LAST query blocks are
    // dispatched first (longest-processing-time order keeps the tail of the launch short)
    const int nqb = (p.S + QB - 1) / QB, per_qb = p.nh * p.B;
    int qbi, h, b, g;
    if (p.l2_map) {   // pair-major: (batch, kv head) -> query block (last = heaviest first) -> query head of the group
        const int nrep = p.nh / p.nkv, W = nrep * nqb;
        const int v = attn_work_index((int)blockIdx.x, W * p.nkv * p.B, 1);
        if (v >= W * p.nkv * p.B) return;
        const int pr = v / W, w = v - pr * W;
        b = pr / p.nkv; g = pr - b * p.nkv;
        qbi = nqb - 1 - w / nrep; h = g * nrep + w % nrep;
    } else {
        const int bid = (int)blockIdx.x, hb = bid % per_qb;
        qbi = nqb - 1 - bid / per_qb; h = hb % p.nh; b = hb / p.nh;
        g = h / (p.nh / p.nkv);
    }
    const int qb0 = qbi * QB;
    const int S = p.S;
    const int kvlen = p.kv_len ? p.kv_len[b] : S;
    const int qw0 = qb0 + wave * 32;
    const int qi = qw0 + c;                       // this lane's query position
    const bool qok = qi < S;
    const long qrow = (long)b * S + (qok ? qi : S - 1);
    const float sc = p.scale * kLog2e;
    const int lim = qi < kvlen - 1 ? qi : kvlen - 1;   // last key this query attends to in block 0
    FragOff<HD> fo;
    fo.init(lane);

    sf_v8s qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const sf_v8s*>(p.q + qrow * p.ldq + h * HD + 16 * ks + 8 * hi);

    sf_v16f acc_o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m = kNegBig, lpart = 0.f;

    const SfBufB kbuf = rows_buf<HD>(p.k0 + (long)b * S * p.ldk + g * HD, p.ldk, S);
    const SfBufB vbuf = rows_buf<HD>(p.v0 + (long)b * S * p.ldk + g * HD, p.ldk, S);
    const unsigned tile_bytes = (unsigned)(64 * p.ldk * 2);
    TileStage<HD, 64, NW> st;
    st.init(p.ldk, wave, lane);
    int kend = qb0 + QB < S ? qb0 + QB : S;  // causal upper bound for this block
    if (kvlen < kend) kend = kvlen;
    const int ntiles = (kend + 63) / 64;
    if (ntiles > 0) {
        st.issue(kbuf, 0, smem);
        st.issue(vbuf, 0, smem + 64 * HD * 2);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) sf_pin(qf[ks]);   // the Q loads are complete HERE, not at their first use in the loop
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 64;
        sf_wait_vm0();
        sf_syncthreads();  // tile kt landed for everyone; buffer (kt+1)&1 is no longer being read
        if (kt + 1 < ntiles && !SF_ATTN_DBG(p, 1)) {
            char* nb = smem + ((kt + 1) & 1) * TILE;
            st.issue(kbuf, (unsigned)(kt + 1) * tile_bytes, nb);
            st.issue(vbuf, (unsigned)(kt + 1) * tile_bytes, nb + 64 * HD * 2);
        }
        const char* lds_k = smem + (kt & 1) * TILE;
        const char* lds_v = lds_k + 64 * HD * 2;
        if (key0 > qw0 + 31) continue;  // whole tile above this wave's diagonal (wave-uniform)
        if (SF_ATTN_DBG(p, 2)) continue;
        const bool need_mask = (key0 + 63 > qw0) || (key0 + 63 >= kvlen);
        sf_v16f s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        // the two 32-key blocks are independent accumulation chains: interleave them so that no MFMA has to
        // wait for the result of the one issued right before it
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s[kb] = sf_mfma32(frag_rows<HD>(lds_k, kb * 32, ks, fo), qf[ks], s[kb]);
        // scores stay unscaled in the accumulators; only diagonal / padded tiles pay for masking (2 VALU per score)
        if (need_mask) {
            const int rel = lim - key0 - 4 * hi;
            mask_scores(s[0], rel);
            mask_scores(s[1], rel - 32);
        }
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, fmaxf(s[0][r], s[1][r]));
        mt = sf_pair_max(mt);
        const float mts = mt * sc;  // running max m lives in the scaled log2 domain
        // deferred rescale: O and l are rescaled only when some row's max grew by more than 2^8
        if (!sf_all(mts - m <= 8.0f)) {
            const float mn = fmaxf(m, mts);
            const float alpha = sf_exp2_raw(m - mn);
            m = mn;
            lpart *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
        }
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = sf_exp2_raw(fmaf(s[kb][r], sc, -m));  // exp2(-inf) == 0 for masked keys
                s[kb][r] = e;
                ps += e;
            }
        lpart += ps;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const sf_v8s pf = pack_bf16x8(s[kb], 8 * jp);
#pragma unroll
                for (int d = 0; d < DB; ++d)
                    acc_o[d] = sf_mfma32(frag_tr<HD>(lds_v, d, kb * 32 + 16 * jp, fo), pf, acc_o[d]);
            }
    }
    float l = sf_pair_sum(lpart);

    // diagonal branch terms: one extra key per later TTT step at the query's own position.  A wave only ever needs the
    // K_i / V_i rows of its OWN 32 queries, so each wave stages them into a private slice of the (now free) tile
    // buffers -- no workgroup barrier per branch -- and the next branch's rows are in flight while this one is applied
    // (round 2: four cooperative 16 KiB stagings, two barriers and an exposed wait per branch).
    if (p.ndiag > 0) {
        constexpr int PRIV = 2 * TILE / NW;             // bytes of LDS per wave: K_i rows | V_i rows
        static_assert(PRIV >= 2 * 32 * HD * 2, "a wave's slice of the tile buffers holds 32 rows of K_i and of V_i");
        char* mine = smem + wave * PRIV;
        TileStage<HD, 32, 1> ds;
        ds.init(p.ldk, 0, lane);
        const unsigned my_rows = (unsigned)((long)qw0 * p.ldk * 2);
        sf_syncthreads();                               // every wave is done with the last K/V tile
        auto stage_diag = [&](int i) {
            const long slice = (long)b * S * p.ldk + g * HD;
            ds.issue(rows_buf<HD>(p.kd[i] + slice, p.ldk, S), my_rows, mine);
            ds.issue(rows_buf<HD>(p.vd[i] + slice, p.ldk, S), my_rows, mine + 32 * HD * 2);
        };
        stage_diag(0);
        for (int i = 0; i < p.ndiag; ++i) {
            sf_wait_vm0();
            sf_wave_lockstep();   // (interpreter only: the other lanes' pieces of this wave's DMA have been copied)
            sf_v8s kk[KS];
            sf_v4s vv4[DB * 4];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kk[ks] = frag_rows<HD>(mine, 0, ks, fo);
            const char* vrow = mine + 32 * HD * 2 + c * (HD * 2) + 8 * hi;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    vv4[d * 4 + j] = *reinterpret_cast<const sf_v4s*>(vrow + (((4 * d + j) ^ swz<HD>(c)) << 4));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sf_pin(kk[ks]);      // the reads have returned: the slice may be overwritten
#pragma unroll
            for (int j = 0; j < DB * 4; ++j) sf_pin(vv4[j]);
            if (i + 1 < p.ndiag) stage_diag(i + 1);
            float dp = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp += dot8(qf[ks], kk[ks]);
            dp = sf_pair_sum(dp);
            const float s2 = dp * sc;
            const float mn = fmaxf(m, s2);
            const float alpha = sf_exp2(m - mn);
            const float e = sf_exp2(s2 - mn);
            m = mn;
            l = l * alpha + e;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const sf_v4s vv = vv4[d * 4 + j];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc_o[d][4 * j + t] = acc_o[d][4 * j + t] * alpha + e * sf_bf2f((sf_bf16)vv[t]);
                }
        }
    }
    if (!qok) return;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    sf_bf16* orow = p.o + qrow * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sf_v4s ov;
#pragma unroll
            for (int t = 0; t < 4; ++t) ov[t] = (short)sf_f2bf(acc_o[d][4 * j + t] * inv);
            *reinterpret_cast<sf_v4s*>(orow + d * 32 + 8 * j + 4 * hi) = ov;
        }
    if (hi == 0) p.lse[((long)b * p.nh + h) * S + qi] = l > 0.f ? (m + log2f(l)) * kLn2 : kNegBig;
}

// ------------------------------------------------------ backward: preprocess
struct AttnBwdPreArgs {
    const sf_bf16* q; long ldq;
    const sf_bf16* o; long ldo;
    const sf_bf16* dout; long lddo;
    const sf_bf16* kd[kMaxDiag];
    const sf_bf16* vd[kMaxDiag];
    float* dkd[kMaxDiag];  // fp32 accumulators [B*S, nkv*hd] (+=)
    float* dvd[kMaxDiag];
    long ldk, lddk;
    int ndiag;
    const float* lse;   // [B, nh, S]
    float* delta;       // [B, nh, S]
    float* dq_init;     // [B*S, nh*hd] fp32, written when ndiag > 0
    int dq_accumulate;  // this launch handles a later chunk of diagonals: dq_init += instead of =
    int B, S, nh, nkv, hd;
    float scale;
};

// A group of HD/8 lanes owns one token row (8 consecutive d per lane, 16-byte loads), so a wave covers 4 (HD=128)
// or 8 (HD=64) rows; wave w handles kv groups w, w+4, ... and walks the query heads of a group serially, which
// keeps the dK_i / dV_i sums over those heads in registers.  Every dot product is an all-reduce over the lane group
// done with DPP row operations (sf_row_sum) -- no LDS crossbar traffic.  One launch handles at most kPreChunk
// diagonals (their K/V/dK/dV slices live in registers); the host splits longer lists into chunks, later chunks
// accumulate into dq_init.
constexpr int kPreChunk = 4;
template <int HD>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2) attn_bwd_pre_kernel(AttnBwdPreArgs p) {
    constexpr int ND = kPreChunk;
    constexpr int LPH = HD / 8;     // lanes per row
    constexpr int RPW = 64 / LPH;   // rows per wave
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int sub = lane / LPH, li = lane % LPH;
    const int N = p.B * p.S;
    const int row_raw = (int)blockIdx.x * RPW + sub;
    const bool live = row_raw < N;
    const int row = live ? row_raw : N - 1;   // dead lane groups shadow a valid row and skip every store
    const int b = row / p.S, t = row - b * p.S;
    const int nrep = p.nh / p.nkv;
    const int d0 = li * 8;
    for (int g = wave; g < p.nkv; g += 4) {
        sf_v8s kv[ND], vv[ND];
        float dk[ND][8], dv[ND][8];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { dk[i][e] = 0.f; dv[i][e] = 0.f; }
            if (i < p.ndiag) {
                kv[i] = *reinterpret_cast<const sf_v8s*>(p.kd[i] + (long)row * p.ldk + g * HD + d0);
                vv[i] = *reinterpret_cast<const sf_v8s*>(p.vd[i] + (long)row * p.ldk + g * HD + d0);
            } else {
                kv[i] = sf_v8s{0, 0, 0, 0, 0, 0, 0, 0};
                vv[i] = kv[i];
            }
        }
        for (int hh = 0; hh < nrep; ++hh) {
            const int h = g * nrep + hh;
            const int col = h * HD + d0;
            float qv[8], ov[8], dov[8];
            SfVec8<sf_bf16>::ld(p.q + (long)row * p.ldq + col, qv);
            SfVec8<sf_bf16>::ld(p.o + (long)row * p.ldo + col, ov);
            SfVec8<sf_bf16>::ld(p.dout + (long)row * p.lddo + col, dov);
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += ov[e] * dov[e];
            dl = sf_row_sum<LPH>(dl);
            const long lidx = ((long)b * p.nh + h) * p.S + t;
            if (li == 0 && live) p.delta[lidx] = dl;
            if (p.ndiag > 0) {
                const float lse = p.lse[lidx];
                float dq[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dq[e] = 0.f;
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    if (i < p.ndiag) {
                        float sdot = 0.f, pdot = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            sdot += qv[e] * sf_bf2f((sf_bf16)kv[i][e]);
                            pdot += dov[e] * sf_bf2f((sf_bf16)vv[i][e]);
                        }
                        sdot = sf_row_sum<LPH>(sdot);
                        pdot = sf_row_sum<LPH>(pdot);
                        const float pi = sf_exp(sdot * p.scale - lse);
                        const float ds = pi * (pdot - dl) * p.scale;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            dq[e] += ds * sf_bf2f((sf_bf16)kv[i][e]);
                            dk[i][e] += ds * qv[e];
                            dv[i][e] += pi * dov[e];
                        }
                    }
                }
                if (live) {
                    float* dqp = p.dq_init + (long)row * (p.nh * HD) + col;
                    if (p.dq_accumulate) {   // later chunk of diagonals: add to what the first chunk wrote
                        float prev[8];
                        SfVec8<float>::ld(dqp, prev);
#pragma unroll
                        for (int e = 0; e < 8; ++e) dq[e] += prev[e];
                    }
                    SfVec8<float>::st(dqp, dq);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
                if (i < p.ndiag) {
                    const long idx = (long)row * p.lddk + g * HD + d0;
                    float a[8], c[8];
                    SfVec8<float>::ld(p.dkd[i] + idx, a);
                    SfVec8<float>::ld(p.dvd[i] + idx, c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a[e] += dk[i][e]; c[e] += dv[i][e]; }
                    SfVec8<float>::st(p.dkd[i] + idx, a);
                    SfVec8<float>::st(p.dvd[i] + idx, c);
                }
        }
    }
}

// ------------------------------------------------------------- backward: dQ
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

template <int HD, int NW>
SF_GLOBAL void SF_LAUNCH_BOUNDS(NW * 64, 2) attn_bwd_dq_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, QB = NW * 32, TILE = 128 * HD * 2;
    SF_DYN_SMEM(smem);  // 2 x { K [64][HD], V [64][HD] }; K serves both S^T = K.Q^T and (transpose-read) dQ^T += K^T.dS^T
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    // 1-D grid, heaviest work first: the causal key range grows with the query block, so the LAST query blocks are
    // dispatched first (longest-processing-time order keeps the tail of the launch short)
    const int nqb = (p.S + QB - 1) / QB, per_qb = p.nh * p.B;
    int qbi, h, b, g;
    if (p.l2_map) {   // pair-major: (batch, kv head) -> query block (last = heaviest first) -> query head of the group
        const int nrep = p.nh / p.nkv, W = nrep * nqb;
        const int v = attn_work_index((int)blockIdx.x, W * p.nkv * p.B, 1);
        if (v >= W * p.nkv * p.B) return;
        const int pr = v / W, w = v - pr * W;
        b = pr / p.nkv; g = pr - b * p.nkv;
        qbi = nqb - 1 - w / nrep; h = g * nrep + w % nrep;
    } else {
        const int bid = (int)blockIdx.x, hb = bid % per_qb;
        qbi = nqb - 1 - bid / per_qb; h = hb % p.nh; b = hb / p.nh;
        g = h / (p.nh / p.nkv);
    }
    const int qb0 = qbi * QB;
    const int S = p.S;
    const int kvlen = p.kv_len ? p.kv_len[b] : S;
    const int qw0 = qb0 + wave * 32;
    const int qi = qw0 + c;
    const bool qok = qi < S;
    const long qrow = (long)b * S + (qok ? qi : S - 1);
    const float sc = p.scale * kLog2e;
    const int lim = qi < kvlen - 1 ? qi : kvlen - 1;
    FragOff<HD> fo;
    fo.init(lane);
    const long li = ((long)b * p.nh + h) * S + (qok ? qi : S - 1);
    float lse2 = p.lse[li] * kLog2e;
    float dlt = p.delta[li];

    sf_v8s qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *reinterpret_cast<const sf_v8s*>(p.q + qrow * p.ldq + h * HD + 16 * ks + 8 * hi);
        dof[ks] = *reinterpret_cast<const sf_v8s*>(p.dout + qrow * p.lddo + h * HD + 16 * ks + 8 * hi);
    }
    sf_v16f acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    const SfBufB kbuf = rows_buf<HD>(p.k0 + (long)b * S * p.ldk + g * HD, p.ldk, S);
    const SfBufB vbuf = rows_buf<HD>(p.v0 + (long)b * S * p.ldv + g * HD, p.ldv, S);
    const unsigned ktile = (unsigned)(64 * p.ldk * 2), vtile = (unsigned)(64 * p.ldv * 2);
    TileStage<HD, 64, NW> stk, stv;
    stk.init(p.ldk, wave, lane);
    stv.init(p.ldv, wave, lane);
    int kend = qb0 + QB < S ? qb0 + QB : S;
    if (kvlen < kend) kend = kvlen;
    const int ntiles = (kend + 63) / 64;
    if (ntiles > 0) {
        stk.issue(kbuf, 0, smem);
        stv.issue(vbuf, 0, smem + 64 * HD * 2);
    }
    // every value loaded ahead of the loop is complete HERE (see TileStage): no compiler-placed vmcnt inside the loop
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { sf_pin(qf[ks]); sf_pin(dof[ks]); }
    sf_pin(lse2);
    sf_pin(dlt);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 64;
        sf_wait_vm0();
        sf_syncthreads();
        if (kt + 1 < ntiles) {
            char* nb = smem + ((kt + 1) & 1) * TILE;
            stk.issue(kbuf, (unsigned)(kt + 1) * ktile, nb);
            stv.issue(vbuf, (unsigned)(kt + 1) * vtile, nb + 64 * HD * 2);
        }
        const char* lds_k = smem + (kt & 1) * TILE;
        const char* lds_v = lds_k + 64 * HD * 2;
        if (key0 > qw0 + 31) continue;
        sf_v16f s[2], dp[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[kb][r] = 0.f; dp[kb][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)  // four independent accumulation chains, round-robin
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                s[kb] = sf_mfma32(frag_rows<HD>(lds_k, kb * 32, ks, fo), qf[ks], s[kb]);
                dp[kb] = sf_mfma32(frag_rows<HD>(lds_v, kb * 32, ks, fo), dof[ks], dp[kb]);
            }
        const bool need_mask = (key0 + 63 > qw0) || (key0 + 63 >= kvlen);  // wave-uniform
        if (need_mask) {
            const int rel = lim - key0 - 4 * hi;
            mask_scores(s[0], rel);          // exp2(-inf) == 0: masked keys carry no probability, hence no dS
            mask_scores(s[1], rel - 32);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = sf_exp2_raw(fmaf(s[kb][r], sc, -lse2));
                s[kb][r] = pv * (dp[kb][r] - dlt);  // dS^T
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const sf_v8s df = pack_bf16x8(s[kb], 8 * jp);
#pragma unroll
                for (int d = 0; d < DB; ++d)
                    acc[d] = sf_mfma32(frag_tr<HD>(lds_k, d, kb * 32 + 16 * jp, fo), df, acc[d]);
            }
    }
    if (!qok) return;
    sf_bf16* orow = p.dq + qrow * p.lddq + h * HD;
    // the diagonal branches' share of dQ (attn_bwd_pre): all 16 vector loads in flight at once (a per-element
    // `if (init) v += init[i]` compiled to 64 dependent dword loads, each behind its own vmcnt(0))
    sf_v4f init[DB * 4];
    if (p.dq_init) {
        const float* irow = p.dq_init + qrow * ((long)p.nh * HD) + h * HD;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) init[d * 4 + j] = *reinterpret_cast<const sf_v4f*>(irow + d * 32 + 8 * j + 4 * hi);
    } else {
#pragma unroll
        for (int j = 0; j < DB * 4; ++j) init[j] = sf_v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sf_v4s ov;
#pragma unroll
            for (int t = 0; t < 4; ++t) ov[t] = (short)sf_f2bf(acc[d][4 * j + t] * p.scale + init[d * 4 + j][t]);
            *reinterpret_cast<sf_v4s*>(orow + d * 32 + 8 * j + 4 * hi) = ov;
        }
}

// ---------------------------------------------------------- backward: dK, dV
// Workgroup = NW waves = NW/2 key sub-blocks of 32 keys x 2 roles: waves [0, NW/2) accumulate dV^T,
// waves [NW/2, NW) accumulate dK^T of the same keys (each recomputes S; a wave then carries ONE
// 64-register accumulator set, so the kernel fits 2 waves/SIMD without spilling and the two roles
// of a key sub-block sit on the same SIMD and overlap exp/LDS work with MFMA).
// LDS (double buffered): Q [64][HD], dO [64][HD], lse2[64], delta[64]; Q^T / dO^T fragments come from the
// same tiles through the hardware transpose read
template <int HD, int NW>
SF_GLOBAL void SF_LAUNCH_BOUNDS(NW * 64, 2) attn_bwd_dkv_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, NSUB = NW / 2, KB = NSUB * 32, TILE = 128 * HD * 2 + 512;
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    const int role = wave / NSUB;  // 0: dV, 1: dK   (wave-uniform)
    const int sub = wave - role * NSUB;
    // 1-D grid, heaviest first: key block 0 sees every query tile, the last one only the final tiles
    const int per_kb = p.nkv * p.B;
    int kbi, g, b;
    if (p.l2_map) {   // pair-major: the key blocks of one (batch, kv head) stream the same Q / dO tiles
        const int nkb = (p.S + KB - 1) / KB;
        const int v = attn_work_index((int)blockIdx.x, nkb * per_kb, 1);
        if (v >= nkb * per_kb) return;
        const int pr = v / nkb;
        kbi = v - pr * nkb; b = pr / p.nkv; g = pr - b * p.nkv;
    } else {
        const int bid = (int)blockIdx.x, gb = bid % per_kb;
        kbi = bid / per_kb; g = gb % p.nkv; b = gb / p.nkv;
    }
    const int kb0 = kbi * KB;
    const int S = p.S, nrep = p.nh / p.nkv;
    const int kvlen = p.kv_len ? p.kv_len[b] : S;
    const int kw0 = kb0 + sub * 32;
    const int ki = kw0 + c;  // this lane's key (column of S)
    const bool kok = ki < S;
    const long krow = (long)b * S + (kok ? ki : S - 1);
    const float sc = p.scale * kLog2e;
    FragOff<HD> fo;
    fo.init(lane);

    sf_v8s kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const sf_v8s*>(p.k0 + krow * p.ldk + g * HD + 16 * ks + 8 * hi);
        vf[ks] = kf[ks];
        if (role == 1) vf[ks] = *reinterpret_cast<const sf_v8s*>(p.v0 + krow * p.ldv + g * HD + 16 * ks + 8 * hi);
    }
    sf_v16f acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    const bool block_live = kb0 < kvlen;  // keys at/after kv_len never receive probability mass
    const int qt_first = kb0 / 64;
    const int nqt = (S + 63) / 64;
    // a query q of the tile is visible to this lane's key iff  ki <= q < S  (and the key itself is valid): with
    // q = q0 + 4*hi + C (C a compile-time constant per register) that is  lo <= C < up  for two per-tile values
    const int key_lo = ki < kvlen ? ki : 0x3fffffff;
    TileStage<HD, 64, NW> stq, stdo;
    stq.init(p.ldq, wave, lane);
    stdo.init(p.lddo, wave, lane);
    const unsigned qtile = (unsigned)(64 * p.ldq * 2), dotile = (unsigned)(64 * p.lddo * 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { sf_pin(kf[ks]); sf_pin(vf[ks]); }   // complete HERE (see TileStage)
    int it = 0;
    if (block_live)
        for (int hh = 0; hh < nrep; ++hh) {
            const int h = g * nrep + hh;
            const SfBufB qbuf = rows_buf<HD>(p.q + (long)b * S * p.ldq + h * HD, p.ldq, S);
            const SfBufB dobuf = rows_buf<HD>(p.dout + (long)b * S * p.lddo + h * HD, p.lddo, S);
            const SfBufB lsebuf = sf_make_bufb(p.lse + ((long)b * p.nh + h) * S, (unsigned)S * 4u);
            const SfBufB dltbuf = sf_make_bufb(p.delta + ((long)b * p.nh + h) * S, (unsigned)S * 4u);
            // lse / delta of the tile's 64 queries ride the same LDS-DMA path (one 4-byte-per-lane piece each, issued by
            // waves 0 and 1): nothing in this loop is a load the compiler counts
            auto stage = [&](char* dst, int qt) {
                stq.issue(qbuf, (unsigned)qt * qtile, dst);
                stdo.issue(dobuf, (unsigned)qt * dotile, dst + 64 * HD * 2);
                if (wave == 0) sf_bufb_glds4(lsebuf, (unsigned)(qt * 64 + lane) * 4u, dst + 128 * HD * 2);
                if (wave == 1 % NW) sf_bufb_glds4(dltbuf, (unsigned)(qt * 64 + lane) * 4u, dst + 128 * HD * 2 + 256);
            };
            // the buffer parity continues across the heads of the group: `it` counts tiles globally
            if (qt_first < nqt) stage(smem + (it & 1) * TILE, qt_first);
            for (int qt = qt_first; qt < nqt; ++qt, ++it) {
                const int q0 = qt * 64;
                sf_wait_vm0();
                sf_syncthreads();
                if (qt + 1 < nqt) stage(smem + ((it + 1) & 1) * TILE, qt + 1);
                const char* lds_q = smem + (it & 1) * TILE;
                const char* lds_do = lds_q + 64 * HD * 2;
                const char* lds_x = role == 0 ? lds_do : lds_q;  // dV^T += dO^T.P   |   dK^T += Q^T.dS
                const float* lds_lse = reinterpret_cast<const float*>(lds_q + 128 * HD * 2);
                const float* lds_dlt = lds_lse + 64;
                if (q0 + 63 < kw0) continue;  // every query of the tile is before this wave's keys
                const int lo = key_lo - q0 - 4 * hi, up = S - q0 - 4 * hi;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    sf_v16f s, dp;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) s = sf_mfma32(frag_rows<HD>(lds_q, qb * 32, ks, fo), kf[ks], s);
                    if (role == 1) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) dp = sf_mfma32(frag_rows<HD>(lds_do, qb * 32, ks, fo), vf[ks], dp);
                    }
                    // rows crow(4j..4j+3) are consecutive: one 16-byte LDS read per 4 rows
                    const bool need_mask = (q0 + qb * 32 < kw0 + 32) || (kw0 + 31 >= kvlen) || (q0 + 63 >= S);  // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ql0 = qb * 32 + 8 * j + 4 * hi;
                        const sf_v4f l4 = *reinterpret_cast<const sf_v4f*>(lds_lse + ql0);
                        sf_v4f d4 = l4;
                        if (role == 1) d4 = *reinterpret_cast<const sf_v4f*>(lds_dlt + ql0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int r = 4 * j + t;
                            float pv = sf_exp2_raw(fmaf(s[r], sc, -kLog2e * l4[t]));
                            if (need_mask) {
                                const int C = qb * 32 + 8 * j + t;
                                if (C < lo || C >= up) pv = 0.f;
                            }
                            s[r] = role == 0 ? pv : pv * (dp[r] - d4[t]);  // P (dV waves) | dS (dK waves)
                        }
                    }
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const sf_v8s f = pack_bf16x8(s, 8 * jp);
#pragma unroll
                        for (int d = 0; d < DB; ++d)
                            acc[d] = sf_mfma32(frag_tr<HD>(lds_x, d, qb * 32 + 16 * jp, fo), f, acc[d]);
                    }
                }
            }
        }
    if (!kok || !block_live) return;
    float* orow = (role == 0 ? p.dv : p.dk) + krow * p.lddk + g * HD;
    const float oscale = role == 0 ? 1.0f : p.scale;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = d * 32 + 8 * j + 4 * hi;
            sf_v4f a = *reinterpret_cast<const sf_v4f*>(orow + col);
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] += acc[d][4 * j + t] * oscale;
            *reinterpret_cast<sf_v4f*>(orow + col) = a;
        }
}

}  // namespace

#define SF_HD_DISPATCH(hd, CALL)                                  \
    do {                                                          \
        if ((hd) == 128) { constexpr int HD = 128; CALL; }        \
        else if ((hd) == 64) { constexpr int HD = 64; CALL; }     \
        else SF_CHECK_ARG(false, "head_dim must be 64 or 128");   \
    } while (0)

constexpr int kAttnWaves = 8;   // waves per workgroup of the dK/dV kernel (4 key sub-blocks x 2 roles)
// fwd / dQ run as 4-wave workgroups, two per CU: one's barrier skew is covered by the other's MFMAs (measured in round 1:
// 0.89 -> 0.77 ms fwd, 0.90 -> 0.79 ms dQ at cfg 2 against one 8-wave workgroup)
constexpr int kAttnFwdWaves = 4;

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

extern "C" int sf_attn_fwd(const void* q, long ldq, const void* k0, long ldk, const void* v0, const void* const* kd,
                           const void* const* vd, int ndiag, const int* kv_len, void* o, long ldo, float* lse, int B,
                           int S, int nh, int nkv, int hd, float scale, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_fwd: bad shape");
    SF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "sf_attn_fwd: row strides must be multiples of 8 (16-byte segments)");
    SF_CHECK_ARG(ndiag >= 0 && ndiag <= kMaxDiag, "sf_attn_fwd: at most 8 diagonal branches");
    AttnFwdArgs p;
    memset(&p, 0, sizeof(p));
    p.q = (const sf_bf16*)q; p.ldq = ldq;
    p.k0 = (const sf_bf16*)k0; p.ldk = ldk;
    p.v0 = (const sf_bf16*)v0;
    for (int i = 0; i < ndiag; ++i) { p.kd[i] = (const sf_bf16*)kd[i]; p.vd[i] = (const sf_bf16*)vd[i]; }
    p.ndiag = ndiag; p.kv_len = kv_len;
    p.o = (sf_bf16*)o; p.ldo = ldo; p.lse = lse;
    p.B = B; p.S = S; p.nh = nh; p.nkv = nkv; p.scale = scale;
    p.dbg = sf_knob("SF_ATTN_DBG", 0);
    p.l2_map = sf_knob("SF_ATTN_L2MAP", 1);
    SF_CHECK_ARG((long)S * ldk * 2 < (1L << 31), "sf_attn_fwd: S * ldk exceeds the 2 GiB range of a buffer descriptor");
    constexpr int NW = kAttnFwdWaves;
    dim3 grid(attn_grid((long)((S + NW * 32 - 1) / (NW * 32)) * nh * B, p.l2_map));
    SF_HD_DISPATCH(hd, SF_ALLOW_SMEM((attn_fwd_kernel<HD, NW>), 2 * 128 * HD * 2);
                   SF_LAUNCH((attn_fwd_kernel<HD, NW>), grid, dim3(NW * 64), 2 * 128 * HD * 2, stream, p));
    return sf_check_launch("sf_attn_fwd");
}

extern "C" int sf_attn_bwd_pre(const void* q, long ldq, const void* o, long ldo, const void* dout, long lddo,
                               const void* const* kd, const void* const* vd, float* const* dkd, float* const* dvd,
                               long ldk, long lddk, int ndiag, const float* lse, float* delta, float* dq_init, int B,
                               int S, int nh, int nkv, int hd, float scale, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_bwd_pre: bad shape");
    SF_CHECK_ARG(ndiag >= 0 && ndiag <= kMaxDiag, "sf_attn_bwd_pre: at most 8 diagonal branches");
    SF_CHECK_ARG(ndiag == 0 || dq_init, "sf_attn_bwd_pre: dq_init required with diagonal branches");
    SF_CHECK_ARG(hd == 64 || hd == 128, "head_dim must be 64 or 128");
    SF_CHECK_ARG(ldq % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldk % 8 == 0 && lddk % 4 == 0,
                 "sf_attn_bwd_pre: strides must be multiples of 8 (16-byte row segments)");
    const int rows_per_block = 64 / (hd / 8);
    const dim3 grid((unsigned)((B * S + rows_per_block - 1) / rows_per_block));
    for (int lo = 0; lo == 0 || lo < ndiag; lo += kPreChunk) {
        AttnBwdPreArgs p;
        memset(&p, 0, sizeof(p));
        p.q = (const sf_bf16*)q; p.ldq = ldq;
        p.o = (const sf_bf16*)o; p.ldo = ldo;
        p.dout = (const sf_bf16*)dout; p.lddo = lddo;
        const int n = ndiag - lo < kPreChunk ? ndiag - lo : kPreChunk;
        for (int i = 0; i < n; ++i) {
            p.kd[i] = (const sf_bf16*)kd[lo + i]; p.vd[i] = (const sf_bf16*)vd[lo + i];
            p.dkd[i] = dkd[lo + i]; p.dvd[i] = dvd[lo + i];
        }
        p.ldk = ldk; p.lddk = lddk; p.ndiag = n > 0 ? n : 0;
        p.lse = lse; p.delta = delta; p.dq_init = dq_init; p.dq_accumulate = lo > 0;
        p.B = B; p.S = S; p.nh = nh; p.nkv = nkv; p.hd = hd; p.scale = scale;
        SF_HD_DISPATCH(hd, SF_LAUNCH((attn_bwd_pre_kernel<HD>), grid, dim3(256), 0, stream, p));
    }
    return sf_check_launch("sf_attn_bwd_pre");
}

static int fill_bwd_args(AttnBwdArgs& p, const void* q, long ldq, const void* dout, long lddo,
                         const void* k0, long ldk, const void* v0, long ldv,
                         const int* kv_len, const float* lse, const float* delta, const float* dq_init, void* dq,
                         long lddq, float* dk, float* dv, long lddk, int B, int S, int nh, int nkv, float scale) {
    memset(&p, 0, sizeof(p));
    p.q = (const sf_bf16*)q; p.ldq = ldq;
    p.dout = (const sf_bf16*)dout; p.lddo = lddo;
    p.k0 = (const sf_bf16*)k0; p.ldk = ldk;
    p.v0 = (const sf_bf16*)v0; p.ldv = ldv;
    p.kv_len = kv_len; p.lse = lse; p.delta = delta; p.dq_init = dq_init;
    p.dq = (sf_bf16*)dq; p.lddq = lddq;
    p.dk = dk; p.dv = dv; p.lddk = lddk;
    p.B = B; p.S = S; p.nh = nh; p.nkv = nkv; p.scale = scale;
    p.l2_map = sf_knob("SF_ATTN_L2MAP", 1);
    return 0;
}

extern "C" int sf_attn_bwd_dq(const void* q, long ldq, const void* dout, long lddo, const void* k0, long ldk,
                              const void* v0, long ldv, const int* kv_len, const float* lse,
                              const float* delta, const float* dq_init, void* dq, long lddq, int B, int S, int nh,
                              int nkv, int hd, float scale, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_bwd_dq: bad shape");
    SF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0,
                 "sf_attn_bwd_dq: row strides must be multiples of 8 (16-byte segments)");
    AttnBwdArgs p;
    fill_bwd_args(p, q, ldq, dout, lddo, k0, ldk, v0, ldv, kv_len, lse, delta, dq_init, dq, lddq,
                  nullptr, nullptr, 0, B, S, nh, nkv, scale);
    SF_CHECK_ARG((long)S * ldk * 2 < (1L << 31) && (long)S * ldv * 2 < (1L << 31),
                 "sf_attn_bwd_dq: S * ld exceeds the 2 GiB range of a buffer descriptor");
    constexpr int NW = kAttnFwdWaves;
    dim3 grid(attn_grid((long)((S + NW * 32 - 1) / (NW * 32)) * nh * B, p.l2_map));
    SF_HD_DISPATCH(hd, SF_ALLOW_SMEM((attn_bwd_dq_kernel<HD, NW>), 2 * 128 * HD * 2);
                   SF_LAUNCH((attn_bwd_dq_kernel<HD, NW>), grid, dim3(NW * 64), 2 * 128 * HD * 2, stream, p));
    return sf_check_launch("sf_attn_bwd_dq");
}

extern "C" int sf_attn_bwd_dkv(const void* q, long ldq, const void* dout, long lddo,
                               const void* k0, long ldk, const void* v0, long ldv, const int* kv_len, const float* lse,
                               const float* delta, float* dk, float* dv, long lddk, int B, int S, int nh, int nkv,
                               int hd, float scale, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_bwd_dkv: bad shape");
    SF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddk % 4 == 0,
                 "sf_attn_bwd_dkv: row strides must be multiples of 8 (16-byte segments)");
    AttnBwdArgs p;
    fill_bwd_args(p, q, ldq, dout, lddo, k0, ldk, v0, ldv, kv_len, lse, delta, nullptr, nullptr, 0, dk,
                  dv, lddk, B, S, nh, nkv, scale);
    SF_CHECK_ARG((long)S * ldq * 2 < (1L << 31) && (long)S * lddo * 2 < (1L << 31),
                 "sf_attn_bwd_dkv: S * ld exceeds the 2 GiB range of a buffer descriptor");
    p.l2_map = sf_knob("SF_ATTN_DKV_L2MAP", 0);   // heaviest-first over ALL pairs wins here (measured: pair-major +20 %)
    constexpr int NW = kAttnWaves;
    dim3 grid(attn_grid((long)((S + NW * 16 - 1) / (NW * 16)) * nkv * B, p.l2_map));  // NW/2 key sub-blocks of 32 keys per workgroup
    SF_HD_DISPATCH(hd, SF_ALLOW_SMEM((attn_bwd_dkv_kernel<HD, NW>), 2 * (128 * HD * 2 + 512));
                   SF_LAUNCH((attn_bwd_dkv_kernel<HD, NW>), grid, dim3(NW * 64), 2 * (128 * HD * 2 + 512), stream, p));
    return sf_check_launch("sf_attn_bwd_dkv");
}
