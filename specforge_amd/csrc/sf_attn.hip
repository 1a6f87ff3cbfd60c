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
//   attn_bwd_dkv: dK / dV of block 0 -- sf_attn_dkv.hip
// Tiles are staged HBM -> LDS by 16-byte LDS-DMA with an XOR chunk swizzle on the source
// address (same scheme as sf_gemm.hip).
#include "sf_attn_common.h"

using namespace sfattn;

namespace {

#ifdef SF_ABLATE   // profiling experiments of the tools build: 1 = stage tile 0 only, 2 = skip the MFMA / softmax work
#define SF_ATTN_DBG(p, bit) ((p).dbg & (bit))
#else
#define SF_ATTN_DBG(p, bit) false
#endif

// ------------------------------------------------------------------ forward
// -inf where the tile-relative key index `c` is past `rel` (= last visible key - first key of the block - 4 * hi)
SF_DEVICE void mask_scores(sf_v16f& s, int rel) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) > rel) s[r] = -INFINITY;
}

template <int HD, int NW>
SF_GLOBAL void SF_LAUNCH_BOUNDS(NW * 64, AttnOcc<HD>::kWgPerCu) attn_fwd_kernel(AttnFwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, QB = NW * 32, TILE = 128 * HD * 2;
    SF_DYN_SMEM(smem);  // 2 x { K [64][HD], V [64][HD] }
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    // 1-D grid, heaviest work first: the causal key range grows with the query block, so the LAST query blocks are
    // dispatched first (longest-processing-time order keeps the tail of the launch short)
    const int nqb = (p.S + QB - 1) / QB, per_qb = p.nh * p.B;
    int qbi, h, b, g;
    if (p.l2_map) {   // pair-major: (batch, kv head) -> query block (last = heaviest first) -> query head of the group
        const int nrep = p.nh / p.nkv, W = nrep * nqb;
        const int v = attn_pair_major_index((int)blockIdx.x, W, p.nkv * p.B);
        if (v < 0) return;
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
            if constexpr (HD <= 128) {
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
            // O = O * alpha + e * V_i: per 32-column block one 16-wide fp32 vector expression (v_pk_mul_f32 / v_pk_fma_f32: two
            // columns per instruction), the bf16 values widened pairwise (lo: shift, hi: mask)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                sf_v16f vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef unsigned sf_v2u_ __attribute__((ext_vector_type(2)));
                    const sf_v2u_ w = __builtin_bit_cast(sf_v2u_, vv4[d * 4 + j]);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        vf[4 * j + 2 * h2] = __builtin_bit_cast(float, w[h2] << 16);
                        vf[4 * j + 2 * h2 + 1] = __builtin_bit_cast(float, w[h2] & 0xffff0000u);
                    }
                }
                acc_o[d] = acc_o[d] * alpha + vf * e;
            }
            } else {
            // head_dim 256: the query fragments and the output accumulators are 192 registers of the wave already -- the branch's
            // K_i / V_i rows are consumed from LDS a few registers at a time, and the next branch is staged only after the last read
            float dp = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp += dot8(qf[ks], frag_rows<HD>(mine, 0, ks, fo));
            dp = sf_pair_sum(dp);
            const float s2 = dp * sc;
            const float mn = fmaxf(m, s2);
            const float alpha = sf_exp2(m - mn);
            const float e = sf_exp2(s2 - mn);
            m = mn;
            l = l * alpha + e;
            const char* vrow = mine + 32 * HD * 2 + c * (HD * 2) + 8 * hi;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const sf_v4s vv = *reinterpret_cast<const sf_v4s*>(vrow + (((4 * d + j) ^ swz<HD>(c)) << 4));
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc_o[d][4 * j + t] = acc_o[d][4 * j + t] * alpha + e * sf_bf2f((sf_bf16)vv[t]);
                }
#pragma unroll
            for (int d = 0; d < DB; ++d) sf_pin(acc_o[d]);       // every LDS read of this branch has returned
            if (i + 1 < p.ndiag) stage_diag(i + 1);
            }
        }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    sf_bf16* orow = p.o + qrow * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d) store_row32_bf16(orow + d * 32, hi, qok, [&](int i) { return acc_o[d][i] * inv; });
    if (!qok) return;
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
    int last;           // index (in this launch) of the diagonal whose sums are FINAL after this launch, or -1: they leave as bf16
    sf_bf16* dk_last; sf_bf16* dv_last; long ld_last;   // ... to dk_last / dv_last [B*S, nkv*hd] instead of back to dkd / dvd
    int B, S, nh, nkv, hd;
    float scale;
};

// A group of HD/8 lanes owns one token row (8 consecutive d per lane, 16-byte loads), so a wave covers 4 (HD=128)
// or 8 (HD=64) rows; wave w handles kv groups w, w+4, ... and walks the query heads of a group serially, which
// keeps the dK_i / dV_i sums over those heads in registers.  Every dot product is an all-reduce over the lane group
// done with DPP row operations (sf_row_sum) -- no LDS crossbar traffic.  One launch handles at most ND
// diagonals (their K/V/dK/dV slices live in registers); the host splits longer lists into chunks, later chunks
// accumulate into dq_init.
// ND = diagonals per launch.  Measured and rejected (round 3): an 8-wide instantiation that takes 5 .. 8 diagonals in ONE launch
// (saves a second read of q / o / dO and an fp32 read-modify-write of dq_init: 56 KB of 216 per row at 6 diagonals) needs 349
// registers = one wave per SIMD, and the streaming kernel then runs no faster than two 4-wide launches (0.660 vs 0.669 ms).
template <int HD, int ND>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2) attn_bwd_pre_kernel(AttnBwdPreArgs p) {
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
                    if (i == p.last) {   // every TTT step that reads this branch has contributed: round once, write the gradient
                        const long oi = (long)row * p.ld_last + g * HD + d0;
                        SfVec8<sf_bf16>::st(p.dk_last + oi, a);
                        SfVec8<sf_bf16>::st(p.dv_last + oi, c);
                    } else {
                        SfVec8<float>::st(p.dkd[i] + idx, a);
                        SfVec8<float>::st(p.dvd[i] + idx, c);
                    }
                }
        }
    }
}

// ------------------------------------------- backward: diagonal branches, blocked (round 4)
// The pair (step k, branch i <= k) of the TTT diagonal terms contributes  dq_k += ds k_i,  dK_i += ds q_k,  dV_i += p dO_k  with
// p = exp(q_k.k_i scale - lse_k), ds = p (dO_k.v_i - delta_k) scale -- all per token.  dq_k is needed at sweep step k and dK_i / dV_i
// at sweep step i (the QKV input gradient of step i feeds step i - 1), so a pair may run at any sweep step in [i, k].  attn_bwd_pre
// runs every pair at step k: the fp32 sums of branch i are read and written back once per later step (21 times at ttt 7: 16 KB
// per token and pair, 5.5 GB of the kernel's 12.6 GB per micro-step), plus their zero fill, plus a second launch re-reading q / o /
// dO whenever a step has more than 4 branches.  Here a launch serves sweep step s with
//   * the step's own q / o / dO: delta, and dq_init from ALL nread <= 6 branches read (their k_i / v_i: 4 KB per branch and token);
//   * the first nacc <= 4 of those branches accumulating dK / dV in registers, over the own step AND over nx <= 8 LATER steps
//     whose q / dO / lse / delta are streamed again (16 KB per step and token, p and ds recomputed) -- so a block of 4 branches takes
//     all the steps above it in ONE pass, first touch (no read, no zero fill), and only the pairs inside a block remain
//     read-modify-write.  A branch whose last contributor is this launch leaves as bf16 (one rounding), into the dqkv slot.
// engine.py (diag_plan) builds the launches; bytes of the diagonal terms per token at ttt 7: 828 KB -> 516 KB.
constexpr int kDiagRead = 6, kDiagAcc = 4, kDiagX = 8;
// row sets per 4-wave workgroup of the diagonal-branch kernel: 1 (a wave per kv group, round-robin) unless there are fewer groups than waves
SF_HD int attn_diag_row_sets(int nkv) { return nkv == 1 ? 4 : nkv == 2 ? 2 : 1; }
struct AttnBwdDiagArgs {
    const sf_bf16* q; long ldq;        // own step (null: this launch only streams later steps into the accumulating branches)
    const sf_bf16* o; long ldo;
    const sf_bf16* dout; long lddo;
    const float* lse; float* delta; float* dq_init; int dq_accumulate;
    const sf_bf16* kd[kDiagRead]; const sf_bf16* vd[kDiagRead]; long ldk; int nread;
    float* dkd[kDiagAcc]; float* dvd[kDiagAcc]; long lddk; int nacc;
    int first[kDiagAcc];               // 1: nothing to add to (first touch): the fp32 sums are not read
    sf_bf16* dk_out[kDiagAcc]; sf_bf16* dv_out[kDiagAcc]; long ld_out;   // non-null: final -> bf16 here instead of fp32 back to dkd / dvd
    const sf_bf16* xq[kDiagX]; const sf_bf16* xdo[kDiagX]; const float* xlse[kDiagX]; const float* xdelta[kDiagX]; int nx;
    int B, S, nh, nkv;
    float scale;
};

template <int HD>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2) attn_bwd_diag_kernel(AttnBwdDiagArgs p) {
    constexpr int LPH = HD / 8;     // lanes per row
    constexpr int RPW = 64 / LPH;   // rows per wave
    constexpr int NR = kDiagRead, NA = kDiagAcc;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int sub = lane / LPH, li = lane % LPH;
    const int N = p.B * p.S;
    // wave w takes kv groups w, w + 4, ...; with FEWER than 4 groups (nkv 1 or 2: the head_dim-256 recipes) the spare waves take further
    // rows instead of idling (half / three quarters of a workgroup did: 0.32 of the HBM rate at nkv 2 where nkv >= 4 reaches 0.58)
    const int rsets = attn_diag_row_sets(p.nkv), gstep = 4 / rsets;
    const int row_raw = ((int)blockIdx.x * rsets + wave / gstep) * RPW + sub;
    const bool live = row_raw < N;
    const int row = live ? row_raw : N - 1;   // dead lane groups shadow a valid row and skip every store
    const int b = row / p.S, t = row - b * p.S;
    const int nrep = p.nh / p.nkv;
    const int d0 = li * 8;
    for (int g = wave % gstep; g < p.nkv; g += gstep) {
        sf_v8s kv[NR], vv[NR];
        float dk[NA][8], dv[NA][8];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (i < p.nread) {
                kv[i] = *reinterpret_cast<const sf_v8s*>(p.kd[i] + (long)row * p.ldk + g * HD + d0);
                vv[i] = *reinterpret_cast<const sf_v8s*>(p.vd[i] + (long)row * p.ldk + g * HD + d0);
            } else {
                kv[i] = sf_v8s{0, 0, 0, 0, 0, 0, 0, 0};
                vv[i] = kv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { dk[i][e] = 0.f; dv[i][e] = 0.f; }
        // ---- the launch's own step: delta, dq_init over every branch read, dK / dV of the accumulating ones.  The kernel is bound by
        // load latency, not bandwidth (8 waves per CU, one 16-byte load per lane and operand in flight): two query heads are loaded
        // per trip (6 loads in flight instead of 3), then both are worked through
        if (p.q) {
            auto own_head = [&](int h, const float (&qv)[8], const float (&ov)[8], const float (&dov)[8]) SF_LAMBDA_INLINE {
                const int col = h * HD + d0;
                float dl = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += ov[e] * dov[e];
                dl = sf_row_sum<LPH>(dl);
                const long lidx = ((long)b * p.nh + h) * p.S + t;
                if (li == 0 && live) p.delta[lidx] = dl;
                if (p.nread > 0) {
                    const float lse = p.lse[lidx];
                    float dq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) dq[e] = 0.f;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (i < p.nread) {
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
                            for (int e = 0; e < 8; ++e) dq[e] += ds * sf_bf2f((sf_bf16)kv[i][e]);
                            if (i < NA) {      // (compile-time: register slots; i < p.nacc at run time)
                                if (i < p.nacc) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) {
                                        dk[i < NA ? i : 0][e] += ds * qv[e];
                                        dv[i < NA ? i : 0][e] += pi * dov[e];
                                    }
                                }
                            }
                        }
                    }
                    if (live && p.dq_init) {
                        float* dqp = p.dq_init + (long)row * (p.nh * HD) + col;
                        if (p.dq_accumulate) {   // a later chunk of this step's branches: add to what the first chunk wrote
                            float prev[8];
                            SfVec8<float>::ld(dqp, prev);
#pragma unroll
                            for (int e = 0; e < 8; ++e) dq[e] += prev[e];
                        }
                        SfVec8<float>::st(dqp, dq);
                    }
                }
            };
            for (int hh = 0; hh < nrep; hh += 2) {
                const int h = g * nrep + hh;
                const bool two = hh + 1 < nrep;
                const int h2 = two ? h + 1 : h;         // (odd group size: the second slot re-reads the first head and is not applied)
                float qa[8], oa[8], da[8];
                SfVec8<sf_bf16>::ld(p.q + (long)row * p.ldq + h * HD + d0, qa);
                SfVec8<sf_bf16>::ld(p.o + (long)row * p.ldo + h * HD + d0, oa);
                SfVec8<sf_bf16>::ld(p.dout + (long)row * p.lddo + h * HD + d0, da);
                const sf_v8s qp = *reinterpret_cast<const sf_v8s*>(p.q + (long)row * p.ldq + h2 * HD + d0);     // second head: kept packed
                const sf_v8s op = *reinterpret_cast<const sf_v8s*>(p.o + (long)row * p.ldo + h2 * HD + d0);
                const sf_v8s dp = *reinterpret_cast<const sf_v8s*>(p.dout + (long)row * p.lddo + h2 * HD + d0);
                own_head(h, qa, oa, da);
                if (two) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { qa[e] = sf_bf2f((sf_bf16)qp[e]); oa[e] = sf_bf2f((sf_bf16)op[e]); da[e] = sf_bf2f((sf_bf16)dp[e]); }
                    own_head(h2, qa, oa, da);
                }
            }
        }
        // ---- later TTT steps streamed into the accumulating branches (their dq / delta were finished at their own sweep step)
        for (int x = 0; x < p.nx; ++x) {
            const sf_bf16* xq = p.xq[x];
            const sf_bf16* xdo = p.xdo[x];
            const float* xl = p.xlse[x];
            const float* xd = p.xdelta[x];
            auto x_head = [&](int h, const float (&qv)[8], const float (&dov)[8]) SF_LAMBDA_INLINE {
                const long lidx = ((long)b * p.nh + h) * p.S + t;
                const float lse = xl[lidx], dl = xd[lidx];
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    if (i < p.nacc) {
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
                            dk[i][e] += ds * qv[e];
                            dv[i][e] += pi * dov[e];
                        }
                    }
                }
            };
            for (int hh = 0; hh < nrep; hh += 2) {
                const int h = g * nrep + hh;
                const bool two = hh + 1 < nrep;
                const int h2 = two ? h + 1 : h;
                float qa[8], da[8];
                SfVec8<sf_bf16>::ld(xq + (long)row * p.ldq + h * HD + d0, qa);
                SfVec8<sf_bf16>::ld(xdo + (long)row * p.lddo + h * HD + d0, da);
                const sf_v8s qp = *reinterpret_cast<const sf_v8s*>(xq + (long)row * p.ldq + h2 * HD + d0);
                const sf_v8s dp = *reinterpret_cast<const sf_v8s*>(xdo + (long)row * p.lddo + h2 * HD + d0);
                x_head(h, qa, da);
                if (two) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { qa[e] = sf_bf2f((sf_bf16)qp[e]); da[e] = sf_bf2f((sf_bf16)dp[e]); }
                    x_head(h2, qa, da);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (i < p.nacc) {
                    const long idx = (long)row * p.lddk + g * HD + d0;
                    float a[8], c[8];
                    if (p.first[i]) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { a[e] = dk[i][e]; c[e] = dv[i][e]; }
                    } else {
                        SfVec8<float>::ld(p.dkd[i] + idx, a);
                        SfVec8<float>::ld(p.dvd[i] + idx, c);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { a[e] += dk[i][e]; c[e] += dv[i][e]; }
                    }
                    if (p.dk_out[i]) {   // every TTT step that reads this branch has contributed: round once, write the gradient
                        const long oi = (long)row * p.ld_out + g * HD + d0;
                        SfVec8<sf_bf16>::st(p.dk_out[i] + oi, a);
                        SfVec8<sf_bf16>::st(p.dv_out[i] + oi, c);
                    } else {
                        SfVec8<float>::st(p.dkd[i] + idx, a);
                        SfVec8<float>::st(p.dvd[i] + idx, c);
                    }
                }
        }
    }
}

// ------------------------------------------------------------- backward: dQ

template <int HD, int NW>
SF_GLOBAL void SF_LAUNCH_BOUNDS(NW * 64, AttnOcc<HD>::kWgPerCu) attn_bwd_dq_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, QB = NW * 32, TILE = 128 * HD * 2;
    SF_DYN_SMEM(smem);  // 2 x { K [64][HD], V [64][HD] }; K serves both S^T = K.Q^T and (transpose-read) dQ^T += K^T.dS^T
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    // 1-D grid, heaviest work first: the causal key range grows with the query block, so the LAST query blocks are
    // dispatched first (longest-processing-time order keeps the tail of the launch short)
    const int nqb = (p.S + QB - 1) / QB, per_qb = p.nh * p.B;
    int qbi, h, b, g;
    if (p.l2_map) {   // pair-major: (batch, kv head) -> query block (last = heaviest first) -> query head of the group
        const int nrep = p.nh / p.nkv, W = nrep * nqb;
        const int v = attn_pair_major_index((int)blockIdx.x, W, p.nkv * p.B);
        if (v < 0) return;
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
        store_row32_bf16(orow + d * 32, hi, qok, [&](int i) { return acc[d][i] * p.scale + init[d * 4 + (i >> 2)][i & 3]; });
}

}  // namespace

#if defined(SF_ABLATE) && !defined(SF_EMU)
#include "../../tools/experiments/sf_attn_w4_variants.inc"
#endif

constexpr int kAttnFwdWaves = 4;

extern "C" int sf_attn_fwd(const void* q, long ldq, const void* k0, long ldk, const void* v0, const void* const* kd,
                           const void* const* vd, int ndiag, const int* kv_len, void* o, long ldo, float* lse, int B,
                           int S, int nh, int nkv, int hd, float scale, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_fwd: bad shape");
    SF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "sf_attn_fwd: row strides must be multiples of 8 (16-byte segments)");
    SF_CHECK_ARG(ndiag >= 0 && ndiag <= kMaxDiag, "sf_attn_fwd: at most 32 diagonal branches");
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
    const long nqb = (S + NW * 32 - 1) / (NW * 32);
    dim3 grid(p.l2_map ? attn_pair_major_grid(nqb * (nh / nkv), (long)nkv * B) : (unsigned)(nqb * nh * B));
#if defined(SF_ABLATE) && !defined(SF_EMU)
    if (sf_knob("SF_ATTN_FWD_W4", 0))   // tools build: the measured-and-rejected one-wave-per-SIMD variant
        return sfattn_w4::attn_fwd(q, ldq, k0, ldk, v0, kd, vd, ndiag, kv_len, o, ldo, lse, B, S, nh, nkv, hd, scale, stream);
#endif
    // head_dim 256: one wave per SIMD, slot-planned (sf_attn_w1.hip); the generic instantiation stays as the tools build's A/B
    if (hd == 256 && sf_knob("SF_ATTN_W1", 1)) return attn_fwd_w1_launch(p, hd, stream);
    SF_HD_DISPATCH(hd, SF_ALLOW_SMEM((attn_fwd_kernel<HD, NW>), 2 * 128 * HD * 2);
                   SF_LAUNCH((attn_fwd_kernel<HD, NW>), grid, dim3(NW * 64), 2 * 128 * HD * 2, stream, p));
    return sf_check_launch("sf_attn_fwd");
}

extern "C" int sf_attn_bwd_pre(const void* q, long ldq, const void* o, long ldo, const void* dout, long lddo,
                               const void* const* kd, const void* const* vd, float* const* dkd, float* const* dvd,
                               long ldk, long lddk, int ndiag, const float* lse, float* delta, float* dq_init, int B,
                               int S, int nh, int nkv, int hd, float scale, void* dk_last, void* dv_last, long ld_last,
                               void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_bwd_pre: bad shape");
    SF_CHECK_ARG(ndiag >= 0 && ndiag <= kMaxDiag, "sf_attn_bwd_pre: at most 32 diagonal branches");
    SF_CHECK_ARG(ndiag == 0 || dq_init, "sf_attn_bwd_pre: dq_init required with diagonal branches");
    SF_CHECK_ARG((!dk_last && !dv_last) || (dk_last && dv_last && ndiag > 0 && ld_last % 8 == 0),
                 "sf_attn_bwd_pre: dk_last / dv_last come together, need a diagonal branch and 16-byte aligned rows");
    SF_CHECK_ARG(hd == 64 || hd == 128 || hd == 256, "head_dim must be 64, 128 or 256");
    SF_CHECK_ARG(ldq % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldk % 8 == 0 && lddk % 4 == 0,
                 "sf_attn_bwd_pre: strides must be multiples of 8 (16-byte row segments)");
    const int rows_per_block = 64 / (hd / 8);
    const dim3 grid((unsigned)((B * S + rows_per_block - 1) / rows_per_block));
    constexpr int chunk = 4;
    for (int lo = 0; lo == 0 || lo < ndiag; lo += chunk) {
        AttnBwdPreArgs p;
        memset(&p, 0, sizeof(p));
        p.q = (const sf_bf16*)q; p.ldq = ldq;
        p.o = (const sf_bf16*)o; p.ldo = ldo;
        p.dout = (const sf_bf16*)dout; p.lddo = lddo;
        const int n = ndiag - lo < chunk ? ndiag - lo : chunk;
        for (int i = 0; i < n; ++i) {
            p.kd[i] = (const sf_bf16*)kd[lo + i]; p.vd[i] = (const sf_bf16*)vd[lo + i];
            p.dkd[i] = dkd[lo + i]; p.dvd[i] = dvd[lo + i];
        }
        p.ldk = ldk; p.lddk = lddk; p.ndiag = n > 0 ? n : 0;
        p.lse = lse; p.delta = delta; p.dq_init = dq_init; p.dq_accumulate = lo > 0;
        p.last = (dk_last && lo + n == ndiag) ? n - 1 : -1;
        p.dk_last = (sf_bf16*)dk_last; p.dv_last = (sf_bf16*)dv_last; p.ld_last = ld_last;
        p.B = B; p.S = S; p.nh = nh; p.nkv = nkv; p.hd = hd; p.scale = scale;
        SF_HD_DISPATCH(hd, SF_LAUNCH((attn_bwd_pre_kernel<HD, chunk>), grid, dim3(256), 0, stream, p));
    }
    return sf_check_launch("sf_attn_bwd_pre");
}

extern "C" int sf_attn_bwd_diag(const void* q, long ldq, const void* o, long ldo, const void* dout, long lddo, const float* lse,
                                float* delta, float* dq_init, int dq_accumulate, const void* const* kd, const void* const* vd,
                                long ldk, int nread, float* const* dkd, float* const* dvd, long lddk, int nacc,
                                const int* first, void* const* dk_out, void* const* dv_out, long ld_out, const void* const* xq,
                                const void* const* xdo, const float* const* xlse, const float* const* xdelta, int nx, int B, int S,
                                int nh, int nkv, int hd, float scale, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_bwd_diag: bad shape");
    SF_CHECK_ARG(nread >= 0 && nread <= kDiagRead && nacc >= 0 && nacc <= kDiagAcc && nacc <= nread && nx >= 0 && nx <= kDiagX,
                 "sf_attn_bwd_diag: at most 6 branches read, the first <= 4 of them accumulating, <= 8 later steps streamed");
    SF_CHECK_ARG(hd == 64 || hd == 128 || hd == 256, "head_dim must be 64, 128 or 256");
    SF_CHECK_ARG(q || (nx > 0 && nacc > 0), "sf_attn_bwd_diag: nothing to do (no own step and nothing streamed)");
    SF_CHECK_ARG(!q || (o && dout && delta && (nread == 0 || (lse && dq_init))), "sf_attn_bwd_diag: own step needs o, dout, delta (and lse, dq_init with branches)");
    SF_CHECK_ARG(nx == 0 || (xq && xdo && xlse && xdelta && nacc > 0), "sf_attn_bwd_diag: streamed steps need q / dO / lse / delta and an accumulating branch");
    SF_CHECK_ARG(ldq % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldk % 8 == 0 && lddk % 4 == 0 && ld_out % 8 == 0,
                 "sf_attn_bwd_diag: strides must be multiples of 8 (16-byte row segments)");
    AttnBwdDiagArgs p;
    memset(&p, 0, sizeof(p));
    p.q = (const sf_bf16*)q; p.ldq = ldq;
    p.o = (const sf_bf16*)o; p.ldo = ldo;
    p.dout = (const sf_bf16*)dout; p.lddo = lddo;
    p.lse = lse; p.delta = delta; p.dq_init = dq_init; p.dq_accumulate = dq_accumulate;
    for (int i = 0; i < nread; ++i) {
        SF_CHECK_ARG(kd && vd && kd[i] && vd[i], "sf_attn_bwd_diag: null branch");
        p.kd[i] = (const sf_bf16*)kd[i]; p.vd[i] = (const sf_bf16*)vd[i];
    }
    p.ldk = ldk; p.nread = nread;
    for (int i = 0; i < nacc; ++i) {
        const bool fin = dk_out && dk_out[i];
        SF_CHECK_ARG(first && dkd && dvd && (fin ? (dv_out && dv_out[i]) != 0 : true), "sf_attn_bwd_diag: accumulating branch needs its buffers");
        SF_CHECK_ARG((fin && first[i]) || (dkd[i] && dvd[i]), "sf_attn_bwd_diag: fp32 sums missing (needed unless first touch AND final)");
        p.dkd[i] = dkd[i]; p.dvd[i] = dvd[i]; p.first[i] = first[i] ? 1 : 0;
        p.dk_out[i] = fin ? (sf_bf16*)dk_out[i] : nullptr; p.dv_out[i] = fin ? (sf_bf16*)dv_out[i] : nullptr;
    }
    p.lddk = lddk; p.nacc = nacc; p.ld_out = ld_out;
    for (int x = 0; x < nx; ++x) {
        SF_CHECK_ARG(xq[x] && xdo[x] && xlse[x] && xdelta[x], "sf_attn_bwd_diag: null streamed step");
        p.xq[x] = (const sf_bf16*)xq[x]; p.xdo[x] = (const sf_bf16*)xdo[x]; p.xlse[x] = xlse[x]; p.xdelta[x] = xdelta[x];
    }
    p.nx = nx;
    p.B = B; p.S = S; p.nh = nh; p.nkv = nkv; p.scale = scale;
    const int rows_per_block = 64 / (hd / 8) * attn_diag_row_sets(nkv);
    const dim3 grid((unsigned)((B * S + rows_per_block - 1) / rows_per_block));
    SF_HD_DISPATCH(hd, SF_LAUNCH((attn_bwd_diag_kernel<HD>), grid, dim3(256), 0, stream, p));
    return sf_check_launch("sf_attn_bwd_diag");
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
#if defined(SF_ABLATE) && !defined(SF_EMU)
    if (sf_knob("SF_ATTN_DQ_W4", 0))    // tools build: the measured-and-rejected one-wave-per-SIMD variant
        return sfattn_w4::attn_bwd_dq(q, ldq, dout, lddo, k0, ldk, v0, ldv, kv_len, lse, delta, dq_init, dq, lddq, B, S, nh, nkv, hd,
                                      scale, stream);
#endif
    AttnBwdArgs p;
    fill_bwd_args(p, q, ldq, dout, lddo, k0, ldk, v0, ldv, kv_len, lse, delta, dq_init, dq, lddq,
                  nullptr, nullptr, 0, B, S, nh, nkv, scale);
    SF_CHECK_ARG((long)S * ldk * 2 < (1L << 31) && (long)S * ldv * 2 < (1L << 31),
                 "sf_attn_bwd_dq: S * ld exceeds the 2 GiB range of a buffer descriptor");
    constexpr int NW = kAttnFwdWaves;
    const long nqb = (S + NW * 32 - 1) / (NW * 32);
    dim3 grid(p.l2_map ? attn_pair_major_grid(nqb * (nh / nkv), (long)nkv * B) : (unsigned)(nqb * nh * B));
    if (hd == 256 && sf_knob("SF_ATTN_W1", 1)) return attn_bwd_dq_w1_launch(p, hd, stream);    // (sf_attn_w1.hip)
    SF_HD_DISPATCH(hd, SF_ALLOW_SMEM((attn_bwd_dq_kernel<HD, NW>), 2 * 128 * HD * 2);
                   SF_LAUNCH((attn_bwd_dq_kernel<HD, NW>), grid, dim3(NW * 64), 2 * 128 * HD * 2, stream, p));
    return sf_check_launch("sf_attn_bwd_dq");
}
