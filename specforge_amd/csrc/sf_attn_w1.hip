// TTT attention at head_dim 256 (gemma3-1b / qwen3-next-80b-a3b / qwen3.5-35b-a3b recipes): forward and dQ as ONE wave per SIMD,
// slot-planned streams (semantics: sf_attn.hip; reference: specforge/modeling/draft/llama3_eagle.py:745-778, backward blueprint
// 1080-1151).
//
// Why a second pair of kernels.  At head_dim 256 a wave's 32 queries carry 128 registers of output accumulators and 64 (forward) or
// 128 (dQ) registers of Q / dO fragments: two waves per SIMD -- what hides the exponentials and fragment reads of the head_dim <= 128
// kernels behind a partner's MFMAs -- do not fit, and the compiler-scheduled one-wave instantiations of those kernels (round 4) ran
// at 0.19 / 0.23 of the MFMA peak: with a single wave nothing covers a `ds_read; s_waitcnt; v_mfma` chain.  At this head width the
// ratio is favourable to a planned stream instead: 64 (forward) / 96 (dQ) MFMAs of 32 cycles per 64-key tile against ~290 / ~330 other
// instructions (4.5 / 3.4 per MFMA; the head_dim 128 forward carries 8, which is why the same structure LOST there: DESIGN section 4).
// So, as in sf_attn_dkv.hip:
//   * the long-lived MFMA state sits in an asm-owned AGPR bank (AgprBank): O^T / dQ^T accumulators, then the Q (and dO) fragments
//     that are the B operands of S^T = K.Q^T and dP^T = V.dO^T; every MFMA is an asm statement naming those registers;
//   * a tile is one instruction stream in source order -- one MFMA per slot, each followed by its fillers (fragment reads 8 slots
//     ahead of their use, the next tile's LDS-DMA pieces, a piece of the softmax / dS arithmetic) and a scheduling fence;
//   * hazards the compiler cannot see into asm for are met by distance: scores are first read by VALU >= 3 slots after their last
//     MFMA, a packed P / dS fragment is consumed >= 6 slots after its conversion.
// K / V tiles (64 keys: 32 KiB each) are double-buffered in 128 KiB of LDS; the next tile's DMA pieces are fillers of the first slots.
//
// Forward: the online softmax runs per 32-key block (unit) so that the arithmetic of block 0 sits beside the QK^T MFMAs of block 1 and
// that of block 1 beside the PV MFMAs of block 0.  Block 1's exponentials are taken SPECULATIVELY against the running maximum as it
// stands (the deferred rescale tolerates growth up to 2^8); when block 1 raises the maximum by more than that -- the first tiles of a
// row, then practically never -- a slow path at the boundary to its PV phase drains the matrix pipe, rescales O and l, and redoes the
// 16 exponentials.  No cross-tile state, no phantom tiles.
#include "sf_attn_common.h"

using namespace sfattn;

namespace {

constexpr int kAheadEpi = 8;   // transposed V_i fragments in flight in the forward's diagonal-branch epilogue

// ================================================================================================================= forward
// AGPR map (HD = 256): a[0:127] O^T (8 x 16), a[128:191] Q fragments (16 x 4).
template <int HD>
struct Fwd1Bank : AgprBank<HD / 32, HD / 16> {
    static constexpr int KS = HD / 16, DB = HD / 32;
    using Base = AgprBank<DB, KS>;
    template <int I> SF_DEVICE void set_q(sf_v8s v) { Base::template set_b<I>(v); }
    template <int I> SF_DEVICE sf_v8s get_q() { return Base::template get_b<I>(); }
    template <int I, bool FIRST> SF_DEVICE void mfma_s(sf_v16f& s, sf_v8s a) { Base::template mfma_vb<I, FIRST>(s, a); }
    template <int D> SF_DEVICE void mfma_o(sf_v8s a, sf_v8s b) { Base::template mfma_acc<D>(a, b); }
    template <int D> SF_DEVICE sf_v16f get_o() { return Base::template get<D>(); }
    template <int D> SF_DEVICE void axpy_o(float f, const sf_v16f& x) { Base::template axpy<D>(f, x); }
    SF_DEVICE void rescale(float f) {
        static_for<0, DB>([&](auto D) SF_LAMBDA_INLINE { Base::template scale<decltype(D)::value>(f); });
    }
};

// ---- one 64-key tile = two 32-key blocks (units) -----------------------------------------------------------------------------------
//   slots 0 .. KS-1        A0: S^T(block 0) = K.Q^T, k-step = slot          KS .. 2KS-1   A1
//   slots g0 .. g0+NG-1    G0: O^T += V^T.P^T(block 0), slot = (jp, d)      g1 ..         G1          (g0 = 2 KS, NG = 2 DB, g1 = g0 + NG)
//   P(0) beside A1: a1+2, +3 mask + row max | a1+4 decision / rescale | a1+5 .. g0-3 the 16 exponentials, sums, packs
//   P(1) beside G0: g0+2, +3 mask + row max | g0+4 check            | g0+5 .. g1-3 the 16 SPECULATIVE exponentials; fix-up at g1
template <int HD, bool MASK>
struct Fwd1Tile {
    static constexpr int KS = HD / 16, DB = HD / 32, NG = 2 * DB, NSLOT = 2 * KS + 2 * NG, kAhead = SF_ATTN_KAHEAD;
    static constexpr int a1 = KS, g0 = 2 * KS, g1 = g0 + NG;
    static constexpr int p0 = a1 + 2, e0lo = p0 + 3, e0n = (g0 - 2) - e0lo;      // exponentials of block 0: slots e0lo .. e0lo + e0n - 1
    static constexpr int p1 = g0 + 2, e1lo = p1 + 3, e1n = (g1 - 2) - e1lo;
    static_assert(KS >= kAhead && NG >= kAhead && e0n >= 2 && e1n >= 2, "plan: head_dim >= 128");

    const char* lds_k;
    const char* lds_v;
    const FragOff<HD>& fo;
    int hi;
    float sc;
    int rel;                    // last visible key of this lane's query - first key of the tile - 4 * hi
    float& m;                   // running row max (scaled log2 domain, equal on both lanes of a pair) and partial row sum
    float& lpart;
    sf_v16f s[2];               // raw scores of the two blocks
    sf_v16f e1;                 // block 1's exponentials (its raw scores stay in s[1] for the slow path)
    float mt[2], lsum1;
    bool ok1;
    sf_v8s pf[2][2];            // [block][jp]
    sf_v8s rf[kAhead], gf[kAhead];

    template <int N> SF_DEVICE sf_v8s load_row() const { return frag_rows<HD>(lds_k, (N / KS) * 32, N % KS, fo); }      // A-slot N
    template <int M> SF_DEVICE sf_v8s load_g() const {                                                                    // G-slot M
        constexpr int u = M / NG, i = M % NG, jp = i / DB, d = i % DB;
        return frag_tr<HD>(lds_v, d, u * 32 + 16 * jp, fo);
    }
    template <int U, int CH> SF_DEVICE void max_chunk() {       // chunk CH of 2: 8 scores of block U
        constexpr int r0 = CH * 8;
        if (MASK) {                                             // -inf past the last visible key (this chunk's 8 scores)
#pragma unroll
            for (int r = r0; r < r0 + 8; ++r) s[U][r] = (U * 32 + (r & 3) + 8 * (r >> 2) > rel) ? -INFINITY : s[U][r];
        }
        float x = CH == 0 ? s[U][r0] : fmaxf(mt[U], s[U][r0]);
#pragma unroll
        for (int r = 1; r < 8; ++r) x = fmaxf(x, s[U][r0 + r]);
        mt[U] = x;
    }
    template <class Bank> SF_DEVICE void raise_max(Bank& bank, float mts) {     // m <- max(m, mts); O and l follow
        const float mn = fmaxf(m, mts);
        const float alpha = sf_exp2_raw(m - mn);
        m = mn;
        lpart *= alpha;
        bank.rescale(alpha);
    }
    template <class Bank> SF_DEVICE void decide0(Bank& bank) {
        mt[0] = sf_pair_max(mt[0]) * sc;                        // the running max lives in the scaled log2 domain
        // deferred rescale: only when some row's max grew by more than 2^8.  The last G MFMAs (previous tile) were issued >= KS + 4
        // slots ago; the next ones follow >= KS - 4 slots later
        if (!sf_all(mt[0] - m <= 8.0f)) raise_max(bank, mt[0]);
    }
    SF_DEVICE void check1() {
        mt[1] = sf_pair_max(mt[1]) * sc;
        ok1 = sf_all(mt[1] - m <= 8.0f);
        lsum1 = 0.f;
    }
    template <int E> SF_DEVICE void element0() {
        const float e = sf_exp2_raw(fmaf(s[0][E], sc, -m));     // exp2(-inf) == 0 for masked keys
        s[0][E] = e;
        lpart += e;
        if constexpr (E % 8 == 7) pf[0][E / 8] = pack_bf16x8(s[0], E - 7);
    }
    template <int E> SF_DEVICE void element1() {                // against the running max as it stands (see fixup1)
        const float e = sf_exp2_raw(fmaf(s[1][E], sc, -m));
        e1[E] = e;
        lsum1 += e;
        if constexpr (E % 8 == 7) pf[1][E / 8] = pack_bf16x8(e1, E - 7);
    }
    // Boundary G0 | G1.  Block 1 raised the row max by more than 2^8 (wave-uniform; the first tiles of a row, then practically never):
    // G0 -- whose products were formed against the old max, like everything in O -- has to be COMPLETE before O is rescaled, and
    // block 1's exponentials are redone against the new max.
    template <class Bank> SF_DEVICE void fixup1(Bank& bank) {
        if (!ok1) {
            bank.drain();
            raise_max(bank, mt[1]);
            lsum1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                e1[r] = sf_exp2_raw(fmaf(s[1][r], sc, -m));
                lsum1 += e1[r];
            }
            pf[1][0] = pack_bf16x8(e1, 0);
            pf[1][1] = pack_bf16x8(e1, 8);
            bank.drain();       // (the accumulator writes of the rescale and the fresh fragments are behind this for the MFMAs that follow)
        }
        lpart += lsum1;
    }
    template <class Bank, class Dma>
    SF_DEVICE void run(Bank& bank, Dma&& dma_piece) {
        static_for<0, kAhead>([&](auto I) SF_LAMBDA_INLINE { rf[decltype(I)::value] = load_row<decltype(I)::value>(); });
        SF_SCHED_FENCE();
        static_for<0, NSLOT>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int S = decltype(I)::value;
            if constexpr (S == g1) {
                fixup1(bank);
                SF_SCHED_FENCE();
            }
            if constexpr (S < g0) {
                constexpr int u = S / KS, ks = S % KS;
                bank.template mfma_s<ks, ks == 0>(s[u], rf[S % kAhead]);
            } else {
                constexpr int M = S - g0, u = M / NG, i = M % NG, jp = i / DB, d = i % DB;
                bank.template mfma_o<d>(gf[M % kAhead], pf[u][jp]);
            }
            // ---- fillers
            if constexpr (S + kAhead < g0) rf[S % kAhead] = load_row<S + kAhead>();
            if constexpr (S >= g0 - kAhead && S < g0) gf[(S - (g0 - kAhead)) % kAhead] = load_g<S - (g0 - kAhead)>();
            if constexpr (S >= g0) {
                if constexpr (S - g0 + kAhead < 2 * NG) gf[(S - g0) % kAhead] = load_g<S - g0 + kAhead>();
            }
            dma_piece(std::integral_constant<int, S>{});
            if constexpr (S == p0 || S == p0 + 1) max_chunk<0, S - p0>();
            if constexpr (S == p0 + 2) decide0(bank);
            if constexpr (S >= e0lo && S < e0lo + e0n) {
                constexpr int k = S - e0lo, x0 = k * 16 / e0n, x1 = (k + 1) * 16 / e0n;
                static_for<x0, x1>([&](auto E) SF_LAMBDA_INLINE { element0<decltype(E)::value>(); });
            }
            if constexpr (S == p1 || S == p1 + 1) max_chunk<1, S - p1>();
            if constexpr (S == p1 + 2) check1();
            if constexpr (S >= e1lo && S < e1lo + e1n) {
                constexpr int k = S - e1lo, x0 = k * 16 / e1n, x1 = (k + 1) * 16 / e1n;
                static_for<x0, x1>([&](auto E) SF_LAMBDA_INLINE { element1<decltype(E)::value>(); });
            }
            SF_SCHED_FENCE();
        });
    }
};

template <int HD>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) attn_fwd_w1_kernel(AttnFwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, NW = 4, QB = NW * 32, TILE = 128 * HD * 2;
    constexpr int NI = TileStage<HD, 64, NW>::NI, NDMA = 2 * NI;   // DMA pieces per wave and tile
    SF_DYN_SMEM(smem);  // 2 x { K [64][HD], V [64][HD] }
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    // 1-D grid, pair-major (one (batch, kv head) per XCD at a time), heaviest query block first inside a pair: sf_attn.hip
    const int nqb = (p.S + QB - 1) / QB, per_qb = p.nh * p.B;
    int qbi, h, b, g;
    if (p.l2_map) {
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

    Fwd1Bank<HD> bank;
    bank.init();
    {
        sf_v8s qt[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qt[ks] = *reinterpret_cast<const sf_v8s*>(p.q + qrow * p.ldq + h * HD + 16 * ks + 8 * hi);
        // (the asm reads the value: the compiler waits for the loads HERE, ahead of the loop -- see TileStage)
        static_for<0, KS>([&](auto I) SF_LAMBDA_INLINE { bank.template set_q<decltype(I)::value>(qt[decltype(I)::value]); });
    }
    float m = kNegBig, lpart = 0.f;

    const SfBufB kbuf = rows_buf<HD>(p.k0 + (long)b * S * p.ldk + g * HD, p.ldk, S);
    const SfBufB vbuf = rows_buf<HD>(p.v0 + (long)b * S * p.ldk + g * HD, p.ldk, S);
    const unsigned tile_bytes = (unsigned)(64 * p.ldk * 2);
    TileStage<HD, 64, NW> st;
    st.init(p.ldk, wave, lane);
    int kend = qb0 + QB < S ? qb0 + QB : S;  // causal upper bound for this block
    if (kvlen < kend) kend = kvlen;
    const int ntiles = (kend + 63) / 64;
    // piece k of tile kt into buffer kt & 1 (an empty stand-in past the last tile: zeros into the buffer nobody reads any more, so
    // the slot stream has no branch)
    auto piece = [&](int kt, int k) SF_LAMBDA_INLINE {
        const bool real = kt < ntiles;
        char* dst = smem + (kt & 1) * TILE;
        if (k < NI) sf_bufb_glds16(sf_bufb_if(kbuf, real), st.off[k] + (unsigned)kt * tile_bytes, dst + (st.piece0 + k) * 1024);
        else sf_bufb_glds16(sf_bufb_if(vbuf, real), st.off[k - NI] + (unsigned)kt * tile_bytes, dst + 64 * HD * 2 + (st.piece0 + k - NI) * 1024);
    };
    if (ntiles > 0) static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(0, decltype(K)::value); });
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 64;
        sf_wait_vm0();
        sf_syncthreads();               // tile kt landed for everyone; buffer (kt + 1) & 1 is no longer being read
        if (key0 > qw0 + 31) {          // whole tile above this wave's diagonal (wave-uniform): only the staging duty remains
            static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(kt + 1, decltype(K)::value); });
            continue;
        }
        auto dma = [&](auto Sl) SF_LAMBDA_INLINE {       // filler: DMA piece (slot - 1) of tile kt + 1
            constexpr int k = decltype(Sl)::value - 1;
            if constexpr (k >= 0 && k < NDMA) piece(kt + 1, k);
        };
        const char* lds_k = smem + (kt & 1) * TILE;
        const bool need_mask = (key0 + 63 > qw0) || (key0 + 63 >= kvlen);   // wave-uniform
        if (need_mask) {
            Fwd1Tile<HD, true> t{lds_k, lds_k + 64 * HD * 2, fo, hi, sc, lim - key0 - 4 * hi, m, lpart};
            t.run(bank, dma);
        } else {
            Fwd1Tile<HD, false> t{lds_k, lds_k + 64 * HD * 2, fo, hi, sc, 0, m, lpart};
            t.run(bank, dma);
        }
    }
    sf_wait_vm0();   // (the stand-in pieces of the last iteration may still be in flight: LDS must not be reused or released under them)
    bank.drain();

    // diagonal branch terms: one extra key per later TTT step at the query's own position -- for the wave's 32 queries a 32-key block
    // whose score matrix is DIAGONAL.  So a branch runs through the same machinery as a key block: the diagonal scores q_t . k_i,t by
    // VALU dot products, the deferred-rescale decision, P^T = diag(exp2(s - m)) packed to bf16 (the rounding every other probability
    // gets), and O^T += V_i^T . P^T as 2 DB MFMAs on the otherwise idle matrix pipe -- the accumulators never leave the bank.  (Round 4's
    // form, O = O alpha + e V_i in VALU, was ~850 instructions per branch and wave at this head width against ~180 + 16 MFMAs.)
    // A wave only ever needs the K_i / V_i rows of its OWN 32 queries: each wave stages them into a private slice of the (now free)
    // tile buffers -- no workgroup barrier per branch.
    float l = sf_pair_sum(lpart);
    if (p.ndiag > 0) {
        constexpr int PRIV = 2 * TILE / NW;             // bytes of LDS per wave: K_i rows | V_i rows
        constexpr int NG = 2 * DB;
        static_assert(PRIV >= 2 * 32 * HD * 2, "a wave's slice of the tile buffers holds 32 rows of K_i and of V_i");
        char* mine = smem + wave * PRIV;
        const char* vslice = mine + 32 * HD * 2;
        sf_v8s qf[KS];          // the Q fragments, back out of the bank (no global load: nothing here is a load the compiler counts)
        static_for<0, KS>([&](auto I) SF_LAMBDA_INLINE { qf[decltype(I)::value] = bank.template get_q<decltype(I)::value>(); });
        TileStage<HD, 32, 1> ds;
        ds.init(p.ldk, 0, lane);
        constexpr int NIE = TileStage<HD, 32, 1>::NI;   // DMA pieces of a K_i (or V_i) slice
        const unsigned my_rows = (unsigned)((long)qw0 * p.ldk * 2);
        sf_syncthreads();                               // every wave is done with the last K/V tile
        // K_i and V_i are staged separately, each as soon as its half of the slice has been read: K_{i+1} arrives under the V_i work,
        // V_{i+1} under the K_{i+1} wait and dot products.  Past the last branch the same pieces go out against an empty descriptor
        // (zeros into a slice nobody reads), so the counted waits stay exact.
        const long slice = (long)b * S * p.ldk + g * HD;
        auto stage_k = [&](int i) SF_LAMBDA_INLINE {
            const bool real = i < p.ndiag;
            ds.issue(sf_bufb_if(rows_buf<HD>(p.kd[real ? i : 0] + slice, p.ldk, S), real), my_rows, mine);
        };
        auto stage_v = [&](int i) SF_LAMBDA_INLINE {
            const bool real = i < p.ndiag;
            ds.issue(sf_bufb_if(rows_buf<HD>(p.vd[real ? i : 0] + slice, p.ldk, S), real), my_rows, mine + 32 * HD * 2);
        };
        stage_k(0);
        stage_v(0);
        // the diagonal element of the 32 x 32 block in the C layout: key row c sits in register (c & 3) + 4 (c >> 3) of the lane whose
        // half `hi` equals bit 2 of c (crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi)
        const int rsel = ((c >> 2) & 1) == hi ? (c & 3) + 4 * (c >> 3) : -1;
        for (int i = 0; i < p.ndiag; ++i) {
            sf_wait_vmcnt<NIE>();  // K_i landed (V_i may still be in flight)
            sf_wave_lockstep();   // (interpreter only: the other lanes' pieces of this wave's DMA have been copied)
            float dp = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp += dot8(qf[ks], frag_rows<HD>(mine, 0, ks, fo));
            sf_pin(dp);           // every read of the K half has returned
            stage_k(i + 1);
            dp = sf_pair_sum(dp);
            const float s2 = dp * sc;
            if (!sf_all(s2 - m <= 8.0f)) {      // deferred rescale, as in the tile loop (the branch MFMAs before it have to be complete)
                bank.drain();
                const float mn = fmaxf(m, s2);
                const float alpha = sf_exp2_raw(m - mn);
                m = mn;
                l *= alpha;
                bank.rescale(alpha);
                bank.drain();
            }
            const float e = sf_exp2_raw(s2 - m);
            l += e;
            sf_v16f pd;
#pragma unroll
            for (int r = 0; r < 16; ++r) pd[r] = r == rsel ? e : 0.f;
            sf_v8s pf[2] = {pack_bf16x8(pd, 0), pack_bf16x8(pd, 8)};
            sf_wait_vmcnt<NIE>();  // V_i landed (K_{i+1} may still be in flight)
            sf_wave_lockstep();
            sf_v8s gf[kAheadEpi];
            static_for<0, kAheadEpi>([&](auto I) SF_LAMBDA_INLINE {
                constexpr int M = decltype(I)::value;
                gf[M] = frag_tr<HD>(vslice, M % DB, 16 * (M / DB), fo);
            });
            sf_valu_to_mfma(pf[0]);
            sf_valu_to_mfma(pf[1]);
            SF_SCHED_FENCE();
            static_for<0, NG>([&](auto I) SF_LAMBDA_INLINE {
                constexpr int M = decltype(I)::value, jp = M / DB, d = M % DB;
                bank.template mfma_o<d>(gf[M % kAheadEpi], pf[jp]);
                if constexpr (M + kAheadEpi < NG) gf[M % kAheadEpi] = frag_tr<HD>(vslice, (M + kAheadEpi) % DB, 16 * ((M + kAheadEpi) / DB), fo);
                SF_SCHED_FENCE();
            });
            // (every read of the V half has been consumed by an MFMA above: it may be overwritten)
            stage_v(i + 1);
        }
        sf_wait_vm0();          // (the stand-in pieces of the last branch: LDS must not be released under them)
        bank.drain();
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    sf_bf16* orow = p.o + qrow * p.ldo + h * HD;
    static_for<0, DB>([&](auto D) SF_LAMBDA_INLINE {
        constexpr int d = decltype(D)::value;
        const sf_v16f a = bank.template get_o<d>();
        store_row32_bf16(orow + d * 32, hi, qok, [&](int i) { return a[i] * inv; });
    });
    if (!qok) return;
    if (hi == 0) p.lse[((long)b * p.nh + h) * S + qi] = l > 0.f ? (m + log2f(l)) * kLn2 : kNegBig;
}

// ====================================================================================================================== dQ
// AGPR map (HD = 256): a[0:127] dQ^T (8 x 16), a[128:191] Q fragments, a[192:255] dO fragments.
template <int HD>
struct Dq1Bank : AgprBank<HD / 32, 2 * (HD / 16)> {
    static constexpr int KS = HD / 16, DB = HD / 32;
    using Base = AgprBank<DB, 2 * KS>;
    template <int I> SF_DEVICE void set_q(sf_v8s v) { Base::template set_b<I>(v); }
    template <int I> SF_DEVICE void set_do(sf_v8s v) { Base::template set_b<KS + I>(v); }
    template <int I, bool FIRST> SF_DEVICE void mfma_s(sf_v16f& s, sf_v8s a) { Base::template mfma_vb<I, FIRST>(s, a); }
    template <int I, bool FIRST> SF_DEVICE void mfma_dp(sf_v16f& s, sf_v8s a) { Base::template mfma_vb<KS + I, FIRST>(s, a); }
    template <int D> SF_DEVICE void mfma_dq(sf_v8s a, sf_v8s b) { Base::template mfma_acc<D>(a, b); }
    template <int D> SF_DEVICE sf_v16f get_dq() { return Base::template get<D>(); }
};

// ---- one 64-key tile = two 32-key blocks (units) -----------------------------------------------------------------------------------
//   slots 0 .. NA-1      A0: even = S^T(block 0) k-step slot / 2, odd = dP^T                NA .. 2NA-1   A1        (NA = 2 KS)
//   slots g0 ..          G0: dQ^T += K^T.dS^T(block 0), slot = (jp, d)                      g1 ..         G1        (g0 = 2 NA, NG = 2 DB)
//   P(0) (dS of block 0: 16 elements) beside A1 from a1 + 3; P(1) beside G0 from g0 + 3
template <int HD, bool MASK>
struct Dq1Tile {
    static constexpr int KS = HD / 16, DB = HD / 32, NA = 2 * KS, NG = 2 * DB, NSLOT = 2 * NA + 2 * NG, kAhead = SF_ATTN_KAHEAD;
    static constexpr int a1 = NA, g0 = 2 * NA, g1 = g0 + NG;
    static constexpr int p0 = a1 + 3, p0n = NA - 6 < 16 ? NA - 6 : 16;
    static constexpr int p1 = g0 + 3, p1n = NG - 5;
    static_assert(NA >= kAhead && NG >= kAhead && p1n >= 2, "plan: head_dim >= 128");

    const char* lds_k;
    const char* lds_v;
    const FragOff<HD>& fo;
    int hi;
    float sc, lse2, dlt;
    int rel;                    // last visible key of this lane's query - first key of the tile - 4 * hi
    sf_v16f s[2], dp[2];
    sf_v8s ds[2][2];
    sf_v8s rf[kAhead], gf[kAhead];

    template <int N> SF_DEVICE sf_v8s load_row() const {         // A-slot N: even = K rows (S), odd = V rows (dP)
        constexpr int kb = N / NA, i = N % NA;
        return frag_rows<HD>((i & 1) ? lds_v : lds_k, kb * 32, i / 2, fo);
    }
    template <int M> SF_DEVICE sf_v8s load_g() const {           // G-slot M
        constexpr int u = M / NG, i = M % NG, jp = i / DB, d = i % DB;
        return frag_tr<HD>(lds_k, d, u * 32 + 16 * jp, fo);
    }
    template <int U, int E> SF_DEVICE void element() {
        float x = fmaf(s[U][E], sc, -lse2);
        if (MASK) x = (U * 32 + (E & 3) + 8 * (E >> 2) <= rel) ? x : -INFINITY;    // exp2(-inf) == 0: no probability, no dS
        const float pv = sf_exp2_raw(x);
        dp[U][E] = pv * (dp[U][E] - dlt);   // dS^T
        if constexpr (E % 8 == 7) ds[U][E / 8] = pack_bf16x8(dp[U], E - 7);
    }
    template <class Bank, class Dma>
    SF_DEVICE void run(Bank& bank, Dma&& dma_piece) {
        static_for<0, kAhead>([&](auto I) SF_LAMBDA_INLINE { rf[decltype(I)::value] = load_row<decltype(I)::value>(); });
        SF_SCHED_FENCE();
        static_for<0, NSLOT>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int S = decltype(I)::value;
            if constexpr (S < g0) {
                constexpr int u = S / NA, i = S % NA, ks = i / 2;
                if constexpr (i & 1) bank.template mfma_dp<ks, ks == 0>(dp[u], rf[S % kAhead]);
                else bank.template mfma_s<ks, ks == 0>(s[u], rf[S % kAhead]);
            } else {
                constexpr int M = S - g0, u = M / NG, i = M % NG, jp = i / DB, d = i % DB;
                bank.template mfma_dq<d>(gf[M % kAhead], ds[u][jp]);
            }
            // ---- fillers
            if constexpr (S + kAhead < g0) rf[S % kAhead] = load_row<S + kAhead>();
            if constexpr (S >= g0 - kAhead && S < g0) gf[(S - (g0 - kAhead)) % kAhead] = load_g<S - (g0 - kAhead)>();
            if constexpr (S >= g0) {
                if constexpr (S - g0 + kAhead < 2 * NG) gf[(S - g0) % kAhead] = load_g<S - g0 + kAhead>();
            }
            dma_piece(std::integral_constant<int, S>{});
            if constexpr (S >= p0 && S < p0 + p0n) {
                constexpr int k = S - p0, x0 = k * 16 / p0n, x1 = (k + 1) * 16 / p0n;
                static_for<x0, x1>([&](auto E) SF_LAMBDA_INLINE { element<0, decltype(E)::value>(); });
            }
            if constexpr (S >= p1 && S < p1 + p1n) {
                constexpr int k = S - p1, x0 = k * 16 / p1n, x1 = (k + 1) * 16 / p1n;
                static_for<x0, x1>([&](auto E) SF_LAMBDA_INLINE { element<1, decltype(E)::value>(); });
            }
            SF_SCHED_FENCE();
        });
    }
};

template <int HD>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) attn_bwd_dq_w1_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, NW = 4, QB = NW * 32, TILE = 128 * HD * 2;
    constexpr int NI = TileStage<HD, 64, NW>::NI, NDMA = 2 * NI;   // DMA pieces per wave and tile
    SF_DYN_SMEM(smem);  // 2 x { K [64][HD], V [64][HD] }; K serves both S^T = K.Q^T and (transpose-read) dQ^T += K^T.dS^T
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
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

    Dq1Bank<HD> bank;
    bank.init();
    {
        sf_v8s qt[KS], dt[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qt[ks] = *reinterpret_cast<const sf_v8s*>(p.q + qrow * p.ldq + h * HD + 16 * ks + 8 * hi);
            dt[ks] = *reinterpret_cast<const sf_v8s*>(p.dout + qrow * p.lddo + h * HD + 16 * ks + 8 * hi);
        }
        static_for<0, KS>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int i = decltype(I)::value;
            bank.template set_q<i>(qt[i]);      // (the asm reads the value: the compiler waits for the loads HERE, ahead of the loop)
            bank.template set_do<i>(dt[i]);
        });
    }
    sf_pin(lse2);
    sf_pin(dlt);

    const SfBufB kbuf = rows_buf<HD>(p.k0 + (long)b * S * p.ldk + g * HD, p.ldk, S);
    const SfBufB vbuf = rows_buf<HD>(p.v0 + (long)b * S * p.ldv + g * HD, p.ldv, S);
    const unsigned ktile = (unsigned)(64 * p.ldk * 2), vtile = (unsigned)(64 * p.ldv * 2);
    TileStage<HD, 64, NW> stk, stv;
    stk.init(p.ldk, wave, lane);
    stv.init(p.ldv, wave, lane);
    int kend = qb0 + QB < S ? qb0 + QB : S;
    if (kvlen < kend) kend = kvlen;
    const int ntiles = (kend + 63) / 64;
    auto piece = [&](int kt, int k) SF_LAMBDA_INLINE {    // piece k of tile kt into buffer kt & 1 (an empty stand-in past the last tile)
        const bool real = kt < ntiles;
        char* dst = smem + (kt & 1) * TILE;
        if (k < NI) sf_bufb_glds16(sf_bufb_if(kbuf, real), stk.off[k] + (unsigned)kt * ktile, dst + (stk.piece0 + k) * 1024);
        else sf_bufb_glds16(sf_bufb_if(vbuf, real), stv.off[k - NI] + (unsigned)kt * vtile, dst + 64 * HD * 2 + (stv.piece0 + k - NI) * 1024);
    };
    if (ntiles > 0) static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(0, decltype(K)::value); });
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 64;
        sf_wait_vm0();
        sf_syncthreads();               // tile kt landed for everyone; buffer (kt + 1) & 1 is no longer being read
        if (key0 > qw0 + 31) {          // the tile is above this wave's diagonal: only the staging duty remains
            static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(kt + 1, decltype(K)::value); });
            continue;
        }
        auto dma = [&](auto Sl) SF_LAMBDA_INLINE {       // filler: DMA piece (slot - 1) of tile kt + 1
            constexpr int k = decltype(Sl)::value - 1;
            if constexpr (k >= 0 && k < NDMA) piece(kt + 1, k);
        };
        const char* lds_k = smem + (kt & 1) * TILE;
        const bool need_mask = (key0 + 63 > qw0) || (key0 + 63 >= kvlen);   // wave-uniform
        if (need_mask) {
            Dq1Tile<HD, true> t{lds_k, lds_k + 64 * HD * 2, fo, hi, sc, lse2, dlt, lim - key0 - 4 * hi};
            t.run(bank, dma);
        } else {
            Dq1Tile<HD, false> t{lds_k, lds_k + 64 * HD * 2, fo, hi, sc, lse2, dlt, 0};
            t.run(bank, dma);
        }
    }
    sf_wait_vm0();   // (the stand-in pieces of the last iteration may still be in flight: LDS must not be released under them)
    bank.drain();
    sf_bf16* orow = p.dq + qrow * p.lddq + h * HD;     // (dead lanes shadow a valid row and skip the stores: store_row32_bf16 exchanges lanes)
    const float* irow = p.dq_init ? p.dq_init + qrow * ((long)p.nh * HD) + h * HD : nullptr;
    // the diagonal branches' share of dQ (attn_bwd_diag / attn_bwd_pre) joins here; a 32-column block at a time (the accumulators
    // come out of the bank 16 registers at a time: no 128-register staging)
    static_for<0, DB>([&](auto D) SF_LAMBDA_INLINE {
        constexpr int d = decltype(D)::value;
        sf_v4f init[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            init[j] = irow ? *reinterpret_cast<const sf_v4f*>(irow + d * 32 + 8 * j + 4 * hi) : sf_v4f{0.f, 0.f, 0.f, 0.f};
        const sf_v16f a = bank.template get_dq<d>();
        store_row32_bf16(orow + d * 32, hi, qok, [&](int i) { return a[i] * p.scale + init[i >> 2][i & 3]; });
    });
}

}  // namespace

namespace sfattn {

int attn_fwd_w1_launch(const AttnFwdArgs& p, int hd, void* stream) {
    SF_CHECK_ARG(hd == 256, "attn_fwd_w1: head_dim 256 only");
    constexpr int HD = 256;
    const long nqb = (p.S + 127) / 128;     // 128 queries per workgroup
    dim3 grid(p.l2_map ? attn_pair_major_grid(nqb * (p.nh / p.nkv), (long)p.nkv * p.B) : (unsigned)(nqb * p.nh * p.B));
    SF_ALLOW_SMEM((attn_fwd_w1_kernel<HD>), 2 * 128 * HD * 2);
    SF_LAUNCH((attn_fwd_w1_kernel<HD>), grid, dim3(256), 2 * 128 * HD * 2, stream, p);
    return sf_check_launch("sf_attn_fwd");
}

int attn_bwd_dq_w1_launch(const AttnBwdArgs& p, int hd, void* stream) {
    SF_CHECK_ARG(hd == 256, "attn_bwd_dq_w1: head_dim 256 only");
    constexpr int HD = 256;
    const long nqb = (p.S + 127) / 128;
    dim3 grid(p.l2_map ? attn_pair_major_grid(nqb * (p.nh / p.nkv), (long)p.nkv * p.B) : (unsigned)(nqb * p.nh * p.B));
    SF_ALLOW_SMEM((attn_bwd_dq_w1_kernel<HD>), 2 * 128 * HD * 2);
    SF_LAUNCH((attn_bwd_dq_w1_kernel<HD>), grid, dim3(256), 2 * 128 * HD * 2, stream, p);
    return sf_check_launch("sf_attn_bwd_dq");
}

}  // namespace sfattn
