// TTT attention backward: dK and dV of the step-0 keys / values (reference blueprint: _FlashCachedMergeFunc.backward,
// specforge/modeling/draft/llama3_eagle.py:1080-1151; semantics as in sf_attn.hip).
//
// One workgroup = 4 waves = 128 keys of one (batch, kv head); ONE wave per SIMD with the whole 512-entry register file
// (__launch_bounds__(256, 1)).  A wave owns 32 keys and BOTH gradients of those keys; it loops over the group's query
// heads and the 64-query tiles at / after its keys.  Per tile and wave, for the two 32-query blocks qb = 0, 1:
//     A(qb): S = Q.K^T, dP = dO.V^T                      16 MFMAs (32x32x16 bf16), A fragments = rows of the Q / dO tile
//     P(qb): P = exp2(S sc - lse log2e), dS = P (dP - delta), both packed to bf16          ~100 VALU
//     G(qb): dV^T += dO^T.P, dK^T += Q^T.dS              16 MFMAs, A fragments by transpose read of the same tiles
// i.e. 64 MFMAs, each product once.  (Round 2 gave the two gradients to two waves of a SIMD, each recomputing S: 80 MFMAs
// per tile where 64 do, the lighter role idle a third of the time, 0.23 of the MFMA peak.)
//
// What hipcc made of the obvious C++ for this structure decided the shape of this file.  With builtin MFMAs and 128
// accumulator registers per wave the register allocator kept the accumulators in architectural VGPRs, spilled fragment
// addresses to AGPRs and copied ~290 registers through v_accvgpr_read / _write per tile, and the pre-RA scheduler --
// pressure-bound -- serialised every fragment read (ds_read; s_waitcnt lgkmcnt(0); v_mfma; ds_read; ...).  So:
//   * the wave's long-lived MFMA state lives in asm-owned AGPRs the compiler never sees: dV^T a[0:16 DB), dK^T next,
//     then the K and V fragments (B operands of S / dP).  Every MFMA is an asm statement naming those registers; the
//     compiler allocates only what VALU touches (S, dP, P, dS, addresses) and the A fragments in flight -- about 200
//     registers, no spills, no copies (tests/test_isa_invariants.py audits the ISA: no compiler v_accvgpr_*, no scratch);
//   * the tile body is ONE instruction stream in source order, a slot plan like the GEMMs': 64 MFMA slots, each followed by
//     its fillers (fragment reads 8 slots ahead of their use, the next-but-one tile's LDS-DMA pieces, one piece of the
//     P / dS arithmetic) and a scheduling barrier, so the stream the hardware sees is the stream written here.  The
//     arithmetic of block 0 sits beside the S / dP MFMAs of block 1, that of block 1 beside the gradient MFMAs of block 0;
//   * hazards the compiler cannot see into asm for are satisfied by distance, not nops: S / dP of a block are first read
//     by VALU three MFMAs after their last MFMA; a packed P / dS fragment is first consumed >= 6 slots after its cvt.
// Q / dO / lse / delta tiles arrive through a 3-deep LDS ring, two tiles ahead, behind a COUNTED vmcnt (nothing in the
// loop is a load the compiler counts: see TileStage in sf_attn_common.h).
#include "sf_attn_common.h"

using namespace sfattn;

namespace {

// AGPR map (HD = 128): a[0:63] dV^T (4 x 16), a[64:127] dK^T, a[128:159] K fragments (8 x 4), a[160:191] V fragments.
template <int HD>
struct DkvBank : AgprBank<2 * (HD / 32), 2 * (HD / 16)> {
    static constexpr int KS = HD / 16, DB = HD / 32;
    using Base = AgprBank<2 * DB, 2 * KS>;
    template <int I> SF_DEVICE void set_k(sf_v8s v) { Base::template set_b<I>(v); }
    template <int I> SF_DEVICE void set_v(sf_v8s v) { Base::template set_b<KS + I>(v); }
    template <int I, bool FIRST> SF_DEVICE void mfma_s(sf_v16f& s, sf_v8s a) { Base::template mfma_vb<I, FIRST>(s, a); }
    template <int I, bool FIRST> SF_DEVICE void mfma_dp(sf_v16f& s, sf_v8s a) { Base::template mfma_vb<KS + I, FIRST>(s, a); }
    template <int D> SF_DEVICE void mfma_dv(sf_v8s a, sf_v8s b) { Base::template mfma_acc<D>(a, b); }
    template <int D> SF_DEVICE void mfma_dk(sf_v8s a, sf_v8s b) { Base::template mfma_acc<DB + D>(a, b); }
    template <int D> SF_DEVICE sf_v16f get_dv() { return Base::template get<D>(); }
    template <int D> SF_DEVICE sf_v16f get_dk() { return Base::template get<DB + D>(); }
};

// ---- one 64-query tile ----------------------------------------------------------------------------------------------
// Slot s of 64 = one MFMA + its fillers.  A fragment consumed in slot s is read in slot s - kAhead (slots < kAhead: in the
// burst in front of slot 0).
//   slots  0..15  A(0): even = S (k-step s/2), odd = dP            16..31  A(1)
//   slots 32..47  G(0): (jp, d): even = dV, odd = dK               48..63  G(1)
//   fillers: DMA pieces of tile it+2 in slots 1..NDMA;  lse / delta reads of block 0 in slots 8..15, of block 1 in 24..31;
//            P(0) element e in slot 18 + e * 14 / 16 (18..31), P(1) in 34..47; each element = 5 VALU (+3 when masked)
template <int HD, bool MASK>
struct DkvTile {
    static constexpr int KS = HD / 16, DB = HD / 32, NA = 2 * KS, NG = 4 * DB, NSLOT = 2 * NA + 2 * NG, kAhead = SF_ATTN_KAHEAD;
    static constexpr int P0 = NA + 2, P1 = 2 * NA + 2, PSPAN = NG - 2 < NA - 2 ? NG - 2 : NA - 2;
    static_assert(NA == NG, "the plan assumes the score and gradient phases have the same number of MFMAs");

    const char* lds_q;
    const char* lds_do;
    const float* lds_lse;
    const float* lds_dlt;
    const FragOff<HD>& fo;
    int hi;
    float sc;
    int lo, up;
    sf_v16f s[2], dp[2];
    sf_v4f l4[2][4], d4[2][4];
    sf_v8s pf[2][2], df[2][2];
    sf_v8s fr[kAhead];          // A fragments in flight: slot s uses fr[s % kAhead]

    // A fragment of slot S (compile-time)
    template <int S> SF_DEVICE sf_v8s load_frag() const {
        if constexpr (S < 2 * NA) {
            constexpr int qb = S / NA, i = S % NA, ks = i / 2;
            return frag_rows<HD>((i & 1) ? lds_do : lds_q, qb * 32, ks, fo);
        } else {
            constexpr int g = S - 2 * NA, qb = g / NG, i = g % NG, jp = i / (2 * DB), d = (i / 2) % DB;
            return frag_tr<HD>((i & 1) ? lds_q : lds_do, d, qb * 32 + 16 * jp, fo);   // dV: dO^T | dK: Q^T
        }
    }
    template <int E> SF_DEVICE void element(int qb) {   // element E = 4 j + t of block qb
        constexpr int j = E / 4, t = E % 4;
        float x = fmaf(s[qb][E], sc, -kLog2e * l4[qb][j][t]);
        if (MASK) {   // a select on the EXPONENT (exp2(-inf) == 0), as a value expression: a conditional assignment compiles to an exec-masked
            // branch per element, and a 0 / 1 factor on the probability turns a masked score far above the row's lse (exp2 -> inf) into NaN
            const int C = qb * 32 + 8 * j + t;
            x = (C >= lo && C < up) ? x : -INFINITY;
        }
        const float pv = sf_exp2_raw(x);
        s[qb][E] = pv;
        dp[qb][E] = pv * (dp[qb][E] - d4[qb][j][t]);
        if constexpr (E % 8 == 7) {
            pf[qb][E / 8] = pack_bf16x8(s[qb], E - 7);
            df[qb][E / 8] = pack_bf16x8(dp[qb], E - 7);
        }
    }
    template <class Bank, class Dma>
    SF_DEVICE void run(Bank& bank, Dma&& dma_piece) {
        // burst: the first kAhead fragments
        static_for<0, kAhead>([&](auto I) SF_LAMBDA_INLINE { fr[decltype(I)::value] = load_frag<decltype(I)::value>(); });
        SF_SCHED_FENCE();
        static_for<0, NSLOT>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int S = decltype(I)::value;
            const sf_v8s a = fr[S % kAhead];
            if constexpr (S < 2 * NA) {
                constexpr int qb = S / NA, i = S % NA, ks = i / 2;
                if constexpr (i & 1) bank.template mfma_dp<ks, ks == 0>(dp[qb], a);
                else bank.template mfma_s<ks, ks == 0>(s[qb], a);
            } else {
                constexpr int g = S - 2 * NA, qb = g / NG, i = g % NG, jp = i / (2 * DB), d = (i / 2) % DB;
                if constexpr (i & 1) bank.template mfma_dk<d>(a, df[qb][jp]);
                else bank.template mfma_dv<d>(a, pf[qb][jp]);
            }
            // ---- fillers of this slot
            if constexpr (S + kAhead < NSLOT) fr[S % kAhead] = load_frag<S + kAhead>();
            dma_piece(std::integral_constant<int, S>{});
            if constexpr (S >= NA / 2 && S < NA / 2 + 8) {          // lse / delta of block 0: 4 + 4 reads
                constexpr int k = S - NA / 2, j = k % 4;
                const int ql0 = 8 * j + 4 * hi;
                if constexpr (k < 4) l4[0][j] = *reinterpret_cast<const sf_v4f*>(lds_lse + ql0);
                else d4[0][j] = *reinterpret_cast<const sf_v4f*>(lds_dlt + ql0);
            }
            if constexpr (S >= NA + NA / 2 && S < NA + NA / 2 + 8) { // ... of block 1
                constexpr int k = S - NA - NA / 2, j = k % 4;
                const int ql0 = 32 + 8 * j + 4 * hi;
                if constexpr (k < 4) l4[1][j] = *reinterpret_cast<const sf_v4f*>(lds_lse + ql0);
                else d4[1][j] = *reinterpret_cast<const sf_v4f*>(lds_dlt + ql0);
            }
            if constexpr (S >= P0 && S < P0 + PSPAN) {               // P(0): 16 elements over PSPAN slots
                constexpr int k = S - P0, e0 = k * 16 / PSPAN, e1 = (k + 1) * 16 / PSPAN;
                static_for<e0, e1>([&](auto E) SF_LAMBDA_INLINE { element<decltype(E)::value>(0); });
            }
            if constexpr (S >= P1 && S < P1 + PSPAN) {               // P(1)
                constexpr int k = S - P1, e0 = k * 16 / PSPAN, e1 = (k + 1) * 16 / PSPAN;
                static_for<e0, e1>([&](auto E) SF_LAMBDA_INLINE { element<decltype(E)::value>(1); });
            }
            SF_SCHED_FENCE();
        });
    }
};

template <int HD>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) attn_bwd_dkv_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, NW = 4, KB = NW * 32, TILE = 128 * HD * 2 + 768, NBUF = 3;
    constexpr int NDMA = 2 * TileStage<HD, 64, NW>::NI + 1;   // DMA pieces per wave and tile
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    // 1-D grid, heaviest first: key block 0 sees every query tile, the last one only the final tiles
    const int per_kb = p.nkv * p.B;
    const int hsplit = p.hsplit > 1 ? p.hsplit : 1;
    int kbi, g, b;
    if (p.l2_map) {   // pair-major (tools A/B only: heaviest-first over ALL pairs measured 20 % faster for this kernel)
        const int nkb = (p.S + KB - 1) / KB;
        const int v = attn_work_index((int)blockIdx.x, nkb * per_kb, 1);
        if (v >= nkb * per_kb) return;
        const int pr = v / nkb;
        kbi = v - pr * nkb; b = pr / p.nkv; g = pr - b * p.nkv;
    } else {
        const int bid = (int)blockIdx.x / hsplit, gb = bid % per_kb;
        kbi = bid / per_kb; g = gb % p.nkv; b = gb / p.nkv;
    }
    const int hs = (int)blockIdx.x % hsplit;     // this workgroup's slice of the group's query heads (heaviest key block still first)
    const int kb0 = kbi * KB;
    const int S = p.S, nrep = (p.nh / p.nkv) / hsplit, hh0 = hs * nrep;
    const int kvlen = p.kv_len ? p.kv_len[b] : S;
    if (kb0 >= kvlen) return;   // keys at / after kv_len never receive probability mass (workgroup-uniform; the reduce skips them too)
    const int kw0 = kb0 + wave * 32;
    const int ki = kw0 + c;  // this lane's key (column of S)
    const bool kok = ki < S;
    const long krow = (long)b * S + (kok ? ki : S - 1);
    FragOff<HD> fo;
    fo.init(lane);

    DkvBank<HD> bank;
    bank.init();
    {
        sf_v8s kt[KS], vt[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kt[ks] = *reinterpret_cast<const sf_v8s*>(p.k0 + krow * p.ldk + g * HD + 16 * ks + 8 * hi);
            vt[ks] = *reinterpret_cast<const sf_v8s*>(p.v0 + krow * p.ldv + g * HD + 16 * ks + 8 * hi);
        }
        static_for<0, KS>([&](auto I) SF_LAMBDA_INLINE {
            constexpr int i = decltype(I)::value;
            bank.template set_k<i>(kt[i]);   // (the asm reads the value: the compiler waits for the load HERE, ahead of the loop)
            bank.template set_v<i>(vt[i]);
        });
    }

    const int qt_first = kb0 / 64;
    const int nqt = (S + 63) / 64;
    const int per_head = nqt - qt_first, n_it = nrep * per_head;
    // a query q of the tile is visible to this lane's key iff  ki <= q < S  (and the key itself is valid): with
    // q = q0 + 4*hi + C (C a compile-time constant per register) that is  lo <= C < up  for two per-tile values
    const int key_lo = ki < kvlen ? ki : 0x3fffffff;
    TileStage<HD, 64, NW> stq, stdo;
    stq.init(p.ldq, wave, lane);
    stdo.init(p.lddo, wave, lane);
    const unsigned qtile = (unsigned)(64 * p.ldq * 2), dotile = (unsigned)(64 * p.lddo * 2);
    const int h0 = g * (p.nh / p.nkv) + hh0;     // first query head of this workgroup
    const sf_bf16* qb_base = p.q + (long)b * S * p.ldq + (long)h0 * HD;
    const sf_bf16* dob_base = p.dout + (long)b * S * p.lddo + (long)h0 * HD;
    const float* lse_base = p.lse + ((long)b * p.nh + h0) * S;
    const float* dlt_base = p.delta + ((long)b * p.nh + h0) * S;
    // Every wave issues the same number of DMA pieces per tile (a counted vmcnt needs one immediate): its Q and dO pieces
    // plus ONE 4-byte-per-lane piece -- lse (wave 0), delta (wave 1), or a re-read of lse into the slot's padding.
    struct Src { SfBufB q, dout, aux; unsigned qoff, dooff, auxoff; char* dst; };
    // Tile `it` of the head-major walk = (head hh, 64-query tile qt).  Two walkers -- the tile being computed and the one being staged two
    // ahead -- advance by one tile per iteration with a compare and an add (round 5: `it / per_head` for each of them was ~100 scalar /
    // vector instructions between two tile bodies, a fifth of a body's matrix-pipe time with nothing issued to the pipe)
    struct Walk { int hh, qt, slot; };
    auto advance = [&](Walk& w) SF_LAMBDA_INLINE {
        if (++w.qt == nqt) { w.qt = qt_first; ++w.hh; }
        if (++w.slot == NBUF) w.slot = 0;
    };
    auto source = [&](const Walk& w) SF_LAMBDA_INLINE {
        Src r;
        r.q = rows_buf<HD>(qb_base + w.hh * HD, p.ldq, S);
        r.dout = rows_buf<HD>(dob_base + w.hh * HD, p.lddo, S);
        r.aux = sf_make_bufb((wave == 1 ? dlt_base : lse_base) + (long)w.hh * S, (unsigned)S * 4u);
        r.qoff = (unsigned)w.qt * qtile; r.dooff = (unsigned)w.qt * dotile; r.auxoff = (unsigned)(w.qt * 64 + lane) * 4u;
        r.dst = smem + w.slot * TILE;
        return r;
    };
    auto piece = [&](const Src& r, int k) SF_LAMBDA_INLINE {   // piece k of NDMA (compile-time k at every call site)
        constexpr int NI = TileStage<HD, 64, NW>::NI;
        if (k < NI) sf_bufb_glds16(r.q, stq.off[k] + r.qoff, r.dst + (stq.piece0 + k) * 1024);
        else if (k < 2 * NI) sf_bufb_glds16(r.dout, stdo.off[k - NI] + r.dooff, r.dst + 64 * HD * 2 + (stdo.piece0 + k - NI) * 1024);
        else sf_bufb_glds4(r.aux, r.auxoff, r.dst + 128 * HD * 2 + (wave < 2 ? wave : 2) * 256);
    };
    // stage the walker's tile -- or, past the last tile, the same pieces against empty descriptors (zeros into a ring slot nobody reads any
    // more): no branch inside the slot stream, and the counted wait at the top of the next iteration stays exact
    auto src_or_empty = [&](const Walk& w, bool real) SF_LAMBDA_INLINE {
        Src r = source(w);
        if (!real) { sf_bufb_empty(r.q); sf_bufb_empty(r.dout); sf_bufb_empty(r.aux); }
        return r;
    };
    Walk wc{0, qt_first, 0}, ws{0, qt_first, 0};        // the tile computed in this iteration; the tile staged next (two ahead in the loop)
    if (n_it > 0) {
        const Src r0 = src_or_empty(ws, true);
        static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(r0, decltype(K)::value); });
        advance(ws);
        const Src r1 = src_or_empty(ws, n_it > 1);      // (keeps the piece count of the counted wait: an empty stand-in for tile 1)
        static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(r1, decltype(K)::value); });
        advance(ws);
    }
    for (int it = 0; it < n_it; ++it) {
        const int q0 = wc.qt * 64;
        sf_wait_vmcnt<NDMA>();          // tile `it` landed; the NDMA pieces of tile it+1 (or its empty stand-ins) may be in flight
        sf_syncthreads();               // ... for every wave; ring slot (it+2) % 3 == (it-1) % 3 is no longer being read
        const bool more = it + 2 < n_it;
        const char* lds_q = smem + wc.slot * TILE;
        const Src nxt = src_or_empty(ws, more);          // the pieces of tile it+2 are fillers of this tile's first slots
        advance(wc);
        advance(ws);
        if (q0 + 63 < kw0) {            // every query of the tile is before this wave's keys: only the staging duty remains
            static_for<0, NDMA>([&](auto K) SF_LAMBDA_INLINE { piece(nxt, decltype(K)::value); });
            continue;
        }
        auto dma = [&](auto Sl) SF_LAMBDA_INLINE {       // filler: DMA piece (slot - 1) of tile it+2
            constexpr int k = decltype(Sl)::value - 1;
            if constexpr (k >= 0 && k < NDMA) piece(nxt, k);
        };
        const bool need_mask = (q0 < kw0 + 32) || (kw0 + 31 >= kvlen) || (q0 + 63 >= S);  // wave-uniform
        const float sc = p.scale * kLog2e;
        const int lo = key_lo - q0 - 4 * hi, up = S - q0 - 4 * hi;
        const float* ll = reinterpret_cast<const float*>(lds_q + 128 * HD * 2);
        if (need_mask) {
            DkvTile<HD, true> t{lds_q, lds_q + 64 * HD * 2, ll, ll + 64, fo, hi, sc, lo, up};
            t.run(bank, dma);
        } else {
            DkvTile<HD, false> t{lds_q, lds_q + 64 * HD * 2, ll, ll + 64, fo, hi, sc, lo, up};
            t.run(bank, dma);
        }
    }
    sf_wait_vm0();   // (the stand-in pieces of the last iterations are still in flight: LDS must not be released under them)
    bank.drain();
    if (!kok) return;
    const bool split = p.hsplit > 1;               // partial sums of this head slice: written, not accumulated
    const long orow = split ? (long)hs * p.part_stride + krow * ((long)p.nkv * HD) : krow * p.lddk;
    float* dkrow = (split ? p.part_k : p.dk) + orow + g * HD;
    float* dvrow = (split ? p.part_v : p.dv) + orow + g * HD;
    static_for<0, DB>([&](auto D) SF_LAMBDA_INLINE {
        constexpr int d = decltype(D)::value;
        const sf_v16f ak = bank.template get_dk<d>(), av = bank.template get_dv<d>();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = d * 32 + 8 * j + 4 * hi;
            sf_v4f a = sf_v4f{0.f, 0.f, 0.f, 0.f}, e = a;
            if (!split) {
                a = *reinterpret_cast<const sf_v4f*>(dkrow + col);
                e = *reinterpret_cast<const sf_v4f*>(dvrow + col);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t] += ak[4 * j + t] * p.scale; e[t] += av[4 * j + t]; }
            *reinterpret_cast<sf_v4f*>(dkrow + col) = a;
            *reinterpret_cast<sf_v4f*>(dvrow + col) = e;
        }
    });
}

// dk / dv += sum over the head slices' partials (fixed order: deterministic).  Key blocks at / after kv_len were never written.
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2) attn_dkv_reduce_kernel(AttnBwdArgs p, int W, int KB) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;       // one float4 of the [B*S, W] gradients
    const long n4 = (long)p.B * p.S * (W / 4);
    if (i4 >= n4) return;
    const long row = i4 / (W / 4);
    const int col = (int)(i4 - row * (W / 4)) * 4;
    const int b = (int)(row / p.S), t = (int)(row - (long)b * p.S);
    const int kvlen = p.kv_len ? p.kv_len[b] : p.S;
    if ((t / KB) * KB >= kvlen) return;
    sf_v4f a = *reinterpret_cast<const sf_v4f*>(p.dk + row * p.lddk + col);
    sf_v4f e = *reinterpret_cast<const sf_v4f*>(p.dv + row * p.lddk + col);
    for (int hs = 0; hs < p.hsplit; ++hs) {
        const sf_v4f x = *reinterpret_cast<const sf_v4f*>(p.part_k + hs * p.part_stride + row * W + col);
        const sf_v4f y = *reinterpret_cast<const sf_v4f*>(p.part_v + hs * p.part_stride + row * W + col);
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += x[k]; e[k] += y[k]; }
    }
    *reinterpret_cast<sf_v4f*>(p.dk + row * p.lddk + col) = a;
    *reinterpret_cast<sf_v4f*>(p.dv + row * p.lddk + col) = e;
}

// ------------------------------------------------ head_dim 256: the role-split kernel
// At head_dim 256 both gradients of 32 keys are 256 accumulator registers -- the whole AGPR file, with the K / V fragments
// (another 128) nowhere to go -- so the one-wave-owns-both-gradients kernel above does not exist for it.  This is round 2's
// kernel (compiler-scheduled builtin MFMAs, generic in HD), kept for that case: the two gradients of a key sub-block go to
// two waves, each recomputing S; at HD 256 a wave carries 128 accumulator + 128 fragment registers, so the workgroup is 4 waves
// (2 key sub-blocks x 2 roles = 64 keys), one wave per SIMD.  Shipped recipes with head_dim 256 (gemma3-1b, qwen3-next-80b-a3b,
// qwen3.5-35b-a3b: llama3_eagle.py:547-550 takes any head_dim) have 4 / 16 query heads: attention is a few percent of their step.
// Workgroup = NW waves = NW/2 key sub-blocks of 32 keys x 2 roles: waves [0, NW/2) accumulate dV^T,
// waves [NW/2, NW) accumulate dK^T of the same keys (each recomputes S; a wave then carries ONE
// 64-register accumulator set, so the kernel fits 2 waves/SIMD without spilling and the two roles
// of a key sub-block sit on the same SIMD and overlap exp/LDS work with MFMA).
// LDS (double buffered): Q [64][HD], dO [64][HD], lse2[64], delta[64]; Q^T / dO^T fragments come from the
// same tiles through the hardware transpose read
template <int HD, int NW>
SF_GLOBAL void SF_LAUNCH_BOUNDS(NW * 64, NW / 4) attn_bwd_dkv_rs_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, NSUB = NW / 2, KB = NSUB * 32, TILE = 128 * HD * 2 + 512;
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    const int role = wave / NSUB;  // 0: dV, 1: dK   (wave-uniform)
    const int sub = wave - role * NSUB;
    // 1-D grid, heaviest first: key block 0 sees every query tile, the last one only the final tiles
    const int per_kb = p.nkv * p.B;
    const int hsplit = p.hsplit > 1 ? p.hsplit : 1;
    int kbi, g, b;
    if (p.l2_map) {   // pair-major: the key blocks of one (batch, kv head) stream the same Q / dO tiles
        const int nkb = (p.S + KB - 1) / KB;
        const int v = attn_work_index((int)blockIdx.x, nkb * per_kb, 1);
        if (v >= nkb * per_kb) return;
        const int pr = v / nkb;
        kbi = v - pr * nkb; b = pr / p.nkv; g = pr - b * p.nkv;
    } else {
        const int bid = (int)blockIdx.x / hsplit, gb = bid % per_kb;
        kbi = bid / per_kb; g = gb % p.nkv; b = gb / p.nkv;
    }
    const int hs = (int)blockIdx.x % hsplit;     // head split (small B * nkv), as in attn_bwd_dkv_kernel
    const int kb0 = kbi * KB;
    const int S = p.S, nrep = (p.nh / p.nkv) / hsplit, h_first = g * (p.nh / p.nkv) + hs * nrep;
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
            const int h = h_first + hh;
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
    const bool split = p.hsplit > 1;      // partial sums of this head slice: written, not accumulated (attn_dkv_reduce_kernel adds them)
    float* orow = split ? (role == 0 ? p.part_v : p.part_k) + (long)hs * p.part_stride + krow * ((long)p.nkv * HD) + g * HD
                        : (role == 0 ? p.dv : p.dk) + krow * p.lddk + g * HD;
    const float oscale = role == 0 ? 1.0f : p.scale;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = d * 32 + 8 * j + 4 * hi;
            sf_v4f a = sf_v4f{0.f, 0.f, 0.f, 0.f};
            if (!split) a = *reinterpret_cast<const sf_v4f*>(orow + col);
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] += acc[d][4 * j + t] * oscale;
            *reinterpret_cast<sf_v4f*>(orow + col) = a;
        }
}

}  // namespace

// Head split of the dK/dV kernel: its parallelism is (batch, kv head, 128-key block) and the longest workgroup (key block 0) walks
// every query tile of every query head of its group -- with B * nkv * S / 128 < 2 x 256 the launch is as long as that one
// workgroup (measured at cfg 4's recipe, bs 1 x 4096, 32 / 4 heads: 0.15 of the MFMA peak where bs 4 reaches 0.43).  The heads
// of a group are then divided over `hsplit` workgroups (the smallest divisor of nh / nkv that brings the grid to 512).
static int dkv_head_split(int B, int S, int nh, int nkv, int hd) {
    const long wgs = (long)((S + 127) / 128) * nkv * B * (hd == 256 ? 2 : 1);   // (head_dim 256: 64 keys per workgroup)
    const int nrep = nh / nkv;
    if (wgs >= 512) return 1;
    for (int d = 2; d < nrep; ++d)
        if (nrep % d == 0 && wgs * d >= 512) return d;
    return nrep;
}
extern "C" long sf_attn_bwd_dkv_workspace_floats(int B, int S, int nh, int nkv, int hd) {
    if (B <= 0 || S <= 0 || nh <= 0 || nkv <= 0 || nh % nkv) return 0;
    const int hs = dkv_head_split(B, S, nh, nkv, hd);
    return hs > 1 ? 2L * hs * B * S * nkv * hd : 0;
}

extern "C" int sf_attn_bwd_dkv(const void* q, long ldq, const void* dout, long lddo,
                               const void* k0, long ldk, const void* v0, long ldv, const int* kv_len, const float* lse,
                               const float* delta, float* dk, float* dv, long lddk, int B, int S, int nh, int nkv,
                               int hd, float scale, float* workspace, long workspace_floats, void* stream) {
    SF_CHECK_ARG(B > 0 && S > 0 && nh > 0 && nkv > 0 && nh % nkv == 0, "sf_attn_bwd_dkv: bad shape");
    SF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddk % 4 == 0,
                 "sf_attn_bwd_dkv: row strides must be multiples of 8 (16-byte segments)");
    SF_CHECK_ARG((long)S * ldq * 2 < (1L << 31) && (long)S * lddo * 2 < (1L << 31),
                 "sf_attn_bwd_dkv: S * ld exceeds the 2 GiB range of a buffer descriptor");
    AttnBwdArgs p;
    memset(&p, 0, sizeof(p));
    p.q = (const sf_bf16*)q; p.ldq = ldq;
    p.dout = (const sf_bf16*)dout; p.lddo = lddo;
    p.k0 = (const sf_bf16*)k0; p.ldk = ldk;
    p.v0 = (const sf_bf16*)v0; p.ldv = ldv;
    p.kv_len = kv_len; p.lse = lse; p.delta = delta;
    p.dk = dk; p.dv = dv; p.lddk = lddk;
    p.B = B; p.S = S; p.nh = nh; p.nkv = nkv; p.scale = scale;
    p.l2_map = sf_knob("SF_ATTN_DKV_L2MAP", 0);   // heaviest-first over ALL pairs wins here (measured: pair-major +20 %)
    SF_CHECK_ARG(hd == 64 || hd == 128 || hd == 256, "head_dim must be 64, 128 or 256");
    int hsplit = dkv_head_split(B, S, nh, nkv, hd);
    if (!workspace || workspace_floats < 2L * hsplit * B * S * nkv * hd || ((size_t)workspace & 15) || p.l2_map) hsplit = 1;
    p.hsplit = hsplit;
    p.part_stride = (long)B * S * nkv * hd;
    p.part_k = workspace;
    p.part_v = workspace ? workspace + hsplit * p.part_stride : nullptr;
    if (hd == 256) {
        constexpr int HD = 256, NW = 4;       // 2 key sub-blocks x 2 roles: 64 keys per workgroup
        dim3 grid(attn_grid((long)((S + NW * 16 - 1) / (NW * 16)) * nkv * B * hsplit, p.l2_map));
        if (sf_knob("SF_ATTN_W1", 1)) {     // the wave-pair kernel of sf_attn_w1_dkv.hip (same grid, same partial-sum layout)
            const int st = attn_bwd_dkv_w1_launch(p, hd, stream);
            if (st) return st;
        } else {                            // (tools build A/B: round 2's role-split kernel)
            SF_ALLOW_SMEM((attn_bwd_dkv_rs_kernel<HD, NW>), 2 * (128 * HD * 2 + 512));
            SF_LAUNCH((attn_bwd_dkv_rs_kernel<HD, NW>), grid, dim3(NW * 64), 2 * (128 * HD * 2 + 512), stream, p);
        }
        if (hsplit > 1) {
            const int W = nkv * hd;
            const long n4 = (long)B * S * (W / 4);
            SF_LAUNCH(attn_dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, p, W, NW * 16);
        }
        return sf_check_launch("sf_attn_bwd_dkv");
    }
    dim3 grid(attn_grid((long)((S + 127) / 128) * nkv * B * hsplit, p.l2_map));   // 128 keys per workgroup
    if (hd == 128) {
        constexpr int HD = 128;
        SF_ALLOW_SMEM((attn_bwd_dkv_kernel<HD>), 3 * (128 * HD * 2 + 768));
        SF_LAUNCH((attn_bwd_dkv_kernel<HD>), grid, dim3(256), 3 * (128 * HD * 2 + 768), stream, p);
    } else {
        constexpr int HD = 64;
        SF_ALLOW_SMEM((attn_bwd_dkv_kernel<HD>), 3 * (128 * HD * 2 + 768));
        SF_LAUNCH((attn_bwd_dkv_kernel<HD>), grid, dim3(256), 3 * (128 * HD * 2 + 768), stream, p);
    }
    if (hsplit > 1) {
        const int W = nkv * hd;
        const long n4 = (long)B * S * (W / 4);
        SF_LAUNCH(attn_dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, p, W, 128);
    }
    return sf_check_launch("sf_attn_bwd_dkv");
}
