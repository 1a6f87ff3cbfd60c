// TTT attention backward at head_dim 256: dK and dV of the step-0 keys / values as a PAIR of waves per 32 keys (semantics: sf_attn.hip;
// reference blueprint: _FlashCachedMergeFunc.backward, specforge/modeling/draft/llama3_eagle.py:1080-1151).
//
// Both gradients of 32 keys are 256 accumulator registers at this head width and the K and V fragments another 128: the one-wave-
// owns-both-gradients kernel of sf_attn_dkv.hip does not exist here.  Round 4 shipped the round-2 structure instead (two waves per 32
// keys, one per gradient, EACH recomputing S = Q.K^T; the dK wave also dP = dO.V^T: 160 MFMAs per 64 queries where 128 are useful, the
// lighter wave idle a third of the time, compiler-scheduled: 0.18 of the MFMA peak).  Here the pair splits the SCORE products instead:
//
//     wave A (role 0): S = Q.K^T  ->  P = exp2(S sc - lse)  ->  dV^T += dO^T.P          owns dV, holds the K fragments
//     wave B (role 1): dP = dO.V^T ->  dS = P (dP - delta)   ->  dK^T += Q^T.dS          owns dK, holds the V fragments
//
// P goes from A to B through LDS (fp32, the lanes' own 64 bytes: both tiles have the same C layout), one tile later: the workgroup's
// per-tile barrier is the only synchronisation.  Every product is computed once, both waves issue 32 MFMAs per 32-query tile, and each
// wave's bank is 128 accumulator + 64 fragment AGPRs (AgprBank), which leaves the compiler its 256 VGPRs.
//
// Pipeline (tile t = 32 queries; body n runs between two workgroup barriers):
//     A:  S(n+1) | exp(n) -> P(n) to LDS and packed | dV(n)            B:  dP(n) | dS(n-1) from P(n-1) and dP(n-1), packed | dK(n-1)
// Q(t) is read in bodies t-1 (rows, A) and t+1 (transposed, B), dO(t) in body t only (rows by B, transposed by A), so the two operands
// have rings of their own: Q five slots {n-1, n, n+1 live; n+2, n+3 in flight}, dO three {n live; n+1, n+2 in flight} -- 16.25 KiB each
// with lse (dO slot) / delta (Q slot) in the tail: 130 KiB, two tiles of DMA lead behind a COUNTED vmcnt, plus the two-deep exchange
// buffer (16 KiB).  The streams are slot plans as in sf_attn_dkv.hip / sf_attn_w1.hip: one MFMA per slot, its fillers, a fence.
#include "sf_attn_common.h"

using namespace sfattn;

namespace {

template <int HD>
struct PairBank : AgprBank<HD / 32, HD / 16> {       // a[0 : 16 DB) the wave's gradient^T, then the K (A) or V (B) fragments
    static constexpr int KS = HD / 16, DB = HD / 32;
    using Base = AgprBank<DB, KS>;
    template <int I> SF_DEVICE void set_f(sf_v8s v) { Base::template set_b<I>(v); }
    template <int I, bool FIRST> SF_DEVICE void mfma_score(sf_v16f& c, sf_v8s a) { Base::template mfma_vb<I, FIRST>(c, a); }
    template <int D> SF_DEVICE void mfma_grad(sf_v8s a, sf_v8s b) { Base::template mfma_acc<D>(a, b); }
    template <int D> SF_DEVICE sf_v16f get_grad() { return Base::template get<D>(); }
};

constexpr int kAhead = SF_ATTN_KAHEAD;

// The first kAhead transposed fragments of a gradient phase (G-slot M: d = M % DB, rows 16 (M / DB) ..)
template <int HD, int M>
SF_DEVICE sf_v8s grad_frag(const char* lds_x, const FragOff<HD>& fo) {
    constexpr int DB = HD / 32;
    return frag_tr<HD>(lds_x, M % DB, 16 * (M / DB), fo);
}

// score phase: c = rows(lds_rows) . bank fragments (KS MFMAs); `fill(slot)` supplies the fillers.  MFMA = false: the fillers alone (the
// pipeline's last bodies: nothing left to score, the previous tile's arithmetic still due).  PRE: the gradient phase follows -- its first
// kAhead fragments (of lds_x) are read in this phase's last kAhead slots, so it starts without a read burst of its own (a body is only
// 32 MFMAs: each exposed burst is ~10 % of it)
template <int HD, bool MFMA, bool PRE, class Bank, class Fill>
SF_DEVICE void score_phase(Bank& bank, const char* lds_rows, const char* lds_x, const FragOff<HD>& fo, sf_v16f& c, sf_v8s (&gf)[kAhead],
                           Fill&& fill) {
    constexpr int KS = HD / 16;
    static_assert(KS >= kAhead, "plan: head_dim >= 128");
    sf_v8s rf[kAhead];
    if constexpr (MFMA) {
        static_for<0, kAhead>([&](auto I) SF_LAMBDA_INLINE { rf[decltype(I)::value] = frag_rows<HD>(lds_rows, 0, decltype(I)::value, fo); });
        SF_SCHED_FENCE();
    }
    static_for<0, KS>([&](auto I) SF_LAMBDA_INLINE {
        constexpr int S = decltype(I)::value;
        if constexpr (MFMA) {
            bank.template mfma_score<S, S == 0>(c, rf[S % kAhead]);
            if constexpr (S + kAhead < KS) rf[S % kAhead] = frag_rows<HD>(lds_rows, 0, S + kAhead, fo);
        }
        if constexpr (PRE && S >= KS - kAhead) gf[S - (KS - kAhead)] = grad_frag<HD, S - (KS - kAhead)>(lds_x, fo);
        fill(std::integral_constant<int, S>{});
        SF_SCHED_FENCE();
    });
}

// gradient phase: acc[d] += tr(lds_x)(d, 16 jp) . f[jp]   (2 DB MFMAs); PRE: gf already holds the first kAhead fragments
template <int HD, bool PRE, class Bank, class Fill>
SF_DEVICE void grad_phase(Bank& bank, const char* lds_x, const FragOff<HD>& fo, const sf_v8s (&f)[2], sf_v8s (&gf)[kAhead], Fill&& fill) {
    constexpr int DB = HD / 32, NG = 2 * DB;
    static_assert(NG >= kAhead, "plan: head_dim >= 128");
    if constexpr (!PRE) {
        static_for<0, kAhead>([&](auto I) SF_LAMBDA_INLINE { gf[decltype(I)::value] = grad_frag<HD, decltype(I)::value>(lds_x, fo); });
        SF_SCHED_FENCE();
    }
    static_for<0, NG>([&](auto I) SF_LAMBDA_INLINE {
        constexpr int M = decltype(I)::value, jp = M / DB, d = M % DB;
        bank.template mfma_grad<d>(gf[M % kAhead], f[jp]);
        if constexpr (M + kAhead < NG) gf[M % kAhead] = grad_frag<HD, M + kAhead>(lds_x, fo);
        fill(std::integral_constant<int, M>{});
        SF_SCHED_FENCE();
    });
}

// one body of a role: [score (+ the previous tile's arithmetic as fillers)] [gradient (+ the DMA pieces as fillers)], either part optional
template <int HD, class Bank, class FillS, class FillG, class Pieces>
SF_DEVICE void pair_body(Bank& bank, bool do_s, bool do_e, const char* lds_rows, const char* lds_x, const FragOff<HD>& fo, sf_v16f& c,
                         const sf_v8s (&f)[2], FillS&& fill_s, FillG&& fill_g, Pieces&& all_pieces) {
    sf_v8s gf[kAhead];
    if (do_s && do_e) {                 // the steady state
        score_phase<HD, true, true>(bank, lds_rows, lds_x, fo, c, gf, fill_s);
        grad_phase<HD, true>(bank, lds_x, fo, f, gf, fill_g);
    } else if (do_e) {                  // the pipeline's tail: nothing left to score
        score_phase<HD, false, true>(bank, lds_rows, lds_x, fo, c, gf, fill_s);
        grad_phase<HD, true>(bank, lds_x, fo, f, gf, fill_g);
    } else {
        if (do_s) {                     // the pipeline's head (or a tile before this pair's keys behind it)
            score_phase<HD, true, false>(bank, lds_rows, lds_x, fo, c, gf, fill_s);
            // the scores are copied right behind this (no gradient phase in between): an MFMA result in VGPRs is read by VALU only after
            // the matrix pipe has delivered it -- the asm MFMAs are invisible to the compiler's hazard recogniser (on the steady path the
            // distance is 16+ MFMA slots)
            sf_mfma_drain();
        }
        all_pieces();
    }
}

template <int HD>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 1) attn_bwd_dkv_pair_kernel(AttnBwdArgs p) {
    constexpr int KS = HD / 16, DB = HD / 32, NW = 4, KB = 64, QT = 32, NQ = 5, ND = 3;
    constexpr int ROWS = QT * HD * 2;                       // bytes of a 32-row Q (or dO) tile
    constexpr int SLOT = ROWS + 256;                        // rows | 64 floats: delta (Q ring) / lse (dO ring) of the tile's queries
    constexpr int XCH = 32 * 32 * 4;                        // one pair's fp32 P tile
    constexpr int NI = TileStage<HD, QT, NW>::NI;           // Q (and dO) pieces per wave and tile
    constexpr int NGRP = 2 * NI + 1;                        // DMA pieces per wave and body: Q + dO + one 4-byte-per-lane piece
    SF_DYN_SMEM(smem);      // NQ x SLOT (Q ring) | ND x SLOT (dO ring) | 2 (parity) x 2 (pairs) x XCH
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id(), c = lane & 31, hi = lane >> 5;
    const int role = wave & 1, sub = wave >> 1;             // pair `sub` = waves 2 sub (A: S, dV) and 2 sub + 1 (B: dP, dK)
    // 1-D grid, heaviest first: key block 0 sees every query tile, the last one only the final tiles
    const int per_kb = p.nkv * p.B;
    const int hsplit = p.hsplit > 1 ? p.hsplit : 1;
    const int bid = (int)blockIdx.x / hsplit, gb = bid % per_kb;
    const int kbi = bid / per_kb, g = gb % p.nkv, b = gb / p.nkv;
    const int hs = (int)blockIdx.x % hsplit;                // head split (small B * nkv), as in attn_bwd_dkv_kernel
    const int kb0 = kbi * KB;
    const int S = p.S, nrep = (p.nh / p.nkv) / hsplit, h_first = g * (p.nh / p.nkv) + hs * nrep;
    const int kvlen = p.kv_len ? p.kv_len[b] : S;
    if (kb0 >= kvlen) return;   // keys at / after kv_len never receive probability mass (workgroup-uniform; the reduce skips them too)
    const int kw0 = kb0 + sub * 32;
    const int ki = kw0 + c;     // this lane's key (column of S)
    const bool kok = ki < S;
    const long krow = (long)b * S + (kok ? ki : S - 1);
    const float sc = p.scale * kLog2e;
    FragOff<HD> fo;
    fo.init(lane);

    PairBank<HD> bank;
    bank.init();
    {
        const sf_bf16* src = role == 0 ? p.k0 + krow * p.ldk : p.v0 + krow * p.ldv;
        sf_v8s ft[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ft[ks] = *reinterpret_cast<const sf_v8s*>(src + g * HD + 16 * ks + 8 * hi);
        // (the asm reads the value: the compiler waits for the loads HERE, ahead of the loop -- see TileStage)
        static_for<0, KS>([&](auto I) SF_LAMBDA_INLINE { bank.template set_f<decltype(I)::value>(ft[decltype(I)::value]); });
    }

    const int qt_first = kb0 / QT, nqt = (S + QT - 1) / QT;
    const int per_head = nqt - qt_first, n_it = nrep * per_head;         // tiles, head-major (workgroup-uniform)
    // a query q of the tile is visible to this lane's key iff  ki <= q < S  (and the key itself is valid): with
    // q = q0 + 4 * hi + C (C a compile-time constant per register) that is  lo <= C < up  for two per-tile values
    const int key_lo = ki < kvlen ? ki : 0x3fffffff;
    TileStage<HD, QT, NW> stq, stdo;
    stq.init(p.ldq, wave, lane);
    stdo.init(p.lddo, wave, lane);
    const unsigned qtile = (unsigned)(QT * p.ldq * 2), dotile = (unsigned)(QT * p.lddo * 2);
    const sf_bf16* qb_base = p.q + (long)b * S * p.ldq + (long)h_first * HD;
    const sf_bf16* dob_base = p.dout + (long)b * S * p.lddo + (long)h_first * HD;
    const float* lse_base = p.lse + ((long)b * p.nh + h_first) * S;
    const float* dlt_base = p.delta + ((long)b * p.nh + h_first) * S;
    char* const q_ring = smem;
    char* const do_ring = smem + NQ * SLOT;
    // Tile t of the head-major walk = (head hh, 32-query tile qt).  Three walkers run ahead of each other -- the Q stream (tile n + 3 in
    // body n), the dO stream (n + 2), the arithmetic (n + 1) -- each advanced by one tile per body with a compare and an add: a division
    // per coordinate and body was ~150 scalar / vector instructions between the last MFMA of a body and the first of the next.
    struct Walk {
        int t, hh, qt, slot;            // tile index (may start below 0), its coordinates (those of tile 0 while t <= 0), its ring slot
    };
    auto advance = [&](Walk& w, int nslots) SF_LAMBDA_INLINE {
        ++w.t;
        if (w.t > 0 && ++w.qt == nqt) { w.qt = qt_first; ++w.hh; }
        if (++w.slot == nslots) w.slot = 0;
    };
    auto real = [&](const Walk& w) SF_LAMBDA_INLINE { return w.t >= 0 && w.t < n_it; };
    // DMA group m (issued in body m; before the loop for m = -3, -2): Q(m+3) + delta(m+3) into Q slot (m+3) % NQ, dO(m+2) + lse(m+2) into dO
    // slot (m+2) % ND.  Every wave issues NGRP pieces per group whatever the tile indices (an empty descriptor outside [0, n_it): zeros
    // into a slot nobody reads), so the counted wait at the top of a body stays exact.  The 4-byte piece covers 64 queries: the second
    // half belongs to the next tile (or is past the sequence: zeros) and is not read.
    Walk wq{0, 0, qt_first, 0}, wd{-1, 0, qt_first, ND - 1};
    struct Grp { SfBufB q, dout, aux; unsigned qoff, dooff, auxoff; char* qdst; char* ddst; char* adst; };
    auto group = [&]() SF_LAMBDA_INLINE {      // descriptors / offsets / destinations of the next group (scalar work, once per body)
        const bool rq = real(wq), rd = real(wd);
        Grp r;
        r.qdst = q_ring + wq.slot * SLOT;
        r.ddst = do_ring + wd.slot * SLOT;
        r.q = sf_bufb_if(rows_buf<HD>(qb_base + wq.hh * HD, p.ldq, S), rq);
        r.dout = sf_bufb_if(rows_buf<HD>(dob_base + wd.hh * HD, p.lddo, S), rd);
        r.qoff = (unsigned)wq.qt * qtile;
        r.dooff = (unsigned)wd.qt * dotile;
        if (wave & 1) {     // delta(m+3) (waves 1 and 3 write the same bytes), lse(m+2) (waves 0 and 2)
            r.aux = sf_bufb_if(sf_make_bufb(dlt_base + (long)wq.hh * S, (unsigned)S * 4u), rq);
            r.auxoff = (unsigned)(wq.qt * QT + lane) * 4u;
            r.adst = r.qdst + ROWS;
        } else {
            r.aux = sf_bufb_if(sf_make_bufb(lse_base + (long)wd.hh * S, (unsigned)S * 4u), rd);
            r.auxoff = (unsigned)(wd.qt * QT + lane) * 4u;
            r.adst = r.ddst + ROWS;
        }
        advance(wq, NQ);
        advance(wd, ND);
        return r;
    };
    auto piece = [&](const Grp& r, int k) SF_LAMBDA_INLINE {   // piece k of NGRP (compile-time k at every call site)
        if (k < NI) sf_bufb_glds16(r.q, stq.off[k] + r.qoff, r.qdst + (stq.piece0 + k) * 1024);
        else if (k < 2 * NI) sf_bufb_glds16(r.dout, stdo.off[k - NI] + r.dooff, r.ddst + (stdo.piece0 + k - NI) * 1024);
        else sf_bufb_glds4(r.aux, r.auxoff, r.adst);
    };
    auto stage = [&]() SF_LAMBDA_INLINE {
        const Grp r = group();
        static_for<0, NGRP>([&](auto K) SF_LAMBDA_INLINE { piece(r, decltype(K)::value); });
    };
    stage();        // groups -3, -2
    stage();

    char* xch = smem + (NQ + ND) * SLOT + sub * XCH;         // + parity * 2 * XCH
    sf_v16f s_cur, s_next, dp_prev, dp_cur;                  // A: S(n), S(n+1)   |   B: dP(n-1), dP(n)
    sf_v8s f[2];                                             // packed P(n) (A) / dS(n-1) (B)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_cur[r] = 0.f; s_next[r] = 0.f; dp_prev[r] = 0.f; dp_cur[r] = 0.f; }
    f[0] = sf_v8s{0, 0, 0, 0, 0, 0, 0, 0};
    f[1] = f[0];

    // the arithmetic's walker (tile n + 1 in body n) and what it leaves behind for tiles n and n - 1: liveness -- a tile whose 32 queries
    // all lie before this pair's keys contributes nothing (wave-uniform, the same for both roles) -- and the first query of tile n
    Walk wc{0, 0, qt_first, 0};
    bool lv_0 = false, lv_m1 = false;
    int q0_0 = 0, slot_q_m1 = NQ - 2, slot_do_0 = ND - 1, par = 1;     // body -1: Q(-2) -> slot 3, dO(-1) -> slot 2, parity of n = -1
    for (int n = -1; n <= n_it; ++n) {
        sf_wait_vmcnt<NGRP>();  // group n - 2 landed (Q(n+1), delta(n+1), dO(n), lse(n)); group n - 1 may be in flight
        sf_syncthreads();       // ... for everyone; Q slot (n+3) % 5, dO slot (n+2) % 3 and exchange parity n & 1 are no longer being read
        // group n's DMA pieces are fillers of the gradient phase's first slots (a body without one issues them on the spot)
        const Grp grp = group();
        auto dma = [&](auto Sl) SF_LAMBDA_INLINE {
            constexpr int k = decltype(Sl)::value;
            if constexpr (k < NGRP) piece(grp, k);
        };
        auto all_pieces = [&]() SF_LAMBDA_INLINE { static_for<0, NGRP>([&](auto K) SF_LAMBDA_INLINE { piece(grp, decltype(K)::value); }); };
        const char* q_m1 = q_ring + slot_q_m1 * SLOT;                    // Q(n-1) | delta(n-1)
        const char* q_p1 = q_ring + wc.slot * SLOT;                      // Q(n+1)
        const char* do_0 = do_ring + slot_do_0 * SLOT;                   // dO(n) | lse(n)
        const int q0_p1 = wc.qt * QT;
        const bool lv_p1 = real(wc) && q0_p1 + QT - 1 >= kw0;
        if (role == 0) {
            // ---- A: S(n+1) beside exp(n); then dV(n)
            const bool do_s = lv_p1, do_e = lv_0;
            const int q0 = q0_0;
            const int lo = key_lo - q0 - 4 * hi, up = S - q0 - 4 * hi;
            const float* ll = reinterpret_cast<const float*>(do_0 + ROWS);
            char* xw = xch + par * 2 * XCH + lane * 16;
            sf_v4f l4[4];
            // exp(n) as fillers of the score slots: lse reads in slots 0, 1; the 16 elements over slots 3 .. 14; P to the partner (fp32, this
            // lane's own 16 bytes of row block j) after every fourth, packed for the own dV MFMAs after every eighth.  MASK: only tiles that
            // touch the diagonal, the padding or the end of the sequence pay for the select (wave-uniform)
            auto fill_for = [&](auto MaskTag) SF_LAMBDA_INLINE {
                return [&](auto Sl) SF_LAMBDA_INLINE {
                    constexpr bool MASK = decltype(MaskTag)::value;
                    constexpr int sl = decltype(Sl)::value;
                    if constexpr (sl < 2) {                          // lse of the tile's rows 8 j + 4 hi + 0..3
                        l4[2 * sl] = *reinterpret_cast<const sf_v4f*>(ll + 8 * (2 * sl) + 4 * hi);
                        l4[2 * sl + 1] = *reinterpret_cast<const sf_v4f*>(ll + 8 * (2 * sl + 1) + 4 * hi);
                    }
                    if constexpr (sl >= 3 && sl < 3 + 12) {          // 16 elements over 12 slots
                        static_for<(sl - 3) * 16 / 12, (sl - 2) * 16 / 12>([&](auto EE) SF_LAMBDA_INLINE {
                            constexpr int E = decltype(EE)::value, j = E / 4, t = E % 4;
                            float x = fmaf(s_cur[E], sc, -kLog2e * l4[j][t]);
                            if constexpr (MASK) {
                                const int C = 8 * j + t;
                                x = (C >= lo && C < up) ? x : -INFINITY;     // exp2(-inf) == 0: masked, padded, out of range
                            }
                            s_cur[E] = sf_exp2_raw(x);
                            if constexpr (E % 4 == 3)
                                *reinterpret_cast<sf_v4f*>(xw + j * 1024) = sf_v4f{s_cur[E - 3], s_cur[E - 2], s_cur[E - 1], s_cur[E]};
                            if constexpr (E % 8 == 7) f[E / 8] = pack_bf16x8(s_cur, E - 7);
                        });
                    }
                };
            };
            const bool need_mask = (q0 < kw0 + 32) || (kw0 + 31 >= kvlen) || (q0 + QT - 1 >= S);      // wave-uniform
            // S(n+1) = Q(n+1) . K^T beside exp(n); then dV^T += dO(n)^T . P(n)
            if (need_mask) pair_body<HD>(bank, do_s, do_e, q_p1, do_0, fo, s_next, f, fill_for(std::true_type{}), dma, all_pieces);
            else pair_body<HD>(bank, do_s, do_e, q_p1, do_0, fo, s_next, f, fill_for(std::false_type{}), dma, all_pieces);
            s_cur = s_next;
        } else {
            // ---- B: dP(n) beside dS(n-1); then dK(n-1)
            const bool do_s = lv_0, do_e = lv_m1;
            const float* dd = reinterpret_cast<const float*>(q_m1 + ROWS);
            const char* xr = xch + (par ^ 1) * 2 * XCH + lane * 16;
            sf_v4f d4[4], p4[4];
            auto fill = [&](auto Sl) SF_LAMBDA_INLINE {
                constexpr int sl = decltype(Sl)::value;
                if constexpr (sl < 2) {
                    d4[2 * sl] = *reinterpret_cast<const sf_v4f*>(dd + 8 * (2 * sl) + 4 * hi);
                    d4[2 * sl + 1] = *reinterpret_cast<const sf_v4f*>(dd + 8 * (2 * sl + 1) + 4 * hi);
                }
                if constexpr (sl >= 2 && sl < 4) {
                    p4[2 * (sl - 2)] = *reinterpret_cast<const sf_v4f*>(xr + (2 * (sl - 2)) * 1024);
                    p4[2 * (sl - 2) + 1] = *reinterpret_cast<const sf_v4f*>(xr + (2 * (sl - 2) + 1) * 1024);
                }
                if constexpr (sl >= 6 && sl < 6 + 8) {
                    static_for<2 * (sl - 6), 2 * (sl - 6) + 2>([&](auto EE) SF_LAMBDA_INLINE {
                        constexpr int E = decltype(EE)::value, j = E / 4, t = E % 4;
                        dp_prev[E] = p4[j][t] * (dp_prev[E] - d4[j][t]);       // dS
                        if constexpr (E % 8 == 7) f[E / 8] = pack_bf16x8(dp_prev, E - 7);
                    });
                }
            };
            // dP(n) = dO(n) . V^T beside dS(n-1); then dK^T += Q(n-1)^T . dS(n-1)
            pair_body<HD>(bank, do_s, do_e, do_0, q_m1, fo, dp_cur, f, fill, dma, all_pieces);
            dp_prev = dp_cur;
        }
        lv_m1 = lv_0;
        lv_0 = lv_p1;
        q0_0 = q0_p1;
        advance(wc, NQ);
        if (++slot_q_m1 == NQ) slot_q_m1 = 0;
        if (++slot_do_0 == ND) slot_do_0 = 0;
        par ^= 1;
    }
    sf_wait_vm0();   // (the stand-in pieces of the last bodies are still in flight: LDS must not be released under them)
    bank.drain();
    if (!kok) return;
    const bool split = p.hsplit > 1;      // partial sums of this head slice: written, not accumulated (attn_dkv_reduce_kernel adds them)
    float* orow = split ? (role == 0 ? p.part_v : p.part_k) + (long)hs * p.part_stride + krow * ((long)p.nkv * HD) + g * HD
                        : (role == 0 ? p.dv : p.dk) + krow * p.lddk + g * HD;
    const float oscale = role == 0 ? 1.0f : p.scale;
    static_for<0, DB>([&](auto D) SF_LAMBDA_INLINE {
        constexpr int d = decltype(D)::value;
        const sf_v16f acc = bank.template get_grad<d>();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = d * 32 + 8 * j + 4 * hi;
            sf_v4f a = sf_v4f{0.f, 0.f, 0.f, 0.f};
            if (!split) a = *reinterpret_cast<const sf_v4f*>(orow + col);
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] += acc[4 * j + t] * oscale;
            *reinterpret_cast<sf_v4f*>(orow + col) = a;
        }
    });
}

}  // namespace

namespace sfattn {

// grid / shared memory as the role-split kernel it replaces: 64 keys per workgroup, `hsplit` head slices (p.hsplit set by the caller)
int attn_bwd_dkv_w1_launch(const AttnBwdArgs& p, int hd, void* stream) {
    SF_CHECK_ARG(hd == 256, "attn_bwd_dkv_w1: head_dim 256 only");
    constexpr int HD = 256;
    constexpr int SMEM = (5 + 3) * (32 * HD * 2 + 256) + 2 * 2 * 32 * 32 * 4;      // Q ring, dO ring, exchange: 146 KiB
    const int hsplit = p.hsplit > 1 ? p.hsplit : 1;
    dim3 grid((unsigned)((long)((p.S + 63) / 64) * p.nkv * p.B * hsplit));
    SF_ALLOW_SMEM((attn_bwd_dkv_pair_kernel<HD>), SMEM);
    SF_LAUNCH((attn_bwd_dkv_pair_kernel<HD>), grid, dim3(256), SMEM, stream, p);
    return 0;
}

}  // namespace sfattn
