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

constexpr int kAhead = 8;

// score phase: c = rows(lds_rows) . bank fragments (KS MFMAs); `fill(slot)` supplies the fillers.  MFMA = false: the fillers alone (the
// pipeline's last bodies: nothing left to score, the previous tile's arithmetic still due)
template <int HD, bool MFMA, class Bank, class Fill>
SF_DEVICE void score_phase(Bank& bank, const char* lds_rows, const FragOff<HD>& fo, sf_v16f& c, Fill&& fill) {
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
        fill(std::integral_constant<int, S>{});
        SF_SCHED_FENCE();
    });
}

// gradient phase: acc[d] += tr(lds_x)(d, 16 jp) . f[jp]   (2 DB MFMAs)
template <int HD, class Bank, class Fill>
SF_DEVICE void grad_phase(Bank& bank, const char* lds_x, const FragOff<HD>& fo, const sf_v8s (&f)[2], Fill&& fill) {
    constexpr int DB = HD / 32, NG = 2 * DB;
    static_assert(NG >= kAhead, "plan: head_dim >= 128");
    sf_v8s gf[kAhead];
    static_for<0, kAhead>([&](auto I) SF_LAMBDA_INLINE {
        constexpr int M = decltype(I)::value;
        gf[M] = frag_tr<HD>(lds_x, M % DB, 16 * (M / DB), fo);
    });
    SF_SCHED_FENCE();
    static_for<0, NG>([&](auto I) SF_LAMBDA_INLINE {
        constexpr int M = decltype(I)::value, jp = M / DB, d = M % DB;
        bank.template mfma_grad<d>(gf[M % kAhead], f[jp]);
        if constexpr (M + kAhead < NG) gf[M % kAhead] = frag_tr<HD>(lds_x, (M + kAhead) % DB, 16 * ((M + kAhead) / DB), fo);
        fill(std::integral_constant<int, M>{});
        SF_SCHED_FENCE();
    });
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
    auto q0_of = [&](int t) SF_LAMBDA_INLINE { return (qt_first + (t % per_head)) * QT; };
    char* const q_ring = smem;
    char* const do_ring = smem + NQ * SLOT;
    // DMA group m (issued in body m; before the loop for m = -3, -2): Q(m+3) + delta(m+3) into Q slot (m+3) % NQ, dO(m+2) + lse(m+2) into dO
    // slot (m+2) % ND.  Every wave issues NGRP pieces per group whatever the tile indices (an empty descriptor outside [0, n_it): zeros
    // into a slot nobody reads), so the counted wait at the top of a body stays exact.  The 4-byte piece covers 64 queries: the second
    // half belongs to the next tile (or is past the sequence: zeros) and is not read.
    auto coords = [&](int t, bool& real, int& hh, int& qt) SF_LAMBDA_INLINE {
        real = t >= 0 && t < n_it;
        const int tt = real ? t : 0;
        hh = tt / per_head;
        qt = qt_first + (tt - hh * per_head);
    };
    struct Grp { SfBufB q, dout, aux; unsigned qoff, dooff, auxoff; char* qdst; char* ddst; char* adst; };
    auto group = [&](int m) SF_LAMBDA_INLINE {      // descriptors / offsets / destinations of group m (scalar work, once per body)
        bool rq, rd;
        int hq, tq, hd_, td;
        coords(m + 3, rq, hq, tq);
        coords(m + 2, rd, hd_, td);
        Grp r;
        r.qdst = q_ring + ((m + 3 + NQ) % NQ) * SLOT;
        r.ddst = do_ring + ((m + 2 + ND) % ND) * SLOT;
        r.q = sf_bufb_if(rows_buf<HD>(qb_base + hq * HD, p.ldq, S), rq);
        r.dout = sf_bufb_if(rows_buf<HD>(dob_base + hd_ * HD, p.lddo, S), rd);
        r.qoff = (unsigned)tq * qtile;
        r.dooff = (unsigned)td * dotile;
        if (wave & 1) {     // delta(m+3) (waves 1 and 3 write the same bytes), lse(m+2) (waves 0 and 2)
            r.aux = sf_bufb_if(sf_make_bufb(dlt_base + (long)hq * S, (unsigned)S * 4u), rq);
            r.auxoff = (unsigned)(tq * QT + lane) * 4u;
            r.adst = r.qdst + ROWS;
        } else {
            r.aux = sf_bufb_if(sf_make_bufb(lse_base + (long)hd_ * S, (unsigned)S * 4u), rd);
            r.auxoff = (unsigned)(td * QT + lane) * 4u;
            r.adst = r.ddst + ROWS;
        }
        return r;
    };
    auto piece = [&](const Grp& r, int k) SF_LAMBDA_INLINE {   // piece k of NGRP (compile-time k at every call site)
        if (k < NI) sf_bufb_glds16(r.q, stq.off[k] + r.qoff, r.qdst + (stq.piece0 + k) * 1024);
        else if (k < 2 * NI) sf_bufb_glds16(r.dout, stdo.off[k - NI] + r.dooff, r.ddst + (stdo.piece0 + k - NI) * 1024);
        else sf_bufb_glds4(r.aux, r.auxoff, r.adst);
    };
    auto stage = [&](int m) SF_LAMBDA_INLINE {
        const Grp r = group(m);
        static_for<0, NGRP>([&](auto K) SF_LAMBDA_INLINE { piece(r, decltype(K)::value); });
    };
    stage(-3);
    stage(-2);

    char* xch = smem + (NQ + ND) * SLOT + sub * XCH;         // + parity * 2 * XCH
    sf_v16f s_cur, s_next, dp_prev, dp_cur;                  // A: S(n), S(n+1)   |   B: dP(n-1), dP(n)
    sf_v8s f[2];                                             // packed P(n) (A) / dS(n-1) (B)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_cur[r] = 0.f; s_next[r] = 0.f; dp_prev[r] = 0.f; dp_cur[r] = 0.f; }
    f[0] = sf_v8s{0, 0, 0, 0, 0, 0, 0, 0};
    f[1] = f[0];

    for (int n = -1; n <= n_it; ++n) {
        sf_wait_vmcnt<NGRP>();  // group n - 2 landed (Q(n+1), delta(n+1), dO(n), lse(n)); group n - 1 may be in flight
        sf_syncthreads();       // ... for everyone; Q slot (n+3) % 5, dO slot (n+2) % 3 and exchange parity n & 1 are no longer being read
        // group n's DMA pieces are fillers of the gradient phase's first slots (a body without one issues them on the spot)
        const Grp grp = group(n);
        auto dma = [&](auto Sl) SF_LAMBDA_INLINE {
            constexpr int k = decltype(Sl)::value;
            if constexpr (k < NGRP) piece(grp, k);
        };
        const char* q_m1 = q_ring + ((n - 1 + NQ) % NQ) * SLOT;          // Q(n-1) | delta(n-1)
        const char* q_p1 = q_ring + ((n + 1 + NQ) % NQ) * SLOT;          // Q(n+1)
        const char* do_0 = do_ring + ((n + ND) % ND) * SLOT;             // dO(n) | lse(n)
        // a tile whose 32 queries all lie before this pair's keys contributes nothing (wave-uniform, the same for both roles)
        auto live = [&](int t) SF_LAMBDA_INLINE { return t >= 0 && t < n_it && q0_of(t) + QT - 1 >= kw0; };
        if (role == 0) {
            // ---- A: S(n+1) beside exp(n); then dV(n)
            const bool do_s = live(n + 1), do_e = live(n);
            const int q0 = do_e ? q0_of(n) : 0;
            const int lo = key_lo - q0 - 4 * hi, up = S - q0 - 4 * hi;
            const float* ll = reinterpret_cast<const float*>(do_0 + ROWS);
            char* xw = xch + (n & 1) * 2 * XCH + lane * 16;
            sf_v4f l4[4];
            // exp(n) as fillers of the score slots: lse reads in slots 0, 1; the 16 elements over slots 4 .. 11; P to the partner (fp32, this
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
                    if constexpr (sl >= 4 && sl < 4 + 8) {
                        static_for<2 * (sl - 4), 2 * (sl - 4) + 2>([&](auto EE) SF_LAMBDA_INLINE {
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
            if (do_e || do_s) {
                if (need_mask) {
                    auto fill = fill_for(std::true_type{});
                    if (do_s) score_phase<HD, true>(bank, q_p1, fo, s_next, fill);
                    else score_phase<HD, false>(bank, q_p1, fo, s_next, fill);
                } else {
                    auto fill = fill_for(std::false_type{});
                    if (do_s) score_phase<HD, true>(bank, q_p1, fo, s_next, fill);
                    else score_phase<HD, false>(bank, q_p1, fo, s_next, fill);
                }
            }
            if (do_e) grad_phase<HD>(bank, do_0, fo, f, dma);              // dV^T += dO(n)^T . P(n)
            else static_for<0, NGRP>([&](auto K) SF_LAMBDA_INLINE { piece(grp, decltype(K)::value); });
            s_cur = s_next;
        } else {
            // ---- B: dP(n) beside dS(n-1); then dK(n-1)
            const bool do_s = live(n), do_e = live(n - 1);
            const float* dd = reinterpret_cast<const float*>(q_m1 + ROWS);
            const char* xr = xch + ((n - 1) & 1) * 2 * XCH + lane * 16;
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
            if (do_e || do_s) {
                if (do_s) score_phase<HD, true>(bank, do_0, fo, dp_cur, fill);
                else score_phase<HD, false>(bank, do_0, fo, dp_cur, fill);
            }
            if (do_e) grad_phase<HD>(bank, q_m1, fo, f, dma);               // dK^T += Q(n-1)^T . dS(n-1)
            else static_for<0, NGRP>([&](auto K) SF_LAMBDA_INLINE { piece(grp, decltype(K)::value); });
            dp_prev = dp_cur;
        }
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
