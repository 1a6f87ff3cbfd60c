// Launcher of the 4-wave NT GEMM (kernel: sf_gemm256w4_kernel.h; instantiations: sf_gemm256w4_i*.hip).
#include "sf_gemm256w4_kernel.h"

int sf_gemm_nt_128_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype, void* stream);

// launched by sf_gemm_nt (sf_gemm.hip) when the shape qualifies
int sf_gemm_nt_256w4_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype,
                            void* stream) {
    const int M = e.M, N = e.N;
    GemmW4Args p;
    p.A = (const sf_bf16*)A; p.lda = lda;
    p.B = (const sf_bf16*)B; p.ldb = ldb;
    p.e = e;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + TM - 1) / TM;
    p.tiles_n = (N + TN - 1) / TN;
    // GM = m-tiles per group of the tile order (a group walks its m-tiles fastest, then the n-tiles; an XCD runs 32 consecutive
    // tiles at a time, so GM sets how many A / B panels its L2 holds at once).  A per-shape choice was measured and rejected
    // (round 3): on one box GM = 1 gained 3-4 % on narrow-N / mid-K shapes and 8 gained 0.5-0.9 % on wide N, on the next box the
    // same sweep showed nothing, and in-step the per-shape table was 0.3 ms SLOWER than 4 everywhere (profiles/old/r3_gemm_gm_ab.jsonl).
    p.gm = sf_knob("SF_GEMM_GM", 4);
    if (p.gm < 1) p.gm = 1;
    long nblk = (long)p.tiles_m * p.tiles_n;
    // The last, partly filled round of the persistent walk: lm_head forward (16384 x 32000: 8000 tiles) leaves 64 tiles for a 32nd
    // round on 256 CUs -- 192 CUs idle for a whole tile time (2.3 % of the launch).  When the remainder is a whole number of
    // COLUMN tiles and small (<= a quarter of the CUs), those columns are peeled off: the persistent kernel walks an exact number of
    // rounds and the peeled columns run as 128 x 128 tiles (4 x the workgroups, a quarter of the work each: one short round of the
    // generic kernel, measured 0.050 ms against the 0.098 ms round it replaces; tools/tail_bench.py).
    // (round 4: the row-addend form too -- at small M its grids are the ones that leave such rounds: 4096 x 5120, cfg 4 at bs 1 x 4096,
    // is 320 tiles = 1 round + 64; 2048 x 9216, DeepSeek-V3 dims at bs 1 x 2048, 288 = 1 + 32.  The addend's columns move with C's.)
    if (!p.e.sw_gu && !p.e.sw_dgu && !p.e.red_part && sf_knob("SF_GEMM_PEEL", 1)) {
        const long cus = sf_w4_grid(1L << 30);
        const long rem = nblk % cus;
        if (nblk > cus && rem != 0 && rem * 4 <= cus && rem % p.tiles_m == 0 && rem / p.tiles_m < p.tiles_n) {
            const int t = (int)(rem / p.tiles_m);               // column tiles to peel
            const int n_main = (p.tiles_n - t) * TN;
            SfGemmEpi et = e;
            et.N = N - n_main;
            et.C = c_dtype == SF_F32 ? (void*)((float*)e.C + n_main) : (void*)((sf_bf16*)e.C + n_main);
            if (e.R) et.R = e.R + n_main;
            if (e.Cadd) et.Cadd = e.Cadd + n_main;
            if (int st = sf_gemm_nt_128_launch(A, lda, (const sf_bf16*)B + (long)n_main * ldb, ldb, K, et, c_dtype, stream)) return st;
            p.e.N = n_main;
            p.N = n_main;
            p.tiles_n -= t;
            nblk = (long)p.tiles_m * p.tiles_n;
        }
    }
    // ragged M: the last row tile is shifted up to end at row M (GemmW4Args::mshift) unless the epilogue reads what it writes
    p.mshift = (M > TM && M % TM != 0 && p.e.beta == 0.f && (const void*)p.e.R != (const void*)p.e.C &&
                (!p.e.sw_gu || (const void*)p.e.sw_gu != (const void*)p.e.sw_dgu) && sf_knob("SF_GEMM_MSHIFT", 1)) ? 1 : 0;
    p.epi_direct = sf_knob("SF_GEMM_EPI_DIRECT", 1);
    // the 32-bit per-lane byte offsets of the buffer-descriptor DMA cover one 256-row tile of either operand
    SF_CHECK_ARG(256L * lda * 2 < (1L << 31) && 256L * ldb * 2 < (1L << 31), "sf_gemm_nt: row stride too large for the 256-tile kernel");
    if (p.e.Cadd) SF_CHECK_ARG(p.e.alpha == 1.0f, "sf_gemm_nt_rowadd: the 4-wave kernel needs alpha == 1");
    // (sw_dgu without sw_gu = the SwiGLU-forward form: C = gate|up [M, N = 2I], sw_dgu = act [M, I])
    const int add = p.e.Cadd ? 1 : (p.e.sw_gu ? 2 : (p.e.sw_dgu ? 3 : (p.e.red_part ? 4 : 0))), f32 = c_dtype == SF_F32 ? 1 : 0;
    SF_CHECK_ARG(add != 4 || (!f32 && p.e.alpha == 1.f && p.e.beta == 0.f && !p.e.R && p.e.red_n0 % TN == 0 && p.e.red_stride >= (N - p.e.red_n0 + 127) / 128),
                 "sf_gemm_nt_teacher: bf16 logits, alpha 1, reduced range on a tile boundary");
    SF_CHECK_ARG(add != 3 || (!f32 && (M % TM == 0 || p.mshift) && N % TN == 0 && p.e.alpha == 1.f && p.e.beta == 0.f && !p.e.R &&
                              (p.e.ldc & 7) == 0 && ((size_t)p.e.C & 15) == 0 && (N / 2 + 256L) * ldb * 2 < (1L << 31)),
                 "sf_gemm_nt_swiglu_fwd: the fused form takes whole bf16 tiles only");
    SF_CHECK_ARG(add != 2 || (!f32 && (M % TM == 0 || p.mshift) && N % TN == 0 && p.e.alpha == 1.f && p.e.beta == 0.f && !p.e.R),
                 "sf_gemm_nt_swiglu_bwd: the fused form takes whole bf16 tiles only");
    // which operand's LDS half is released and re-staged first: B for narrow N, A for wide N (measured, see the header)
    int sched = N <= 8192 ? 12 : 13;
#define SF_W4_CASE(F32, ADD, SCHED) \
    if (f32 == F32 && add == ADD && sched == SCHED) return sf_w4_launch_##F32##_##ADD##_##SCHED(p, nblk, stream);
#define SF_W4_LOCAL(F32, ADD, SCHED)                                                                                      \
    if (f32 == F32 && add == ADD && sched == SCHED) {                                                                    \
        SF_W4_SMEM((gemm_nt_256w4_kernel<F32, ADD, SCHED>), w4_smem_bytes<ADD>());                                       \
        SF_LAUNCH((gemm_nt_256w4_kernel<F32, ADD, SCHED>), dim3(sf_w4_grid(nblk, ADD == 1)), dim3(256), w4_smem_bytes<ADD>(), stream, p);   \
        return sf_check_launch("sf_gemm_nt(256w4 tools)");                                                               \
    }
#ifdef SF_ABLATE   // A/B variants, tools only (tools/experiments/sf_gemm256w4_*.inc)
    p.cyc = sf_knob("SF_GEMM_CYC", 0);
    p.stagger = sf_knob("SF_GEMM_STAGGER", 0);
    {
        const int ks = sf_knob("SF_GEMM_SCHED", -1);
        if (ks >= 0) {
            SF_CHECK_ARG(!f32 && !add, "SF_GEMM_SCHED: plain bf16 output only (A/B knob)");
            sched = ks;
            SF_W4_LOCAL(0, 0, 0) SF_W4_LOCAL(0, 0, 2) SF_W4_LOCAL(0, 0, 3) SF_W4_LOCAL(0, 0, 5) SF_W4_LOCAL(0, 0, 9) SF_W4_LOCAL(0, 0, 11)
        }
        const int abl = sf_knob("SF_GEMM_ABL", 0);
#define SF_ABL_CASE(V)                                                                                                    \
    if (abl == V) {                                                                                                      \
        SF_W4_SMEM((gemm_nt_256w4_kernel<0, 0, 0, V>));                                                                  \
        SF_LAUNCH((gemm_nt_256w4_kernel<0, 0, 0, V>), dim3(sf_w4_grid(nblk)), dim3(256), kW4SmemBytes, stream, p);        \
        return sf_check_launch("abl");                                                                                   \
    }
        SF_ABL_CASE(1) SF_ABL_CASE(2) SF_ABL_CASE(3) SF_ABL_CASE(8) SF_ABL_CASE(16)
#undef SF_ABL_CASE
        // the same decomposition for the product plan (B first): SF_GEMM_ABL12 = 1 no reads | 2 no DMA | 4 no barriers | 8 no waits
        const int abl12 = sf_knob("SF_GEMM_ABL12", 0);
#define SF_ABL12_CASE(V)                                                                                                  \
    if (abl12 == V) {                                                                                                    \
        SF_W4_SMEM((gemm_nt_256w4_kernel<0, 0, 12, V>));                                                                 \
        SF_LAUNCH((gemm_nt_256w4_kernel<0, 0, 12, V>), dim3(sf_w4_grid(nblk)), dim3(256), kW4SmemBytes, stream, p);       \
        return sf_check_launch("abl12");                                                                                 \
    }
        SF_ABL12_CASE(1) SF_ABL12_CASE(2) SF_ABL12_CASE(4) SF_ABL12_CASE(8) SF_ABL12_CASE(12) SF_ABL12_CASE(14) SF_ABL12_CASE(15)
#undef SF_ABL12_CASE
    }
    if (int st = sf_gemm_nt_256w4_variants_launch(p, nblk, c_dtype, stream); st != -1) return st;
#endif
    SF_W4_CASE(0, 0, 12) SF_W4_CASE(0, 0, 13) SF_W4_CASE(1, 0, 12) SF_W4_CASE(1, 0, 13)
    SF_W4_CASE(0, 1, 12) SF_W4_CASE(0, 1, 13) SF_W4_CASE(1, 1, 12) SF_W4_CASE(1, 1, 13)
    SF_W4_CASE(0, 2, 12) SF_W4_CASE(0, 2, 13)
    SF_W4_CASE(0, 3, 12) SF_W4_CASE(0, 3, 13)
    SF_W4_CASE(0, 4, 12) SF_W4_CASE(0, 4, 13)
#undef SF_W4_CASE
#undef SF_W4_LOCAL
    SF_CHECK_ARG(false, "sf_gemm_nt(256w4): no kernel for this configuration");
}

// ---- split-K for under-filled grids (round 4) -----------------------------------------------------------------------------------
// With at most half as many 256 x 256 tiles as CUs -- every N = H GEMM of a bs 1 x 4096 recipe at H = 2048 is 16 x 8 = 128 tiles on 256
// CUs -- half the chip idles for the whole launch, and the 128 x 128 kernel that fills it runs at 0.85 - 0.99 PFLOP/s where this kernel
// reaches 1.4.  When the caller provides a workspace, K is cut into 2 (or 4) chunks instead: tiles x chunks work units of the SAME
// kernel (fp32 plain form, one unit per CU), fp32 partials, and a fixed-order reduce (deterministic) that rounds once and applies the
// residual like the plain epilogue does.  Returns -1 when the shape does not qualify (the caller dispatches as before).
namespace {
template <int OUT_F32>
SF_GLOBAL void nt_splitk_reduce_kernel(const float* ws, int ksplit, long part_stride, void* C, long ldc, const sf_bf16* R, long ldr, int M, int N,
                                       const float* Cadd, long ldadd, int add_S, int add_Spad, int add_off) {
    const long n8 = N / 8, total = (long)M * n8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / n8;
        const int n = (int)(i - m * n8) * 8;
        float a[8];
        SfVec8<float>::ld(ws + m * N + n, a);
        for (int y = 1; y < ksplit; ++y) {
            float b[8];
            SfVec8<float>::ld(ws + y * part_stride + m * N + n, b);
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] += b[r];
        }
        if (Cadd) {       // row-addend form: the fp32 addend joins before the single rounding, as in the fused epilogue
            const long bb = m / add_S;
            float b[8];
            SfVec8<float>::ld(Cadd + (bb * add_Spad + (m - bb * add_S) + add_off) * ldadd + n, b);
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] += b[r];
        }
        if (OUT_F32) {
            SfVec8<float>::st((float*)C + m * ldc + n, a);
        } else {
            if (R) {      // round the projection first, then add the residual (bf16 + bf16), as sf_gemm_store4 does
                float rr[8];
                SfVec8<sf_bf16>::ld(R + m * ldr + n, rr);
#pragma unroll
                for (int r = 0; r < 8; ++r) a[r] = sf_round_bf(a[r]) + rr[r];
            }
            SfVec8<sf_bf16>::st((sf_bf16*)C + m * ldc + n, a);
        }
    }
}
}  // namespace

int sf_gemm_nt_256w4_splitk_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype,
                                   float* workspace, long workspace_floats, void* stream) {
    const int M = e.M, N = e.N;
    if (!workspace || ((size_t)workspace & 15) || e.sw_gu || e.sw_dgu || e.red_part || e.alpha != 1.0f || e.beta != 0.0f) return -1;
    if (e.Cadd && ((e.ldadd & 3) || ((size_t)e.Cadd & 15) || e.R)) return -1;       // (the reduce reads the addend in 16-byte pieces)
    // (M need not be whole tiles -- real batches are ragged: the partials are laid out in whole row tiles, see the kernel's epilogue)
    if (M < 8 || N % TN || K % TK || (e.ldc & 7) || ((size_t)e.C & 15) || (e.R && ((e.ldr & 7) || ((size_t)e.R & 15)))) return -1;
    const int tiles_m = (M + TM - 1) / TM;
    const long Mpad = (long)tiles_m * TM;
    const long tiles = (long)tiles_m * (N / TN);
    const long cus = sf_w4_grid(1L << 30);
#ifdef SF_EMU
    const int min_chunk = 2 * TK;          // (interpreter: the multi-unit path at test sizes)
#else
    const int min_chunk = 16 * TK;         // a chunk of at least 1024: prologue + epilogue stay a small part of a unit
#endif
    // chunks: c = 2 .. 8 equal runs of K-tiles (the last one may be shorter), one round of tiles x c units; the cheapest by a simple
    // model -- a unit is its K-tiles at ~1.3 us each + ~12 us of prologue / epilogue, the reduce streams (4 c + 4) bytes per element at
    // ~4 TB/s.  (5 row tiles, 1025 .. 1280 tokens at batch 1: 80 tiles take 3 chunks = 240 units instead of 2 = 160.)
    const int kt = K / TK;
    int ksplit = 0, chunk_kt = 0;
    double best = 0;
    for (int c = 2; c <= 8; ++c) {
        const int ck = (kt + c - 1) / c;
        if (tiles * c > cus || (long)(c - 1) * ck >= kt || ck * TK < min_chunk || workspace_floats < (long)c * Mpad * N) continue;
        const double cost = ck * 1.3 + 12.0 + (4.0 * c + 4.0) * (double)Mpad * N / 4e6;
        if (!ksplit || cost < best) { ksplit = c; chunk_kt = ck; best = cost; }
    }
    if (!ksplit) return -1;
    GemmW4Args p;
    p.A = (const sf_bf16*)A; p.lda = lda;
    p.B = (const sf_bf16*)B; p.ldb = ldb;
    p.e = e;
    p.e.C = workspace; p.e.ldc = N; p.e.R = nullptr; p.e.ldr = 0;
    p.e.Cadd = nullptr;                     // (the addend of the row-addend form joins in the reduce)
    p.M = M; p.N = N; p.K = chunk_kt * TK;
    p.k_total = K;
    p.tiles_m = tiles_m; p.tiles_n = N / TN;
    p.gm = 4;
    p.ksplit = ksplit; p.ks_a = p.K; p.ks_b = p.K; p.ks_c = Mpad * N;
#ifdef SF_ABLATE
    p.cyc = 0; p.stagger = 0;
#endif
    SF_CHECK_ARG(256L * lda * 2 < (1L << 31) && 256L * ldb * 2 < (1L << 31), "sf_gemm_nt: row stride too large for the 256-tile kernel");
    const long units = tiles * ksplit;
    const int sched = N <= 8192 ? 12 : 13;
    int st = sched == 12 ? sf_w4_launch_1_0_12(p, units, stream) : sf_w4_launch_1_0_13(p, units, stream);
    if (st) return st;
    const long work = (long)M * (N / 8);
    const int rgrid = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
    if (c_dtype == SF_F32)
        SF_LAUNCH((nt_splitk_reduce_kernel<1>), dim3(rgrid), dim3(256), 0, stream, (const float*)workspace, ksplit, Mpad * N, e.C, e.ldc, e.R, e.ldr, M, N,
                  e.Cadd, e.ldadd, e.add_S, e.add_Spad, e.add_off);
    else
        SF_LAUNCH((nt_splitk_reduce_kernel<0>), dim3(rgrid), dim3(256), 0, stream, (const float*)workspace, ksplit, Mpad * N, e.C, e.ldc, e.R, e.ldr, M, N,
                  e.Cadd, e.ldadd, e.add_S, e.add_Spad, e.add_off);
    return sf_check_launch("sf_gemm_nt(split-K)");
}
