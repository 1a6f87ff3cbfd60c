// bf16 MFMA GEMM for the draft layer's dense projections (the only GEMM-shaped work on the
// path: fc, q/k/v, o, gate/up, down, lm_head, teacher head -- SURVEY.md 2.2 G1-G9).
//
//   C[M,N] = alpha * A[M,K] . B[N,K]^T  (+ beta * C)  (+ R[M,N])       A, B bf16, fp32 accumulate
//
// "NT" form: both operands are K-contiguous, which is what nn.Linear's forward is
// (x[M,K] . W[N,K]^T, llama3_eagle.py:555-566,1513-1515,1674-1693).  dgrad and wgrad are
// brought to the same form by the caller with pre-transposed operands (sf_transpose).
//
// Structure (CDNA4): 128x128 output tile per 256-thread workgroup (2x2 waves, each 64x64 as
// 4x4 mfma_f32_16x16x32_bf16 tiles, operands swapped so that a lane owns 4 consecutive
// output columns), BK = 64, operands staged HBM -> LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4) into a double buffer; bank conflicts are broken by an XOR
// swizzle of the 16-byte chunk index applied on the global source address and again on
// the ds_read_b128 address (the LDS image itself must stay lane-linear for LDS-DMA).
// Workgroup ids are remapped so that each XCD (private L2) walks a compact group of tiles.
#include "sf_api_internal.h"
#include "sf_util.h"
#include "sf_gemm_epilogue.h"
#include <stdlib.h>

namespace {

#ifdef SF_EMU
static const sf_bf16 sf_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
__device__ const sf_bf16 sf_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB

struct GemmArgs {
    const sf_bf16* A;
    long lda;
    const sf_bf16* B;
    long ldb;
    SfGemmEpi e;
    int M, N, K;
    int tiles_m, tiles_n;
};

// XCD-aware, grouped tile order: blocks land on XCD (id % 8); give every XCD a contiguous
// run of the tile sequence (bijective for any grid size), and order the sequence in
// column groups of 8 tile-rows so one XCD's L2 sees few distinct A/B panels.
SF_DEVICE void tile_coords(int bid, int nblk, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int seq = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int g = seq / per_group;
    const int first_m = g * GM;
    const int gsize = (tiles_m - first_m < GM) ? (tiles_m - first_m) : GM;
    const int in_g = seq - g * per_group;
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
}

template <int OUT_F32>
SF_GLOBAL void SF_LAUNCH_BOUNDS(256, 2) gemm_nt_kernel(GemmArgs p) {
    SF_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = sf_wave_id();
    const int wr = wave >> 1, wc = wave & 1;
    int tm, tn;
    tile_coords((int)blockIdx.x, (int)gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (p.K + BK - 1) / BK;

    sf_v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sf_v4f{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * kStageBytes;
        char* lb = la + BM * BK * 2;
        const int k0 = kt * BK;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int rr = (wave * 4 + t) * 8 + (lane >> 3);
            const int pc = lane & 7;
            const int kk = k0 + ((pc ^ (rr & 7)) << 3);
            const int gm = m0 + rr, gn = n0 + rr;
            const sf_bf16* sa = (gm < p.M && kk < p.K) ? p.A + (long)gm * p.lda + kk : sf_zero16;
            const sf_bf16* sb = (gn < p.N && kk < p.K) ? p.B + (long)gn * p.ldb + kk : sf_zero16;
            sf_glds16(sa, la + (wave * 4 + t) * 1024);  // uniform base; lane i lands at +16*i
            sf_glds16(sb, lb + (wave * 4 + t) * 1024);
        }
    };

    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        sf_wait_vm0();
        sf_syncthreads();
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
        const char* la = smem + (kt & 1) * kStageBytes;
        const char* lb = la + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            sf_v8s a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wr * 64 + i * 16 + (lane & 15);
                const int rb = wc * 64 + i * 16 + (lane & 15);
                const int lc = ks * 4 + (lane >> 4);
                a[i] = *reinterpret_cast<const sf_v8s*>(la + ra * 128 + ((lc ^ (ra & 7)) << 4));
                b[i] = *reinterpret_cast<const sf_v8s*>(lb + rb * 128 + ((lc ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = sf_mfma16(b[j], a[i], acc[i][j]);  // D[n][m]
        }
    }

    // epilogue: lane owns C[m][n .. n+3], m = 16-row tile row (lane&15), n = 4*(lane>>4)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            sf_gemm_store4<OUT_F32>(p.e, m0 + wr * 64 + i * 16 + (lane & 15), n0 + wc * 64 + j * 16 + 4 * (lane >> 4), v);
        }
}

}  // namespace

int sf_gemm_nt_256_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype, void* stream);
int sf_gemm_nt_256w4_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype, void* stream);
int sf_gemm_nt_128_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype, void* stream);
#ifdef SF_ABLATE
// TOOLS BUILD ONLY (measured and rejected, round 6; DESIGN section 4 "Round 6"): the pair-resident 256 x 128 kernel -- two 4-wave workgroups
// per CU so that one's epilogue runs under the other's K loop (tools/experiments/sf_gemm_p2.inc; tools/p2_ab.py).  SF_GEMM_P2=<bits> per call:
// bit 0 = plain bf16, bit 1 = row-addend, bit 2 = fused d(SwiGLU); -1 from the launcher = the shape / epilogue does not qualify.
#include "../../tools/experiments/sf_gemm_p2.inc"
#endif
// tools build only (-DSF_ABLATE): SF_GEMM_TILE=128 pins the 128x128 kernel
static bool sf_gemm_use_256() {
    static const bool use = sf_knob("SF_GEMM_TILE", 256) != 128;
    return use;
}

#ifndef SF_EMU
static int sf_gemm_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n >= 8 ? n : 256;
    }();
    return cus;
}
#endif

static int sf_gemm_dispatch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype,
                            void* stream) {
    const int M = e.M, N = e.N;
#ifdef SF_ABLATE
    {
        const int p2 = sf_knob("SF_GEMM_P2", 0);
        if ((e.Cadd ? (p2 & 2) : (p2 & 1)) && !e.sw_gu && !e.sw_dgu && !e.red_part) {
            const int st = sf_gemm_nt_p2_launch(A, lda, B, ldb, K, e, c_dtype, stream);
            if (st != -1) return st;
        }
    }
#endif
    // chip-filling shapes (>= one 256x256 tile per CU, long K: every GEMM of the training step) take the 4-wave
    // software-pipelined kernel; smaller ones the 8-wave ping-pong kernel, whose prologue/epilogue is shorter.
    // (tools build: SF_GEMM_W4=0 pins the ping-pong kernel, =1 forces the 4-wave kernel for every 256-tile shape)
    static const int w4_mode = sf_knob("SF_GEMM_W4", -1);
#ifdef SF_EMU   // interpreter build (tests): "chip-filling" means nothing there -- long-K shapes take the 4-wave kernel so that
                // both 256-tile kernels are exercised by the CPU suite
    const bool big = K >= 512;
#else
    // (round 4: from MORE THAN HALF a tile per CU on -- 129 tiles; it was a whole tile per CU.  On grids of 144 ... 288 tiles the 4-wave
    //  kernel is 3 - 26 % faster than the ping-pong kernel at every K tried, tools/w4_small_ab.py, profiles/r4_w4_small_ab.jsonl: DeepSeek-V3
    //  dims at batch 1 are 8 x 28 = 224 tiles and spent 30 % of their step in the ping-pong kernel.  Up to half a tile per CU: the
    //  128 x 128 kernel or split-K, below.)
    const bool big = 2L * ((M + 255) / 256) * ((N + 255) / 256) > sf_gemm_cus() && K >= 512;
#endif
    const bool w4_ok = !(e.Cadd && e.alpha != 1.0f);   // its addend path starts the accumulators from Cadd
#ifndef SF_EMU
    // Under-filled grids (round 4): with at most HALF as many 256 x 256 tiles as CUs -- the N = H GEMMs of a bs 1 x 4096 recipe at
    // H = 2048 are 16 x 8 = 128 tiles on 256 CUs -- the 128 x 128 kernel (4 x the workgroups, two per CU) keeps every CU busy and wins
    // by 8 ... 39 % (4096 x 2048 x {4096, 5120, 6144, 12288, 32000}, 2048 x 4096 x 4096); from 160 tiles up the 256-tile kernels are
    // faster again (tools/small_tile_ab.py, profiles/r4_small_tile_ab.jsonl).
    {
        const int cus = sf_gemm_cus();
        static const int small_tiles = sf_knob("SF_GEMM_SMALL128", 1);
        if (small_tiles && 2L * ((M + 255) / 256) * ((N + 255) / 256) <= cus && sf_gemm_use_256())
            return sf_gemm_nt_128_launch(A, lda, B, ldb, K, e, c_dtype, stream);
    }
#endif
    if (K % 64 == 0 && K >= 64 && M >= 192 && N >= 192 && sf_gemm_use_256() && w4_ok &&
        (w4_mode == 1 || (w4_mode < 0 && big)))
        return sf_gemm_nt_256w4_launch(A, lda, B, ldb, K, e, c_dtype, stream);
    if (K % 64 == 0 && K >= 64 && M >= 192 && N >= 192 && sf_gemm_use_256())
        return sf_gemm_nt_256_launch(A, lda, B, ldb, K, e, c_dtype, stream);
    return sf_gemm_nt_128_launch(A, lda, B, ldb, K, e, c_dtype, stream);
}

// the 128 x 128 kernel (any shape; also the tail launch of the 4-wave kernel's peeled last round, sf_gemm256w4.hip)
int sf_gemm_nt_128_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype, void* stream) {
    GemmArgs p;
    p.A = (const sf_bf16*)A; p.lda = lda;
    p.B = (const sf_bf16*)B; p.ldb = ldb;
    p.e = e;
    p.M = e.M; p.N = e.N; p.K = K;
    p.tiles_m = (e.M + BM - 1) / BM;
    p.tiles_n = (e.N + BN - 1) / BN;
    const long nblk = (long)p.tiles_m * p.tiles_n;
    SF_CHECK_ARG(nblk < (1L << 31), "sf_gemm_nt: grid too large");
    if (c_dtype == SF_F32)
        SF_LAUNCH((gemm_nt_kernel<1>), dim3((unsigned)nblk), dim3(256), 2 * kStageBytes, stream, p);
    else
        SF_LAUNCH((gemm_nt_kernel<0>), dim3((unsigned)nblk), dim3(256), 2 * kStageBytes, stream, p);
    return sf_check_launch("sf_gemm_nt");
}

static int sf_gemm_check(long lda, long ldb, long ldc, long ldr, int M, int N, int K, int c_dtype, const void* R) {
    SF_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "sf_gemm_nt: negative shape");
    SF_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "sf_gemm_nt: K, lda, ldb must be multiples of 8 (16-byte rows)");
    SF_CHECK_ARG(ldc % 4 == 0 && ldr % 4 == 0, "sf_gemm_nt: ldc, ldr must be multiples of 4");
    SF_CHECK_ARG(c_dtype == SF_BF16 || c_dtype == SF_F32, "sf_gemm_nt: c_dtype");
    SF_CHECK_ARG(!(R && c_dtype == SF_F32), "sf_gemm_nt: residual epilogue is bf16-only");
    // Operands of ANY total size are fine (a 2.7 GB [N, 2I] activation, an 18 GB stash): the LDS-DMA buffer descriptor is rebuilt per
    // 256-row tile from a 64-bit base and only the offsets INSIDE a tile are 32-bit, range-checked against num_records = 2^31 - 1.
    // What must fit is therefore one tile's span: 256 rows of the leading dimension plus the K advance.
    SF_CHECK_ARG(256 * lda * 2 + (long)K * 2 < (1L << 31) && 256 * ldb * 2 + (long)K * 2 < (1L << 31),
                 "sf_gemm_nt: 256 rows of an operand (256 * ld * 2 bytes + K * 2) must span less than 2 GiB (buffer-descriptor offsets are 32-bit)");
    return 0;
}

int sf_gemm_nt_256w4_splitk_launch(const void* A, long lda, const void* B, long ldb, int K, const SfGemmEpi& e, int c_dtype, float* workspace,
                                   long workspace_floats, void* stream);

// sf_gemm_nt with a workspace (ABI 5): C = A . B^T (+ R after the rounding), alpha 1, beta 0.  Under-filled grids take the split-K form of the
// 4-wave kernel through `workspace` (>= 4 * roundup(M, 256) * N floats covers every split; less simply disables it); every other shape is sf_gemm_nt.
extern "C" int sf_gemm_nt_ws(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N, int K,
                             const void* R, long ldr, float* workspace, long workspace_floats, void* stream) {
    if (int st = sf_gemm_check(lda, ldb, ldc, ldr, M, N, K, c_dtype, R)) return st;
    if (M == 0 || N == 0) return 0;
    SfGemmEpi e;
    e.C = C; e.ldc = ldc; e.R = (const sf_bf16*)R; e.ldr = ldr;
    e.Cadd = nullptr; e.ldadd = 0; e.add_S = 1; e.add_Spad = 1; e.add_off = 0;
    e.M = M; e.N = N; e.alpha = 1.0f; e.beta = 0.0f;
    if (sf_gemm_use_256()) {
        const int st = sf_gemm_nt_256w4_splitk_launch(A, lda, B, ldb, K, e, c_dtype, workspace, workspace_floats, stream);
        if (st != -1) return st;
    }
    return sf_gemm_dispatch(A, lda, B, ldb, K, e, c_dtype, stream);
}

extern "C" int sf_gemm_nt(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M, int N,
                          int K, float alpha, float beta, const void* R, long ldr, void* stream) {
    if (int st = sf_gemm_check(lda, ldb, ldc, ldr, M, N, K, c_dtype, R)) return st;
    if (M == 0 || N == 0) return 0;
    SfGemmEpi e;
    e.C = C; e.ldc = ldc; e.R = (const sf_bf16*)R; e.ldr = ldr;
    e.Cadd = nullptr; e.ldadd = 0; e.add_S = 1; e.add_Spad = 1; e.add_off = 0;
    e.M = M; e.N = N; e.alpha = alpha; e.beta = beta;
    return sf_gemm_dispatch(A, lda, B, ldb, K, e, c_dtype, stream);
}

// d(act) = A . B^T fused with d(SwiGLU): dgu[:, :I] = d(gate), dgu[:, I:] = d(up) from gu [M, 2I] (see SfGemmEpi).  The fused
// epilogue exists in the 4-wave kernel for whole tiles only; every other shape runs the two steps through `dact`.
extern "C" int sf_gemm_nt_swiglu_bwd(const void* A, long lda, const void* B, long ldb, int M, int I, int K, const void* gu,
                                     long ldgu, void* dgu, long lddgu, void* dact, long lddact, void* stream) {
    if (int st = sf_gemm_check(lda, ldb, lddact, 0, M, I, K, SF_BF16, nullptr)) return st;
    SF_CHECK_ARG(gu && dgu && dact && ldgu % 8 == 0 && lddgu % 8 == 0 && lddact % 8 == 0 && I % 8 == 0,
                 "sf_gemm_nt_swiglu_bwd: gu / dgu / dact rows must be 16-byte aligned");
    if (M == 0 || I == 0) return 0;
    SfGemmEpi e;
    e.C = dact; e.ldc = lddact; e.R = nullptr; e.ldr = 0;
    e.Cadd = nullptr; e.ldadd = 0; e.add_S = 1; e.add_Spad = 1; e.add_off = 0;
    e.M = M; e.N = I; e.alpha = 1.f; e.beta = 0.f;
#ifdef SF_EMU
    const bool big = K >= 512;
#else
    const bool big = 2L * ((M + 255) / 256) * (I / 256) > sf_gemm_cus() && K >= 512;      // (more than half a tile per CU, as in sf_gemm_dispatch)
#endif
    static const int fuse = sf_knob("SF_GEMM_SWIGLU_FUSE", 1);
    const bool aligned = ((size_t)gu & 15) == 0 && ((size_t)dgu & 15) == 0 && ((size_t)dact & 15) == 0;
#ifdef SF_ABLATE
    if (fuse && aligned && (sf_knob("SF_GEMM_P2", 0) & 4)) {
        SfGemmEpi ef = e;
        ef.sw_gu = (const sf_bf16*)gu; ef.sw_ldgu = ldgu;
        ef.sw_dgu = (sf_bf16*)dgu; ef.sw_lddgu = lddgu;
        const int st = sf_gemm_nt_p2_launch(A, lda, B, ldb, K, ef, SF_BF16, stream);
        if (st != -1) return st;
    }
#endif
    if (fuse && big && aligned && M >= 256 && I % 256 == 0 && K % 64 == 0 && sf_gemm_use_256()) {
        // ragged M (real data: the collator pads a batch to its own longest sample): the whole 256-row tiles take the fused kernel,
        // the last M % 256 rows the two steps -- same bits either way, so the split is invisible
        // (M > 256, not whole tiles: ONE launch with the last row tile shifted up to end at row M -- GemmW4Args::mshift -- unless the
        //  caller writes the gradients over gate|up; M < 256 never gets here)
        const bool one = M % 256 == 0 || (M > 256 && (const void*)gu != (const void*)dgu && sf_knob("SF_GEMM_MSHIFT", 1));
        const int Mf = one ? M : M / 256 * 256, Mt = M - Mf;
        SfGemmEpi ef = e;
        ef.M = Mf;
        ef.sw_gu = (const sf_bf16*)gu; ef.sw_ldgu = ldgu;
        ef.sw_dgu = (sf_bf16*)dgu; ef.sw_lddgu = lddgu;
        if (int st = sf_gemm_nt_256w4_launch(A, lda, B, ldb, K, ef, SF_BF16, stream)) return st;
        if (Mt == 0) return 0;
        SfGemmEpi et = e;
        et.M = Mt;
        et.C = (sf_bf16*)dact + (long)Mf * lddact;
        if (int st = sf_gemm_dispatch((const sf_bf16*)A + (long)Mf * lda, lda, B, ldb, K, et, SF_BF16, stream)) return st;
        return sf_swiglu_bwd(et.C, SF_BF16, lddact, (const sf_bf16*)gu + (long)Mf * ldgu, ldgu, Mt, I, (sf_bf16*)dgu + (long)Mf * lddgu,
                             lddgu, stream);
    }
    if (int st = sf_gemm_dispatch(A, lda, B, ldb, K, e, SF_BF16, stream)) return st;
    return sf_swiglu_bwd(dact, SF_BF16, lddact, gu, ldgu, M, I, dgu, lddgu, stream);
}

// gu = A . Wgu^T (the fused gate|up projection, [M, 2I]) and act = round(silu(gate)) * up in ONE launch when the 4-wave kernel
// can take the shape (whole tiles; see ADD = 3 in sf_gemm256w4_kernel.h); every other shape: the GEMM, then sf_swiglu_fwd.
extern "C" int sf_gemm_nt_swiglu_fwd(const void* A, long lda, const void* Wgu, long ldw, int M, int I, int K, void* gu, long ldgu,
                                     void* act, long ldact, void* stream) {
    SF_CHECK_ARG(I >= 0 && I < (1 << 29), "sf_gemm_nt_swiglu_fwd: bad I");
    if (int st = sf_gemm_check(lda, ldw, ldgu, 0, M, 2 * I, K, SF_BF16, nullptr)) return st;
    SF_CHECK_ARG(gu && act && ldgu % 8 == 0 && ldact % 8 == 0 && I % 8 == 0, "sf_gemm_nt_swiglu_fwd: gu / act rows must be 16-byte aligned");
    if (M == 0 || I == 0) return 0;
    SfGemmEpi e;
    e.C = gu; e.ldc = ldgu; e.R = nullptr; e.ldr = 0;
    e.Cadd = nullptr; e.ldadd = 0; e.add_S = 1; e.add_Spad = 1; e.add_off = 0;
    e.M = M; e.N = 2 * I; e.alpha = 1.f; e.beta = 0.f;
#ifdef SF_EMU
    const bool big = K >= 512;
#else
    const bool big = 2L * ((M + 255) / 256) * (I / 128) > sf_gemm_cus() && K >= 512;
#endif
    static const int fuse = sf_knob("SF_GEMM_SWIGLU_FWD_FUSE", 1);
    const bool aligned = ((size_t)gu & 15) == 0 && ((size_t)act & 15) == 0;
    if (fuse && big && aligned && M >= 256 && I % 128 == 0 && K % 64 == 0 && (I + 256L) * ldw * 2 < (1L << 31) && sf_gemm_use_256()) {
        // ragged M: one launch with the last row tile shifted up to end at row M (GemmW4Args::mshift; the rows two tiles share get the
        // same bits twice); with the knob off: whole tiles fused, the tail rows in two steps (same bits)
        const bool one = M % 256 == 0 || (M > 256 && sf_knob("SF_GEMM_MSHIFT", 1));
        const int Mf = one ? M : M / 256 * 256, Mt = M - Mf;
        SfGemmEpi ef = e;
        ef.M = Mf;
        ef.sw_dgu = (sf_bf16*)act; ef.sw_lddgu = ldact;   // (sw_gu stays null: that is what selects the forward form)
        if (int st = sf_gemm_nt_256w4_launch(A, lda, Wgu, ldw, K, ef, SF_BF16, stream)) return st;
        if (Mt == 0) return 0;
        SfGemmEpi et = e;
        et.M = Mt;
        et.C = (sf_bf16*)gu + (long)Mf * ldgu;
        if (int st = sf_gemm_dispatch((const sf_bf16*)A + (long)Mf * lda, lda, Wgu, ldw, K, et, SF_BF16, stream)) return st;
        return sf_swiglu_fwd(et.C, SF_BF16, ldgu, Mt, I, (sf_bf16*)act + (long)Mf * ldact, ldact, stream);
    }
    if (int st = sf_gemm_dispatch(A, lda, Wgu, ldw, K, e, SF_BF16, stream)) return st;
    return sf_swiglu_fwd(gu, SF_BF16, ldgu, M, I, act, ldact, stream);
}

// Teacher head: z = A . Wp^T where Wp is the frozen head with its rows permuted draft-sub-vocabulary-first (see
// sf_teacher_reduce_perm).  When the chip-filling kernel takes the shape and `part` is given, only the first *vz_out =
// roundup(Vd, 256) columns are stored; every later 128-column block of a row leaves as one {max, sum exp, argmax column, 0} record
// in `part` (*nparts_out blocks, row-major: block q of row r at part[(r * part_stride + q) * 4]).  Otherwise all Vt columns are stored and *nparts_out = 0.  z must have room for Vt columns.
// 1 when sf_gemm_nt_teacher (given `part`) takes the reduced form for this shape -- then z only needs roundup(Vd, 256) columns
extern "C" int sf_gemm_nt_teacher_reduces(int M, int Vt, int K, int Vd) {
#ifdef SF_EMU
    const bool big = K >= 512;
#else
    const bool big = (long)((M + 255) / 256) * ((Vt + 255) / 256) >= 256 && K >= 512;
#endif
    static const int fuse = sf_knob("SF_GEMM_TEACHER_FUSE", 1);
    const int vz = (Vd + 255) / 256 * 256;
    return (fuse && big && Vd > 0 && vz < Vt && K % 64 == 0 && M >= 192 && sf_gemm_use_256()) ? 1 : 0;
}

extern "C" int sf_gemm_nt_teacher(const void* A, long lda, const void* Wp, long ldw, int M, int Vt, int K, int Vd, void* z, long ldz,
                                  float* part, long part_stride, int* vz_out, int* nparts_out, void* stream) {
    if (int st = sf_gemm_check(lda, ldw, ldz, 0, M, Vt, K, SF_BF16, nullptr)) return st;
    const bool will_reduce = part && sf_gemm_nt_teacher_reduces(M, Vt, K, Vd) && part_stride >= (Vt - (Vd + 255) / 256 * 256 + 127) / 128 &&
                             ((size_t)part & 15) == 0;
    SF_CHECK_ARG(Vd > 0 && Vd <= Vt && vz_out && nparts_out && (ldz >= Vt || (will_reduce && ldz >= (Vd + 255) / 256 * 256)),
                 "sf_gemm_nt_teacher: bad shape / missing outputs (z narrower than Vt columns needs the reduced form: sf_gemm_nt_teacher_reduces)");
    *vz_out = Vt;
    *nparts_out = 0;
    if (M == 0) return 0;
    SfGemmEpi e;
    e.C = z; e.ldc = ldz; e.R = nullptr; e.ldr = 0;
    e.Cadd = nullptr; e.ldadd = 0; e.add_S = 1; e.add_Spad = 1; e.add_off = 0;
    e.M = M; e.N = Vt; e.alpha = 1.f; e.beta = 0.f;
#ifdef SF_EMU
    const bool big = K >= 512;
#else
    const bool big = (long)((M + 255) / 256) * ((Vt + 255) / 256) >= 256 && K >= 512;
#endif
    const int vz = (Vd + 255) / 256 * 256;
    (void)big;
    if (will_reduce) {
        e.red_part = part; e.red_stride = part_stride; e.red_n0 = vz;
        *vz_out = vz;
        *nparts_out = (Vt - vz + 127) / 128;
        return sf_gemm_nt_256w4_launch(A, lda, Wp, ldw, K, e, SF_BF16, stream);
    }
    return sf_gemm_dispatch(A, lda, Wp, ldw, K, e, SF_BF16, stream);
}

extern "C" int sf_gemm_nt_rowadd(const void* A, long lda, const void* B, long ldb, void* C, int c_dtype, long ldc, int M,
                                 int N, int K, float alpha, const float* Cadd, long ldadd, int S, int Spad, int off,
                                 float* workspace, long workspace_floats, void* stream) {
    if (int st = sf_gemm_check(lda, ldb, ldc, 0, M, N, K, c_dtype, nullptr)) return st;
    SF_CHECK_ARG(Cadd && ldadd % 4 == 0 && S > 0 && Spad >= S && off >= 0 && off + S <= Spad && M % S == 0,
                 "sf_gemm_nt_rowadd: bad addend layout");
    if (M == 0 || N == 0) return 0;
    SfGemmEpi e;
    e.C = C; e.ldc = ldc; e.R = nullptr; e.ldr = 0;
    e.Cadd = Cadd; e.ldadd = ldadd; e.add_S = S; e.add_Spad = Spad; e.add_off = off;
    e.M = M; e.N = N; e.alpha = alpha; e.beta = 0.f;
    if (workspace && sf_gemm_use_256()) {       // under-filled grid (batch-1 recipes): split-K, the addend joins in the reduce (sf_gemm_nt_ws)
        const int st = sf_gemm_nt_256w4_splitk_launch(A, lda, B, ldb, K, e, c_dtype, workspace, workspace_floats, stream);
        if (st != -1) return st;
    }
    return sf_gemm_dispatch(A, lda, B, ldb, K, e, c_dtype, stream);
}
