// Platform layer of the specforge_amd kernels.
//
// Product build (hipcc --offload-arch=gfx950): thin inline wrappers over the gfx950
// builtins -- MFMA, LDS-DMA (global_load_lds), wave shuffles, s_barrier.
// Test build (-DSF_EMU, host compiler): the same names map onto tests/emu/sf_emu.h,
// a fiber-based SIMT interpreter used to debug kernel index logic on a machine
// without a GPU.  There is no third path and nothing here selects between them at
// run time.
#pragma once
#include <stdint.h>

typedef short sf_v8s __attribute__((ext_vector_type(8)));   // 8 x bf16 bits
typedef short sf_v4s __attribute__((ext_vector_type(4)));   // 4 x bf16 bits
typedef float sf_v4f __attribute__((ext_vector_type(4)));
typedef unsigned int sf_v2u __attribute__((ext_vector_type(2)));   // 4 x bf16 bits as two packed dwords
typedef int sf_v4i __attribute__((ext_vector_type(4)));
typedef float sf_v16f __attribute__((ext_vector_type(16)));
typedef unsigned short sf_bf16;  // raw bf16 bits

#ifdef SF_EMU
// ------------------------------------------------------------------ emulator
#include "sf_emu.h"
#define SF_GLOBAL
#define SF_DEVICE inline
#define SF_HD inline
#define SF_SHARED static thread_local
#define SF_LAUNCH_BOUNDS(t, w)
using sfemu::dim3;
#define threadIdx (sfemu::cur_tid())
#define blockIdx (sfemu::cur_bid())
#define blockDim (sfemu::cur_bdim())
#define gridDim (sfemu::cur_gdim())
typedef void* sfStream_t;
#define SF_LAUNCH(kernel, grid, block, smem, stream, ...) \
    sfemu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define SF_DYN_SMEM(name) char* name = sfemu::dyn_smem_base()
static inline const char* sf_launch_error() { return nullptr; }

SF_DEVICE void sf_syncthreads() { sfemu::block_barrier(); }
SF_DEVICE int sf_lane() { return sfemu::lane_id(); }
template <typename T> SF_DEVICE T sf_shfl_xor(T v, int m) { return sfemu::shfl_xor(v, m); }
template <typename T> SF_DEVICE T sf_shfl(T v, int l) { return sfemu::shfl(v, l); }
SF_DEVICE sf_v4f sf_mfma16(sf_v8s a, sf_v8s b, sf_v4f c) { return sfemu::mfma_16x16x32_bf16(a, b, c); }
SF_DEVICE sf_v16f sf_mfma32(sf_v8s a, sf_v8s b, sf_v16f c) { return sfemu::mfma_32x32x16_bf16(a, b, c); }
SF_DEVICE void sf_mfma16_acc(sf_v8s a, sf_v8s b, sf_v4f& c) { c = sfemu::mfma_16x16x32_bf16(a, b, c); }
SF_DEVICE void sf_mfma32_acc(sf_v8s a, sf_v8s b, sf_v16f& c) { c = sfemu::mfma_32x32x16_bf16(a, b, c); }
SF_DEVICE void sf_acc_touch(sf_v16f&) {}
SF_DEVICE void sf_valu_to_mfma(sf_v8s&) {}
SF_DEVICE void sf_mfma_drain() {}
SF_DEVICE void sf_acc_touch(sf_v4f&) {}
// a wave's lanes run in lockstep on the GPU; the interpreter's lanes are fibres and need an explicit rendezvous between
// an LDS write and a read of another lane's data within the wave
SF_DEVICE void sf_wave_lockstep() { sfemu::wave_sync(); }
SF_DEVICE void sf_glds16(const void* g, void* l) { sfemu::global_load_lds16(g, l); }
SF_DEVICE void sf_glds16_opaque(const void* g, void* l) { sfemu::global_load_lds16(g, l); }
SF_DEVICE void sf_flag_arrive(unsigned* c) { if (sfemu::lane_id() == 0) sfemu::atomic_add(c, 1u); }
SF_DEVICE unsigned sf_flag_peek(const unsigned* c) { return *(volatile const unsigned*)c; }
SF_DEVICE void sf_flag_wait(const unsigned* c, unsigned target) {
    long spins = 0;
    while (*(volatile const unsigned*)c < target) {
        sfemu::yield();
        if (++spins > 50000000L) { fprintf(stderr, "[sfemu] sf_flag_wait: no progress\n"); abort(); }
    }
}
struct SfBuf { const char* base; unsigned bytes; };
SF_DEVICE SfBuf sf_make_buf(const void* base, unsigned bytes) { return SfBuf{(const char*)base, bytes}; }
SF_DEVICE void sf_buf_glds16(SfBuf b, unsigned voff, unsigned soff, void* l) {
    static const char zero16[16] = {0};
    sfemu::global_load_lds16(voff + 16 <= b.bytes ? b.base + voff + soff : zero16, l);
}
// raw descriptor for the asm (compiler-opaque) buffer LDS-DMA: base may move by a scalar add per K-tile
struct SfBufRaw { const char* base; };
SF_DEVICE SfBufRaw sf_make_buf_raw(const void* base) { return SfBufRaw{(const char*)base}; }
SF_DEVICE void sf_buf_glds16_opaque(SfBufRaw b, unsigned voff, unsigned soff, void* l) {
    sfemu::global_load_lds16(b.base + voff + soff, l);
}
// bounded raw buffer for LDS-DMA (zeros past `bytes`), 16 B / 4 B per lane
struct SfBufB { const char* base; unsigned bytes; };
SF_DEVICE SfBufB sf_make_bufb(const void* base, unsigned bytes) { return SfBufB{(const char*)base, bytes}; }
SF_DEVICE void sf_bufb_empty(SfBufB& b) { b.bytes = 0; }
SF_DEVICE SfBufB sf_bufb_if(SfBufB b, bool on) { if (!on) b.bytes = 0; return b; }
SF_DEVICE void sf_bufb_glds16(SfBufB b, unsigned voff, void* l) {
    static const char zero16[16] = {0};
    sfemu::global_load_lds16(voff + 16 <= b.bytes ? b.base + voff : zero16, l);
}
SF_DEVICE void sf_bufb_glds4(SfBufB b, unsigned voff, void* l) {
    static const char zero4[4] = {0};
    sfemu::global_load_lds4(voff + 4 <= b.bytes ? b.base + voff : zero4, l);
}
template <typename T> SF_DEVICE void sf_pin(T&) {}
SF_DEVICE float sf_pair_max(float v) { return fmaxf(v, sfemu::shfl_xor(v, 32)); }
SF_DEVICE float sf_pair_sum(float v) { return v + sfemu::shfl_xor(v, 32); }
SF_DEVICE int sf_wave_id() { return sfemu::wave_index(); }
template <int MASK> SF_DEVICE void sf_xor_pair(float x, float& a, float& b) {
    const float o = sfemu::shfl_xor(x, MASK);
    const bool up = (sfemu::lane_id() & MASK) != 0;
    a = up ? o : x;
    b = up ? x : o;
}
template <int MASK> SF_DEVICE void sf_xor_pair(int x, int& a, int& b) {
    const int o = sfemu::shfl_xor(x, MASK);
    const bool up = (sfemu::lane_id() & MASK) != 0;
    a = up ? o : x;
    b = up ? x : o;
}
SF_DEVICE sf_v4s sf_ds_read_tr16(const void* l) { return sfemu::ds_read_tr16_b64(l); }
// v_permlane32_swap_b32 a, b: lanes 32..63 of a exchange with lanes 0..31 of b (the other two halves stay)
SF_DEVICE void sf_swap_halves(unsigned& a, unsigned& b) {
    const int l = sfemu::lane_id();
    const unsigned from_a = sfemu::shfl(a, l ^ 32), from_b = sfemu::shfl(b, l ^ 32);
    if (l < 32) b = from_a; else a = from_b;
}
// v_permlane16_swap_b32 a, b: the odd 16-lane rows of a (lanes 16..31, 48..63) exchange with the even rows of b (lanes 0..15, 32..47)
SF_DEVICE void sf_swap_rows16(unsigned& a, unsigned& b) {
    const int l = sfemu::lane_id();
    const unsigned from_a = sfemu::shfl(a, l ^ 16), from_b = sfemu::shfl(b, l ^ 16);
    if (l & 16) a = from_b; else b = from_a;
}
SF_DEVICE bool sf_all(bool pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= sfemu::shfl_xor(v, m);
    return v != 0;
}
SF_DEVICE void sf_wait_vm0() {}
template <int N> SF_DEVICE void sf_wait_vmcnt() {}
SF_DEVICE void sf_setprio_hi() {}
SF_DEVICE void sf_setprio_lo() {}
template <typename T> SF_DEVICE T sf_atomic_add(T* p, T v) { return sfemu::atomic_add(p, v); }
SF_DEVICE float sf_exp(float x) { return expf(x); }
SF_DEVICE float sf_exp2(float x) { return exp2f(x); }
SF_DEVICE float sf_exp2_raw(float x) { return exp2f(x); }
SF_DEVICE float sf_exp_fast(float x) { return exp2f(x * 1.4426950408889634f); }
SF_DEVICE float sf_rcp_fast(float x) { return 1.0f / x; }
SF_DEVICE float sf_log(float x) { return logf(x); }
SF_DEVICE float sf_rsqrt(float x) { return 1.0f / sqrtf(x); }

#else
// -------------------------------------------------------------------- gfx950
#include <hip/hip_runtime.h>
#define SF_GLOBAL __global__
#define SF_DEVICE __device__ __forceinline__
#define SF_HD __host__ __device__ __forceinline__
#define SF_SHARED __shared__
#define SF_LAUNCH_BOUNDS(t, w) __launch_bounds__(t, w)
typedef hipStream_t sfStream_t;
#define SF_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define SF_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
static inline const char* sf_launch_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

SF_DEVICE void sf_syncthreads() { __syncthreads(); }
SF_DEVICE int sf_lane() { return (int)(threadIdx.x & 63u); }
template <typename T> SF_DEVICE T sf_shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }
template <typename T> SF_DEVICE T sf_shfl(T v, int l) { return __shfl(v, l, 64); }
SF_DEVICE sf_v4f sf_mfma16(sf_v8s a, sf_v8s b, sf_v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// In-place accumulate with the accumulator pinned to AGPRs (tied asm operand).  With 256 accumulator registers per
// lane the compiler's own allocation of the builtin form is fragile: under pressure it renames MFMA destinations and
// parks accumulator tiles in VGPRs (v_accvgpr_write + s_nop in front of every MFMA).  The asm form cannot be renamed.
// The hazard recogniser does not see an MFMA inside asm: call sf_mfma_drain() before the accumulators are read.
SF_DEVICE void sf_mfma16_acc(sf_v8s a, sf_v8s b, sf_v4f& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// Fenced on both sides for the instruction scheduler: an accumulator read (v_accvgpr_read) is a register-only
// instruction that a "memory" clobber does not hold back -- the compiler hoisted such reads above the nops (seen with a
// short epilogue: wrong, run-to-run different results).
SF_DEVICE void sf_mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// "This accumulator is (re)defined here, in an AGPR": placed after sf_mfma_drain() it pins every later use of the value
// behind the drain.  Without it the register allocator may copy an asm MFMA's result to a VGPR right after the MFMA that
// produced it (a v_accvgpr_read one instruction later reads the old value: the hazard recogniser cannot see the asm).
SF_DEVICE void sf_acc_touch(sf_v4f& c) { asm volatile("" : "+a"(c)); }
// The 32x32x16 form of sf_mfma16_acc: accumulator tied to 16 AGPRs.  For accumulators that only MFMAs touch until the
// epilogue (attention dK^T / dV^T): left to the register allocator they compete with VALU-visible values for the 256
// architectural VGPRs and get shuttled through v_accvgpr_read / _write around every use.
SF_DEVICE void sf_mfma32_acc(sf_v8s a, sf_v8s b, sf_v16f& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
SF_DEVICE void sf_acc_touch(sf_v16f& c) { asm volatile("" : "+a"(c)); }
// A fragment a VALU instruction has just written (packed P / dS) is read by an MFMA inside asm, which the hazard recogniser
// does not see: the required wait states between the two go here, tied to the value so nothing moves across
SF_DEVICE void sf_valu_to_mfma(sf_v8s& f) { asm volatile("s_nop 1" : "+v"(f)); }
SF_DEVICE void sf_wave_lockstep() {}   // lanes of a wave execute LDS instructions in order, in lockstep
SF_DEVICE sf_v16f sf_mfma32(sf_v8s a, sf_v8s b, sf_v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// LDS-DMA, 16 B per lane.  `l` MUST be the same (wave-uniform) LDS address in every lane: the
// hardware writes lane i's 16 bytes at M0 + 16*i with M0 = readfirstlane(l), so a per-lane
// pointer goes wrong as soon as the compiler issues the load under a partial EXEC mask
// (first ACTIVE lane != lane 0) -- observed on gfx950 when a select between two source
// pointers was lowered to two exec-masked loads.
SF_DEVICE int sf_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// (a, b) = (x of the lane with bit MASK clear, x of the lane with it set) of the lane pair {l, l ^ MASK}, MASK = 16 or 32: the SAME
// ordered pair on both lanes.  gfx950's v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / wave halves between two
// registers in the VALU -- no LDS crossbar round trip (ds_bpermute + lgkmcnt wait: ~100 cycles each in a one-wave-per-SIMD epilogue).
template <int MASK> SF_DEVICE void sf_xor_pair(float x, float& a, float& b) {
    static_assert(MASK == 16 || MASK == 32, "row / half exchanges only");
    // inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc (ROCm 7.2) treats the two float results of the builtin as equal
    // when both operands are one value and folds every comparison between them (reproduced in a ten-line kernel: the swaps stay in
    // the ISA, their second result is dead).  s_nop 1: the VALU-write -> permlane-swap hazard the compiler would have padded.
    a = x;
    b = x;
    if constexpr (MASK == 32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int MASK> SF_DEVICE void sf_xor_pair(int x, int& a, int& b) {
    float fa, fb;
    sf_xor_pair<MASK>(__builtin_bit_cast(float, x), fa, fb);
    a = __builtin_bit_cast(int, fa); b = __builtin_bit_cast(int, fb);
}
SF_DEVICE bool sf_all(bool pred) { return __all(pred ? 1 : 0) != 0; }
// v_permlane32_swap_b32 a, b: lanes 32..63 of a exchange with lanes 0..31 of b (the other two halves stay).  s_nop 1: the
// VALU-write -> permlane-read hazard (both operands usually come straight out of a v_cvt_pk)
SF_DEVICE void sf_swap_halves(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// v_permlane16_swap_b32 a, b: the odd 16-lane rows of a exchange with the even rows of b (same hazard)
SF_DEVICE void sf_swap_rows16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// ds_read_b64_tr_b16: per 16-lane group, lane i passes the address of 8-byte piece i of a 4x16 bf16 block
// (piece i = row i/4, columns 4*(i%4)..+3; any row stride) and receives column i (rows 0..3).  Verified on
// MI355X by tools/probes/tr_probe.hip.
SF_DEVICE sf_v4s sf_ds_read_tr16(const void* l) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sf_v4s*)l);
}
SF_DEVICE void sf_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// LDS-DMA through a raw buffer descriptor (buffer_load_dwordx4 ... lds): the address is
// SGPR descriptor base + 32-bit per-lane voffset + SGPR soffset, so a lane carries one dword of address instead of a
// 64-bit pointer and a K-loop advances with scalar adds only.  The descriptor must be wave-uniform.  gfx9 raw buffers
// range-check voffset (not soffset) against `bytes` and return zeros past it.
struct SfBuf { __amdgpu_buffer_rsrc_t r; };
SF_DEVICE SfBuf sf_make_buf(const void* base, unsigned bytes) {
    SfBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
    return b;
}
SF_DEVICE void sf_buf_glds16(SfBuf b, unsigned voff, unsigned soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)l, 16, (int)voff, (int)soff, 0, 0);
}
// The same LDS-DMA as inline asm.  The compiler orders every LDS read it cannot disambiguate (ds_read_b64_tr_b16 has no
// memory operand) behind outstanding LDS-DMA *builtins* with an automatic `s_waitcnt vmcnt(0)`, which serialises a
// software-pipelined loop.  Hidden in asm, the DMA is invisible to that logic: the kernel's own counted vmcnt +
// barrier are then the only ordering, exactly as the hardware requires.  `l` must be wave-uniform.
SF_DEVICE void sf_glds16_opaque(const void* g, void* l) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
// Buffer-descriptor LDS-DMA as inline asm (both properties at once: scalar K advance AND invisible to the compiler's
// LDS ordering).  The descriptor is assembled by hand: {base[31:0], base[47:32] (stride 0), num_records = 2^31-1,
// DST_SEL/format word 0x00020000}; every field must be wave-uniform.
struct SfBufRaw { sf_v4i w; };
SF_DEVICE SfBufRaw sf_make_buf_raw(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    SfBufRaw b;
    b.w = sf_v4i{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
    return b;
}
SF_DEVICE void sf_buf_glds16_opaque(SfBufRaw b, unsigned voff, unsigned soff, void* l) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 : : "v"(voff), "s"(b.w), "s"(soff), "s"(lds) : "memory", "m0");
}
// The two halves of that DMA as separate program points, the way hand-scheduled kernels issue it: M0 (the LDS
// destination) is written in an EARLIER instruction slot, so the DMA itself is a single instruction with no `s_nop`
// for the M0 write -> LDS-DMA hazard.  Between sf_m0_set and sf_buf_glds16_m0 the compiler must have no reason to touch
// M0 (no LDS-DMA builtins, no indirect register indexing): the plan-scheduled GEMM loop satisfies that.
SF_DEVICE void sf_m0_set(void* l) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
    asm volatile("s_mov_b32 m0, %0" : : "s"(lds) : "m0");
}
SF_DEVICE void sf_buf_glds16_m0(SfBufRaw b, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(b.w), "s"(soff) : "memory");
}
// Bounded raw-buffer LDS-DMA from inline asm: {base, stride 0, num_records = bytes, raw format}.  The hardware range-checks
// the per-lane voffset against num_records and delivers zeros past it, so a ragged last tile needs neither a source
// select nor an exec-masked branch (the pointer form compiled to ~10 instructions per 1-KiB piece); the DMA is
// invisible to the compiler's vmcnt bookkeeping -- the kernel's own counted waits are the only ordering.
// Every descriptor field must be wave-uniform.  16 B per lane (lane i lands at M0 + 16 i) / 4 B per lane (M0 + 4 i).
struct SfBufB { sf_v4i w; };
SF_DEVICE SfBufB sf_make_bufb(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    SfBufB b;
    b.w = sf_v4i{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    return b;
}
SF_DEVICE void sf_bufb_empty(SfBufB& b) { b.w[2] = 0; }   // num_records = 0: every access is out of range, i.e. reads zeros
// `b` when the (wave-uniform) condition holds, else an empty buffer; the patched word is forced back into an SGPR (a select of
// whole descriptors can end up as a per-lane select, and the DMA's descriptor operand must be scalar)
SF_DEVICE SfBufB sf_bufb_if(SfBufB b, bool on) { b.w[2] = __builtin_amdgcn_readfirstlane(on ? b.w[2] : 0); return b; }
SF_DEVICE void sf_bufb_glds16(SfBufB b, unsigned voff, void* l) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 : : "v"(voff), "s"(b.w), "s"(lds) : "memory", "m0");
}
SF_DEVICE void sf_bufb_glds4(SfBufB b, unsigned voff, void* l) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds"
                 : : "v"(voff), "s"(b.w), "s"(lds) : "memory", "m0");
}
// "this value is complete here": an empty asm that reads and re-defines the register(s).  The compiler has to place the
// wait for whatever load produces the value BEFORE it.  Used on fragments loaded ahead of a software-pipelined loop:
// left alone, the waitcnt pass puts a vmcnt(N) at the first use INSIDE the loop, and since vmcnt is one in-order counter
// that wait also drains the LDS-DMA prefetch of the next tile (which the compiler cannot see) on every iteration.
template <typename T> SF_DEVICE void sf_pin(T& v) { asm volatile("" : "+v"(v)); }
// all-reduce over the lane pair (l, l ^ 32): v_permlane32_swap exchanges the upper half of one register with the lower
// half of the other, so {a, b} = swap(v, v) holds {v[l], v[l ^ 32]} in some order in every lane -- no select, no LDS
SF_DEVICE float sf_pair_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
SF_DEVICE float sf_pair_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// Workgroup-level arrive / wait on a monotonic LDS counter: a barrier whose "arrive" and "wait" halves are separate
// program points (gfx950 has no split s_barrier).  arrive = release (everything this wave did to LDS is complete),
// wait = acquire.  One lane per wave arrives.
SF_DEVICE void sf_flag_arrive(unsigned* c) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
SF_DEVICE unsigned sf_flag_peek(const unsigned* c) { return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SF_DEVICE void sf_flag_wait(const unsigned* c, unsigned target) {
    while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
SF_DEVICE void sf_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// counted wait: at most N of this wave's VMEM operations (LDS-DMA pieces included) still outstanding
template <int N> SF_DEVICE void sf_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "i"(N) : "memory"); }
SF_DEVICE void sf_setprio_hi() { __builtin_amdgcn_s_setprio(1); }
SF_DEVICE void sf_setprio_lo() { __builtin_amdgcn_s_setprio(0); }
template <typename T> SF_DEVICE T sf_atomic_add(T* p, T v) { return atomicAdd(p, v); }
SF_DEVICE float sf_exp(float x) { return expf(x); }
SF_DEVICE float sf_exp2(float x) { return exp2f(x); }
// bare v_exp_f32 (no denormal-range fix-up): for softmax terms, where tiny results may flush to 0
SF_DEVICE float sf_exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
// exp(x) for the streamed softmax terms (x <= 0): one multiply + v_exp_f32, relative error ~|x| * 2^-24; libm's expf costs
// ~10 VALU instructions per element and made the 128 k-column teacher rows VALU-bound
SF_DEVICE float sf_exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// 1 / x as one v_rcp_f32 (1 ulp): for the sigmoid of the SwiGLU epilogues, whose result is rounded to bf16 right after; the IEEE
// division is ~8 VALU instructions per element and the fused GEMM epilogues run them with the matrix pipe idle
SF_DEVICE float sf_rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
SF_DEVICE float sf_log(float x) { return logf(x); }
SF_DEVICE float sf_rsqrt(float x) { return rsqrtf(x); }
#endif

// ------------------------------------------------------------- bf16 helpers
SF_HD float sf_bf2f(sf_bf16 h) {
    union { uint32_t u; float f; } x;
    x.u = (uint32_t)h << 16;
    return x.f;
}
// round-to-nearest-even, NaN stays NaN (same as torch's float->bfloat16)
// (native __bf16 cast: lowers to v_cvt_pk_bf16_f32 on gfx950 -- one VALU op per two values instead of
// ~9 for the bit-twiddled form; the host compiler's soft conversion is bit-identical for non-NaN inputs)
SF_HD sf_bf16 sf_f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(sf_bf16, b);
}
SF_HD float sf_round_bf(float f) { return sf_bf2f(sf_f2bf(f)); }
// c + a.lo * b.lo + a.hi * b.hi for two bf16 pairs packed in 32 bits each (gfx950: v_dot2c_f32_bf16, one VALU instruction for what is two
// conversions per operand and two fmas otherwise).  Interpreter: the same sum in fp32.
SF_DEVICE float sf_dot2_bf16(unsigned a, unsigned b, float c) {
#ifdef SF_EMU
    const float a0 = sf_bf2f((sf_bf16)(a & 0xffffu)), a1 = sf_bf2f((sf_bf16)(a >> 16));
    const float b0 = sf_bf2f((sf_bf16)(b & 0xffffu)), b1 = sf_bf2f((sf_bf16)(b >> 16));
    return c + a0 * b0 + a1 * b1;
#else
    typedef __bf16 sf_bf2_ __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sf_bf2_, a), __builtin_bit_cast(sf_bf2_, b), c, false);
#endif
}
// two floats -> one dword of two bf16 (a in the low half): ONE v_cvt_pk_bf16_f32.  Four shorts assembled from scalar sf_f2bf casts
// compile to three conversions + v_perm_b32 + v_alignbit_b32 per four values (hipcc pairs the middle two): 11 instead of 3
// instructions per 8-byte store of a GEMM epilogue.
#ifdef SF_EMU
SF_HD uint32_t sf_pack2_bf16(float a, float b) { return (uint32_t)sf_f2bf(a) | ((uint32_t)sf_f2bf(b) << 16); }
#else
typedef float sf_v2f_ __attribute__((ext_vector_type(2)));
typedef __bf16 sf_v2bf_ __attribute__((ext_vector_type(2)));
SF_HD uint32_t sf_pack2_bf16(float a, float b) {
    sf_v2f_ v = {a, b};
    sf_v2bf_ r = __builtin_convertvector(v, sf_v2bf_);
    return __builtin_bit_cast(uint32_t, r);
}
#endif

// element type traits for kernels templated on bf16 / fp32 storage
template <typename T> struct SfElem;
template <> struct SfElem<sf_bf16> {
    static SF_HD float ld(const sf_bf16* p) { return sf_bf2f(*p); }
    static SF_HD void st(sf_bf16* p, float v) { *p = sf_f2bf(v); }
    static SF_HD float rnd(float v) { return sf_round_bf(v); }
};
template <> struct SfElem<float> {
    static SF_HD float ld(const float* p) { return *p; }
    static SF_HD void st(float* p, float v) { *p = v; }
    static SF_HD float rnd(float v) { return v; }
};

// all-reduce sum inside aligned groups of LANES (8 or 16) lanes.  Product build: DPP (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror) -- plain VALU ops, no LDS crossbar; the emulator's xor butterfly adds the same pairs
// in the same order, so both are bit-identical.
#ifdef SF_EMU
template <int LANES> SF_DEVICE float sf_row_sum(float v) {
    for (int m = 1; m < LANES; m <<= 1) v += sf_shfl_xor(v, m);
    return v;
}
#else
template <int CTRL> SF_DEVICE float sf_dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int LANES> SF_DEVICE float sf_row_sum(float v) {
    v += sf_dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += sf_dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += sf_dpp_f32<0x141>(v);   // row_half_mirror
    if (LANES >= 16) v += sf_dpp_f32<0x140>(v);  // row_mirror
    if (LANES == 32) v += sf_shfl_xor(v, 16);     // the other 16-lane DPP row of the group (head_dim 256: 32 lanes per token row)
    return v;
}
#endif

// block-wide reductions over 256-thread (or any multiple-of-64) workgroups
SF_DEVICE float sf_wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += sf_shfl_xor(v, m);
    return v;
}
SF_DEVICE float sf_wave_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, sf_shfl_xor(v, m));
    return v;
}
