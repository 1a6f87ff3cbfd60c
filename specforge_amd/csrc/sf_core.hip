// Library-level entry points of the C-ABI: version, build flavour, last-error string.
#include "sf_api_internal.h"

char* sf_error_buffer() {
    static thread_local char buf[SF_ERROR_BUFFER_BYTES] = {0};
    return buf;
}

extern "C" int sf_abi_version(void) { return SF_ABI_VERSION; }

extern "C" int sf_is_emulated(void) {
#ifdef SF_EMU
    return 1;
#else
    return 0;
#endif
}

extern "C" const char* sf_last_error(void) { return sf_error_buffer(); }

#if defined(SF_ABLATE) && !defined(SF_EMU)
// TOOLS BUILD ONLY (never in the product library): a stand-in for a communication kernel's CU footprint.  `n_wg`
// workgroups of 256 threads (a few registers, 8 KiB of LDS: the shape of an RCCL channel) each hold one CU slot for
// `usec` microseconds.  A GEMM workgroup that needs a whole CU (512 registers per SIMD lane) cannot share the CU with
// one of these, exactly as it cannot share it with an RCCL workgroup: tools/contention_bench.py uses it to measure
// what the weight-gradient phase loses when a collective holds 8-32 CUs.
namespace {
__global__ void __launch_bounds__(256) cu_hog_kernel(long long ticks, unsigned* sink) {
    __shared__ unsigned pad[2048];
    pad[threadIdx.x] = threadIdx.x;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && pad[(threadIdx.x * 7) & 2047] == 0xffffffffu) *sink = 1;
}
}  // namespace
extern "C" int sf_tool_cu_hog(int n_wg, long long usec, void* stream) {
    // wall_clock64 ticks at 100 MHz on gfx9
    hipLaunchKernelGGL(cu_hog_kernel, dim3((unsigned)n_wg), dim3(256), 0, (hipStream_t)stream, usec * 100, (unsigned*)nullptr);
    return sf_check_launch("sf_tool_cu_hog");
}
#endif
