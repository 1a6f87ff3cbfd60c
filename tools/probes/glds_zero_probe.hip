// Probe: does global_load_lds (LDS-DMA) from a __device__ const array deliver the data?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ const unsigned short zero_const[8] = {0,0,0,0,0,0,0,0};
__device__ __attribute__((aligned(64))) unsigned short zero_glob[32];
__global__ void k(const unsigned short* real, const unsigned short* zmalloc, int mode, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 8];
    int lane = threadIdx.x;
    for (int i = 0; i < 8; ++i) lds[lane * 8 + i] = 0xBEEF;   // poison
    __syncthreads();
    const unsigned short* src;
    bool oob = (lane & 1);
    if (mode == 0) src = oob ? zero_const : real + lane * 8;
    else if (mode == 1) src = oob ? zmalloc : real + lane * 8;
    else if (mode == 2) src = oob ? zero_glob : real + lane * 8;
    else src = real + lane * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + lane * 8), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned bad = 0;
    for (int i = 0; i < 8; ++i) {
        unsigned short v = lds[lane * 8 + i];
        unsigned short want = (oob && mode < 3) ? 0 : (unsigned short)(lane * 8 + i + 1);
        if (v != want) bad |= (1u << i);
    }
    out[blockIdx.x * 64 + lane] = bad | ((unsigned)lds[lane * 8] << 16);
}
int main(int argc, char** argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1;
    unsigned short h[512];
    for (int i = 0; i < 512; ++i) h[i] = i + 1;
    unsigned short *real, *zm; unsigned* out;
    hipMalloc(&real, sizeof(h)); hipMemcpy(real, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&zm, 256); hipMemset(zm, 0, 256);
    hipMalloc(&out, 64 * 64 * 4);
    for (int mode = 0; mode < 4; ++mode) {
        if (only >= 0 && mode != only) continue;
        hipMemset(out, 0xff, 64 * 64 * 4);
        k<<<64, 64>>>(real, zm, mode, out);
        std::vector<unsigned> ho(64 * 64);
        hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
        int nbad = 0; unsigned first = 0; int firsti = -1;
        for (size_t i = 0; i < ho.size(); ++i) if (ho[i] & 0xffff) { if (!nbad) { first = ho[i]; firsti = (int)i; } ++nbad; }
        printf("mode %d: bad lanes %d / %zu  first idx %d val %08x  err=%s\n", mode, nbad, ho.size(), firsti, first, hipGetErrorString(hipGetLastError())); fflush(stdout);
    }
    return 0;
}
