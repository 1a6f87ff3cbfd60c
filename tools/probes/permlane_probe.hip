// what v_permlane16_swap / v_permlane32_swap (gfx950) return for (x, x): prints r[0], r[1] per lane for x = lane id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    auto r2 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    o[threadIdx.x * 4 + 0] = r[0]; o[threadIdx.x * 4 + 1] = r[1];
    o[threadIdx.x * 4 + 2] = r2[0]; o[threadIdx.x * 4 + 3] = r2[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: swap32 (%2u, %2u)  swap16 (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
