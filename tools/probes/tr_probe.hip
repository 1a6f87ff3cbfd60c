#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];   // X[row][col], 64x64, value = row*64+col
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x, i = l & 15, grp = l >> 4;
    // group g reads the 4x16 block with rows 4*g..4*g+3, cols 0..15; lane i supplies piece i: row 4g + i/4, cols 4*(i%4)
    short* p = &lds[(4 * grp + (i >> 2)) * 64 + 4 * (i & 3)];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    std::vector<short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        int want = (4 * (l >> 4) + j) * 64 + (l & 15);   // X[row 4g+j][col i]
        if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d got %d (row %d col %d) want %d\n", l, j, h[l*4+j], h[l*4+j]/64, h[l*4+j]%64, want); ++bad; }
    }
    printf("tr probe mismatches: %d\n", bad);
    return 0;
}
