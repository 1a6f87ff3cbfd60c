// What the matrix pipe sustains under this part's power budget with NOTHING else in the loop: one wave per SIMD, 256 accumulator
// registers, operands resident in VGPRs, back-to-back MFMAs.  Four arms: {v_mfma_f32_16x16x32_bf16 (the GEMM's), v_mfma_f32_32x32x16_bf16
// (the attention's)} x {zero operands, random operands}.  Prints TFLOP/s per arm (hipEvents over ~0.3 s of launches after a warm-up) --
// the ceiling any bf16 GEMM on this box can be priced against besides the datasheet's 2.5 PFLOP/s (which assumes 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power_probe tools/probes/mfma_power_probe.hip && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 128 MFMAs 16x16x32 per trip: 8 A x 8 B fragments, two k-halves (the GEMM kernel's K-tile without its fillers)
__global__ void __launch_bounds__(256, 1) k16(const v8s* src, float* sink, int trips) {
    v8s f[2][16];
    for (int s = 0; s < 2; ++s)
        for (int g = 0; g < 16; ++g) f[s][g] = src[((s * 16 + g) * 256 + threadIdx.x) % 8192];
    v4f acc[8][8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(f[s][j]), "v"(f[s][8 + i]));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float r = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) r += acc[i][j][0] + acc[i][j][3];
    if (r == 12345.678f) sink[0] = r;
}

// 64 MFMAs 32x32x16 per trip (the same 128 x 128 x 64 of work): 4 A x 4 B fragments, four k-steps
__global__ void __launch_bounds__(256, 1) k32(const v8s* src, float* sink, int trips) {
    v8s f[4][8];
    for (int s = 0; s < 4; ++s)
        for (int g = 0; g < 8; ++g) f[s][g] = src[((s * 8 + g) * 256 + threadIdx.x) % 8192];
    v16f acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(f[s][j]), "v"(f[s][4 + i]));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float r = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][15];
    if (r == 12345.678f) sink[0] = r;
}

static unsigned short bf16(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int n = 8192 * 8;
    std::vector<unsigned short> h(n);
    unsigned short *dz, *dr;
    float* sink;
    hipMalloc(&dz, n * 2); hipMalloc(&dr, n * 2); hipMalloc(&sink, 4);
    hipMemset(dz, 0, n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) {   // ~N(0, 0.02): sums of 12 uniforms
        float s = 0.f;
        for (int k = 0; k < 12; ++k) s += (float)rand() / RAND_MAX;
        h[i] = bf16((s - 6.f) * 0.02f);
    }
    hipMemcpy(dr, h.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int trips = 20000;                       // x 128 x 16 cycles = 41 M cycles ~ 20 ms per launch
    const double flop_per_launch = (double)cus * 4 * trips * 128.0 * 128.0 * 64.0 * 2.0;
    for (int rep = 0; rep < 2; ++rep)
        for (int arm = 0; arm < 4; ++arm) {
            const v8s* src = (const v8s*)((arm & 1) ? dr : dz);
            auto go = [&]() {
                if (arm < 2) k16<<<cus, 256>>>(src, sink, trips);
                else k32<<<cus, 256>>>(src, sink, trips);
            };
            for (int w = 0; w < 8; ++w) go();      // warm-up: let the clock settle under this arm's load
            hipEventRecord(e0);
            const int L = 15;
            for (int w = 0; w < L; ++w) go();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double tf = flop_per_launch * L / (ms * 1e-3) / 1e12;
            printf("{\"probe\": \"mfma_power\", \"rep\": %d, \"mfma\": \"%s\", \"operands\": \"%s\", \"cus\": %d, \"tflops\": %.1f, \"eff_clock_ghz\": %.3f}\n", rep,
                   arm < 2 ? "16x16x32" : "32x32x16", (arm & 1) ? "random" : "zero", cus, tf, tf * 1e12 / (cus * 4 * 1024.0) / 1e9);
        }
    return 0;
}
