// instantiations of the 4-wave NT GEMM (sf_gemm256w4_kernel.h): bf16 output with d(SwiGLU) in the epilogue (ADD = 2)
#include "sf_gemm256w4_kernel.h"

SF_W4_DEFINE(0, 2, 12)
SF_W4_DEFINE(0, 2, 13)
