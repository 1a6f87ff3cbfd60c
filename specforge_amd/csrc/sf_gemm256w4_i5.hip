// instantiations of the 4-wave NT GEMM (sf_gemm256w4_kernel.h): fused gate|up projection with SwiGLU forward in the epilogue (ADD = 3)
#include "sf_gemm256w4_kernel.h"

SF_W4_DEFINE(0, 3, 12)
SF_W4_DEFINE(0, 3, 13)
