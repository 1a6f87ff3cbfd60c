// instantiations of the 4-wave NT GEMM (sf_gemm256w4_kernel.h): OUT_F32 = 1, ADD = 0, both operand orders
#include "sf_gemm256w4_kernel.h"

SF_W4_DEFINE(1, 0, 12)
SF_W4_DEFINE(1, 0, 13)
