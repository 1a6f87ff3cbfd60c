// instantiations of the 4-wave NT GEMM (sf_gemm256w4_kernel.h): OUT_F32 = 1, ADD = 1, both operand orders
#include "sf_gemm256w4_kernel.h"

SF_W4_DEFINE(1, 1, 12)
SF_W4_DEFINE(1, 1, 13)
