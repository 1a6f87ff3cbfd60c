// instantiations of the 4-wave NT GEMM (sf_gemm256w4_kernel.h): teacher head, column tiles past the draft sub-vocabulary reduced in
// the epilogue instead of stored (ADD = 4)
#include "sf_gemm256w4_kernel.h"

SF_W4_DEFINE(0, 4, 12)
SF_W4_DEFINE(0, 4, 13)
