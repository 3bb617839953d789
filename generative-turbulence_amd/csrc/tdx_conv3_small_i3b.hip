// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 3 M tiles per wave, bf16 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE(3, false, conv3_small_go_3b)
