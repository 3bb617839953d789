// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 4 M tiles per wave, bf16 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE(4, false, conv3_small_go_4b)
