// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 7 M tiles per wave, split-precision fp32 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE(7, true, conv3_small_go_7s)
