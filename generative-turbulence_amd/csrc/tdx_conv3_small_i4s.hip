// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 4 M tiles per wave, split-precision fp32 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE(4, true, conv3_small_go_4s)
