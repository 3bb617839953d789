// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 3 M tiles per wave, split-precision fp32 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE(3, true, conv3_small_go_3s)
