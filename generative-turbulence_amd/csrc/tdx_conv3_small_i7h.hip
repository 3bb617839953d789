// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 7 M tiles per wave, fp16 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE_F16(7, conv3_small_go_7h)
