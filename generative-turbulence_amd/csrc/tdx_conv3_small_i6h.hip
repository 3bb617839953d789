// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 6 M tiles per wave, fp16 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE_F16(6, conv3_small_go_6h)
