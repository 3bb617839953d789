// One instantiation of the small-grid conv kernel (tdx_conv3_small_kernel.h): 4 M tiles per wave, fp16 tensors.
#include "tdx_conv3_small_kernel.h"
SMALL_INSTANCE_F16(4, conv3_small_go_4h)
