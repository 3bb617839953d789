// One instantiation of the split-precision MFMA conv kernel (tdx_conv3_mfma_split_kernel.h): NT = 1, zero padding, brick mains, permuted axes.
#include "tdx_conv3_mfma_split_kernel.h"
SPLIT_INSTANCE(1, true, BRICK_MAIN, true, conv3_mfma_split_go_1zmp)
