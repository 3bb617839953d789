// One instantiation of the split-precision MFMA conv kernel (tdx_conv3_mfma_split_kernel.h): NT = 1, replicate padding, brick bigs, permuted axes.
#include "tdx_conv3_mfma_split_kernel.h"
SPLIT_INSTANCE(1, false, BRICK_BIG, true, conv3_mfma_split_go_1rbp)
