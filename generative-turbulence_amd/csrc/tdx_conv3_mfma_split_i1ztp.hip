// One instantiation of the split-precision MFMA conv kernel (tdx_conv3_mfma_split_kernel.h): NT = 1, zero padding, brick thins, permuted axes.
#include "tdx_conv3_mfma_split_kernel.h"
SPLIT_INSTANCE(1, true, BRICK_THIN, true, conv3_mfma_split_go_1ztp)
