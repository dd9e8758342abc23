// ce_slice_dim64.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 64 floats
#define AE_SL_INSTANTIATE_DIM 64
#include "ce_slice_kernels.h"
