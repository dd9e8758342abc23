// ce_slice_dim3.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 3 floats
#define AE_SL_INSTANTIATE_DIM 3
#include "ce_slice_kernels.h"
