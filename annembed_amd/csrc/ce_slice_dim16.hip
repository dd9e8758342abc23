// ce_slice_dim16.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 16 floats
#define AE_SL_INSTANTIATE_DIM 16
#include "ce_slice_kernels.h"
