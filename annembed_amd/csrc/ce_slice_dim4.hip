// ce_slice_dim4.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 4 floats
#define AE_SL_INSTANTIATE_DIM 4
#include "ce_slice_kernels.h"
