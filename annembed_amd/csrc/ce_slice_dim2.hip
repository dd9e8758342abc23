// ce_slice_dim2.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 2 floats
#define AE_SL_INSTANTIATE_DIM 2
#include "ce_slice_kernels.h"
