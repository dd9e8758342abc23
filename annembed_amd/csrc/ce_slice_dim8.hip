// ce_slice_dim8.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 8 floats
#define AE_SL_INSTANTIATE_DIM 8
#include "ce_slice_kernels.h"
