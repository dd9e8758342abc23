// ce_slice_dim32.hip -- the time-sliced mode's kernels (ce_slice_kernels.h) for rows of 32 floats
#define AE_SL_INSTANTIATE_DIM 32
#include "ce_slice_kernels.h"
