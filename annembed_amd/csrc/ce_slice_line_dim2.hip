// ce_slice_line_dim2.hip -- the time-sliced mode's step kernel on node lines (ce_slice_kernels.h: LineRec) for rows of 2 floats
#define AE_SL_INSTANTIATE_LINE_DIM 2
#include "ce_slice_kernels.h"
