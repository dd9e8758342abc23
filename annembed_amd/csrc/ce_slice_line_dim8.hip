// ce_slice_line_dim8.hip -- the time-sliced mode's step kernel on node lines (ce_slice_kernels.h: LineRec) for rows of 8 floats
#define AE_SL_INSTANTIATE_LINE_DIM 8
#include "ce_slice_kernels.h"
