// edgeconv_f16b.hip -- the two-plane f16x2 EdgeConv kernel: edgeconv_f16.hip compiled with EF_V2 (see its header),
// entry point l3d_edgeconv_forward_f16b, parameters from the fifth packed copy (edgeconv_layout.h).
#define EF_V2 1
#include "edgeconv_f16.hip"
