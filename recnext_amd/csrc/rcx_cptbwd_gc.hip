// Translation unit 2 of 4 of the tiled backward kernels (rcx_cptbwd_kernels.h): one kernel family per file, compiled in parallel.
#define RCX_CPTBWD_PART 2
#include "rcx_cptbwd_kernels.h"
