// Translation unit 5 of 5 of the tiled backward kernels (rcx_cptbwd_kernels.h): one kernel family per file, compiled in parallel.
#define RCX_CPTBWD_PART 5
#include "rcx_cptbwd_kernels.h"
