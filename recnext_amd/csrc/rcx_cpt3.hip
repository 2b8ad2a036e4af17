// The shorter-ladder instantiations of the channel-per-lane tiled kernel (rcx_cpt_kernel.h): 56x56 / level 3 and 28x28 / level 2 -- stages 1
// and 2 of a 448x448 input, and the inner blocks the nested schedule meets; see rcx_cpt.hip.  (A translation unit of its own: parallel builds.)
#include "rcx_cpt_kernel.h"

namespace rcx {
namespace cpt {

hipError_t launch_lv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int H, int mode, int dtype, hipStream_t s)
{
    const SavedPyr sv{};
    if (H == 56) return cb16(N, C) ? launch_md<4, 4, 3>(x, y, wpack, bpack, N, C, mode, dtype, s, sv) : launch_md<4, 2, 3>(x, y, wpack, bpack, N, C, mode, dtype, s, sv);
    // channel counts that are not multiples of 64: 32-channel workgroups (the alternative here is the LDS-pyramid kernel, not the banded one)
    if (C % 64 != 0) return launch_md<2, 2, 2>(x, y, wpack, bpack, N, C, mode, dtype, s, sv);
    return launch_md<2, 1, 2>(x, y, wpack, bpack, N, C, mode, dtype, s, sv);
}

}  // namespace cpt
}  // namespace rcx
