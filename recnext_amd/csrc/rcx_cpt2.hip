// The 28x28 / level 3 instantiations of the channel-per-lane tiled kernel (rcx_cpt_kernel.h); see rcx_cpt.hip.
#include "rcx_cpt_kernel.h"

namespace rcx {
namespace cpt {

hipError_t launch_t2(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s, const SavedPyr& sv)
{
    return launch_md<2, 1>(x, y, wpack, bpack, N, C, mode, dtype, s, sv);
}

}  // namespace cpt
}  // namespace rcx
