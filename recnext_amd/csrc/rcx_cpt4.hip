// The 16-pixel-tile instantiations of the channel-per-lane tiled kernel (rcx_cpt_kernel.h, TS = 16): the 64 x 64 / level 3 block -- stage 1 of a
// 512 x 512 input (detection/recnext.py:11-36, BASELINE config 5) and, in float32, the inner block of the 128 x 128 / level 4 split schedule.  Every
// plane of its ladder is even (64 -> 32 -> 16 -> 8): only the exact-2x resize pattern occurs.  (A translation unit of its own: parallel builds.)
#include "rcx_cpt_kernel.h"

namespace rcx {
namespace cpt {

hipError_t launch_ts16(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s)
{
    if (dtype == 1) return mode == 1 ? launch16<1, bf16_t>(x, y, wpack, bpack, N, C, s) : launch16<0, bf16_t>(x, y, wpack, bpack, N, C, s);
    if (dtype == 2) return mode == 1 ? launch16<1, f16_t>(x, y, wpack, bpack, N, C, s) : launch16<0, f16_t>(x, y, wpack, bpack, N, C, s);
    return mode == 1 ? launch16<1, float>(x, y, wpack, bpack, N, C, s) : launch16<0, float>(x, y, wpack, bpack, N, C, s);
}

}  // namespace cpt
}  // namespace rcx
