// Register-resident RecConv2d for the 16 * 2^k planes (16x16/L1, 32x32/L2, 64x64/L3 -- the 256^2 and 512^2 inputs): same
// kernels as rcx_lanes.hip with all 16 lanes of a group active; the DPP row boundary supplies the horizontal zero padding.
#include "rcx_lanes_kernels.h"

namespace rcx {
namespace lanes {

template <int W0, int LEVEL, int MODE, typename TIO>
static hipError_t launch16_whole(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    if (p.waves == 8) return launch_w<W0, LEVEL, 16, MODE, 8, TIO>(x, y, wpack, bpack, p, s);
    return launch_w<W0, LEVEL, 16, MODE, 4, TIO>(x, y, wpack, bpack, p, s);
}

template <int W0, int LEVEL, int MODE, typename TIO>
static hipError_t launch16_banded(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    if (p.waves == 8) return launch_bw<W0, LEVEL, 16, MODE, 8, 4, TIO>(x, y, wpack, bpack, p, s);
    return launch_bw<W0, LEVEL, 16, MODE, 4, 4, TIO>(x, y, wpack, bpack, p, s);
}

template <int MODE, typename TIO>
static hipError_t launch16(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    if (p.w0 == 16) return launch16_whole<16, 1, MODE, TIO>(x, y, wpack, bpack, p, s);
    if (p.w0 == 32) return launch16_banded<32, 2, MODE, TIO>(x, y, wpack, bpack, p, s);
    if (p.w0 == 64) return launch16_banded<64, 3, MODE, TIO>(x, y, wpack, bpack, p, s);
    return hipErrorInvalidConfiguration;
}

}  // namespace lanes

hipError_t lanes16_recconv(const void* x, void* y, const float* wpack, const float* bpack,
                           int N, int C, int H, int W, int level, int k, int mode, int dtype, hipStream_t s)
{
    const lanes::LanesPlan p = lanes::plan(N, C, H, W, level, k, dtype);
    if (!p.ok || p.w0 % 16 != 0) return hipErrorInvalidConfiguration;
    if (dtype == 1) return mode == 1 ? lanes::launch16<1, bf16_t>(x, y, wpack, bpack, p, s) : lanes::launch16<0, bf16_t>(x, y, wpack, bpack, p, s);
    return mode == 1 ? lanes::launch16<1, float>(x, y, wpack, bpack, p, s) : lanes::launch16<0, float>(x, y, wpack, bpack, p, s);
}

}  // namespace rcx
