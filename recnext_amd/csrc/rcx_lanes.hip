// Register-resident RecConv2d for the 7 * 2^k planes (7x7, 14x14, 28x28 ...): the whole pyramid of
// model/recnext.py:24-34 lives in VGPRs, horizontal taps come from DPP lane shifts, vertical taps are
// register indices.  LDS only transposes NHWC global memory <-> the lane layout.
//
// Lane layout.  A wave owns 64/LPC channels of one image; LPC consecutive lanes own one channel, LA of
// them active (7 of 8, or 14 of 16) -- the inactive "guard" lanes are EXEC-disabled for the whole
// compute section, so a DPP shift that crosses a channel boundary reads zero (bound_ctrl semantics,
// tools/ubench/dpp_fmac.hip): the guard lanes ARE the horizontal zero padding.  At pyramid level l
// (width W_l = W_0 / 2^l while that is >= LA) lane j of the group holds the B_l = W_l / LA adjacent
// columns j*B_l .. j*B_l+B_l-1 of every row, as registers row[r][0..B_l-1].  The stride-2 convolution
// maps (W, B) -> (W/2, B/2) with no lane movement at all: the de-interleave is free.  Below B = 1 the
// levels are "dilated": column c of the 4-wide level sits in lane c*D.
//
// All shapes, row indices and the vertical resize tables are compile-time; the horizontal resize uses
// per-lane weights computed once with ATen's float index arithmetic (rcx_common.h), so the border
// clamping and the irregular 4 -> 7 step need no special code.
#include "rcx_lanes.h"
#include "rcx_launch.h"

namespace rcx {
namespace lanes {

// Diagnostic build only (-DRCX_STAMPS): thread 0 of the first workgroups records the cycle counter at phase boundaries
// (tools/stamps_lanes.py).  The shipped library compiles these to nothing.
#ifdef RCX_STAMPS
__device__ unsigned long long* g_lane_stamps = nullptr;
#define RCX_LSTAMP(id)                                                                                   \
    do {                                                                                                \
        if (threadIdx.x == 0 && g_lane_stamps && blockIdx.x < 256)                                      \
            g_lane_stamps[blockIdx.x * 64 + (id)] = __builtin_readcyclecounter();                       \
    } while (0)
#else
#define RCX_LSTAMP(id) do { } while (0)
#endif

}  // namespace lanes
}  // namespace rcx

#include "rcx_lanes_kernels.h"

namespace rcx {
namespace lanes {

template <int MODE, typename TIO>
static hipError_t launch_m(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    if (p.w0 == 7) return launch_t<7, 1, 8, MODE, TIO>(x, y, wpack, bpack, p, s);
    if (p.w0 == 14) return launch_t<14, 2, 8, MODE, TIO>(x, y, wpack, bpack, p, s);
    if (p.w0 == 28) return launch_b<28, 3, 8, MODE, 4, TIO>(x, y, wpack, bpack, p, s);
    if (p.w0 == 56) return launch_b<56, 4, 16, MODE, 4, TIO>(x, y, wpack, bpack, p, s);
    return hipErrorInvalidConfiguration;
}


#ifdef RCX_STAMPS
hipError_t set_stamp_buffer(void* p)
{
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stamps), &q, sizeof(q));
}
#endif

}  // namespace lanes

bool lanes_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    return cpt_applicable(N, C, H, W, level, k, dtype) || cpl14_applicable(N, C, H, W, level, k, dtype) || cpl14_short_applicable(N, C, H, W, level, k, dtype) || cpl7b_applicable(N, C, H, W, level, k, dtype) || lanes::plan(N, C, H, W, level, k, dtype).ok;
}

int lanes_describe(int N, int C, int H, int W, int level, int k, int mode, int dtype, char* buf, int len)
{
    if (cpt_applicable(N, C, H, W, level, k, dtype)) return cpt_describe(N, C, H, level, mode, dtype, buf, len);
    if (cpl14_applicable(N, C, H, W, level, k, dtype)) return cpl14_describe(N, C, mode, dtype, buf, len);
    if (cpl14_short_applicable(N, C, H, W, level, k, dtype)) return cpl14_short_describe(N, C, mode, buf, len);
    if (cpl7b_applicable(N, C, H, W, level, k, dtype)) return cpl7b_describe(N, C, mode, buf, len);
    const lanes::LanesPlan p = lanes::plan(N, C, H, W, level, k, dtype);
    if (!p.ok) return 0;
    // kernel=<template arguments>: W0, LEVEL, lanes per channel, mode, waves[, band rows] -- the name rocprofv3 reports
    char kern[64];
    if (p.banded) snprintf(kern, sizeof(kern), "k_recconv_lanes_banded<%d, %d, %d, %d, %d, %d>", p.w0, p.level, p.lpc, mode, p.waves, p.sr);
    else snprintf(kern, sizeof(kern), "k_recconv_lanes<%d, %d, %d, %d, %d>", p.w0, p.level, p.lpc, mode, p.waves);
    return snprintf(buf, len, "lanes(%s,cb=%d,ni=%d,nt=%d,lds=%zu)", kern, p.waves * 64 / p.lpc, p.args.ni, p.waves * 64, p.lds);
}

hipError_t lanes_recconv(const void* x, void* y, const float* wpack, const float* bpack,
                         int N, int C, int H, int W, int level, int k, int mode, int dtype, hipStream_t s)
{
    if (cpt_applicable(N, C, H, W, level, k, dtype)) return cpt_recconv(x, y, wpack, bpack, N, C, H, level, mode, dtype, s);
    if (cpl14_applicable(N, C, H, W, level, k, dtype)) return cpl14_recconv(x, y, wpack, bpack, N, C, mode, dtype, s);
    if (cpl14_short_applicable(N, C, H, W, level, k, dtype)) return cpl14_short_recconv(x, y, wpack, bpack, N, C, mode, dtype, s);
    if (cpl7b_applicable(N, C, H, W, level, k, dtype)) return cpl7b_recconv(x, y, wpack, bpack, N, C, mode, dtype, s);
    const lanes::LanesPlan p = lanes::plan(N, C, H, W, level, k, dtype);
    if (!p.ok) return hipErrorInvalidConfiguration;
    if (p.w0 % 16 == 0) return lanes16_recconv(x, y, wpack, bpack, N, C, H, W, level, k, mode, dtype, s);
    if (dtype == 1) return mode == 1 ? lanes::launch_m<1, bf16_t>(x, y, wpack, bpack, p, s) : lanes::launch_m<0, bf16_t>(x, y, wpack, bpack, p, s);
    return mode == 1 ? lanes::launch_m<1, float>(x, y, wpack, bpack, p, s) : lanes::launch_m<0, float>(x, y, wpack, bpack, p, s);
}

}  // namespace rcx
