// Channel-per-lane RecConv2d for the 7x7 / level 1 block (model/recnext.py:24-34 at the last stage of RecNeXt-M*):
// one LANE owns a whole (image, channel) plane -- 49 pixels in registers -- so the two convolutions on 7x7, the conv on
// 4x4, the resize and the add need no neighbour exchange at all: no DPP (a DPP move would put the SIMD into its slow
// issue mode, rcx_lanes.h), no LDS, no barrier.  A wave covers 64 consecutive channels of one image; in NHWC those are
// 128 contiguous bytes per pixel (bf16), so every load and store instruction moves whole sectors.  The zero padding is
// resolved at compile time (border taps are simply not issued: 841 instead of 1225 FMAs for the 7x7 conv); the 7x7 conv
// runs two columns per v_pk_fma_f32.  Same arithmetic as the lanes kernels (float32 throughout, one final rounding).
#include <hip/hip_runtime.h>
#include "rcx_opts.h"
#include <stdlib.h>

#include "rcx_common.h"
#include "rcx_lanes.h"
#include "rcx_launch.h"

namespace rcx {
namespace cpl {

using lanes::f32x2;
using lanes::vtab;
using lanes::VT;

template <typename TIO> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct IO<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 25 taps + bias of conv `conv` for this lane's channel (tap-major pack: consecutive channels are consecutive floats)
__device__ __forceinline__ void load_conv(const float* __restrict__ wpack, const float* __restrict__ bpack, int conv, int C, int c,
                                          int has_bias, float (&w)[25], float& b)
{
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = wpack[((size_t)conv * 25 + t) * C + c];
    b = has_bias ? bpack[(size_t)conv * C + c] : 0.f;
}

template <int MODE, typename TIO>
__global__ __launch_bounds__(64)
void k_recconv_cpl7(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                    int N, int C, int has_bias)
{
    constexpr int W = 7, WC = 4;
    const int nb = C / 64;
    const int cb = blockIdx.x % nb, n = blockIdx.x / nb;
    if (n >= N) return;
    const int c = cb * 64 + threadIdx.x;
    const TIO* xp = x + (size_t)n * W * W * C + c;
    TIO* yp = y + (size_t)n * W * W * C + c;

    float X[W][W];
#pragma unroll
    for (int r = 0; r < W; ++r)
#pragma unroll
        for (int q = 0; q < W; ++q) X[r][q] = IO<TIO>::ld(xp + (size_t)(r * W + q) * C);

    float w[25], b;
    // ---- F1 = down(x): stride 2, pad 2, 7x7 -> 4x4 (conv 0 of the pack)
    load_conv(wpack, bpack, 0, C, c, has_bias, w, b);
    float F1[WC][WC];
#pragma unroll
    for (int o = 0; o < WC; ++o)
#pragma unroll
        for (int i = 0; i < WC; ++i) {
            float a = b;
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const int r = 2 * o + u - 2, q = 2 * i + v - 2;
                    if (r >= 0 && r < W && q >= 0 && q < W) a = fmaf(X[r][q], w[u * 5 + v], a);
                }
            F1[o][i] = a;
        }
    // ---- C1 = conv_0(F1) on 4x4 (conv 1 of the pack)
    load_conv(wpack, bpack, 1, C, c, has_bias, w, b);
    float C1[WC][WC];
#pragma unroll
    for (int o = 0; o < WC; ++o)
#pragma unroll
        for (int i = 0; i < WC; ++i) {
            float a = b;
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const int r = o + u - 2, q = i + v - 2;
                    if (r >= 0 && r < WC && q >= 0 && q < WC) a = fmaf(F1[r][q], w[u * 5 + v], a);
                }
            C1[o][i] = a;
        }
    // ---- T0 = x + resize(C1, 7x7): columns first, then rows (the order of the lanes kernels), ATen index arithmetic
    float Hh[WC][W];
#pragma unroll
    for (int i = 0; i < WC; ++i)
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const VT t = vtab(MODE, WC, W, q);
            Hh[i][q] = (MODE == 1 || t.i0 == t.i1) ? C1[i][t.i0] : fmaf(t.l, C1[i][t.i1], (1.f - t.l) * C1[i][t.i0]);
        }
#pragma unroll
    for (int r = 0; r < W; ++r) {
        const VT t = vtab(MODE, WC, W, r);
#pragma unroll
        for (int q = 0; q < W; ++q)
            X[r][q] += (MODE == 1 || t.i0 == t.i1) ? Hh[t.i0][q] : fmaf(t.l, Hh[t.i1][q], (1.f - t.l) * Hh[t.i0][q]);
    }
    // ---- y = conv_1(T0) on 7x7 (conv 2 of the pack): column pairs (0,1) (2,3) (4,5) packed, column 6 scalar
    load_conv(wpack, bpack, 2, C, c, has_bias, w, b);
#pragma unroll
    for (int o = 0; o < W; ++o) {
        f32x2 a2[3] = {f32x2{b, b}, f32x2{b, b}, f32x2{b, b}};
        float a6 = b;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int r = o + u - 2;
            if (r < 0 || r >= W) continue;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float wv = w[u * 5 + v];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int q0 = 2 * k + v - 2, q1 = q0 + 1;
                    const bool ok0 = q0 >= 0 && q0 < W, ok1 = q1 >= 0 && q1 < W;
                    if (ok0 && ok1) a2[k] = __builtin_elementwise_fma(f32x2{X[r][q0], X[r][q1]}, f32x2{wv, wv}, a2[k]);
                    else if (ok0) a2[k].x = fmaf(X[r][q0], wv, a2[k].x);
                    else if (ok1) a2[k].y = fmaf(X[r][q1], wv, a2[k].y);
                }
                const int q6 = 6 + v - 2;
                if (q6 >= 0 && q6 < W) a6 = fmaf(X[r][q6], wv, a6);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            IO<TIO>::st(yp + (size_t)(o * W + 2 * k) * C, a2[k].x);
            IO<TIO>::st(yp + (size_t)(o * W + 2 * k + 1) * C, a2[k].y);
        }
        IO<TIO>::st(yp + (size_t)(o * W + 6) * C, a6);
    }
}

static inline bool enabled()
{
    const char* v = rcx::opt::value(rcx::opt::CPL);
    return !(v && *v == '0');
}

}  // namespace cpl

bool cpl7_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    (void)N;
    return cpl::enabled() && H == 7 && W == 7 && level == 1 && k == 5 && C % 64 == 0 && (dtype == 0 || dtype == 1);
}

int cpl7_describe(int N, int C, int mode, char* buf, int len)
{
    return snprintf(buf, len, "cpl(k_recconv_cpl7<%d>,cb=64,nt=64,blocks=%d,lds=0)", mode, N * (C / 64));
}

hipError_t cpl7_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * (C / 64));
    const int hb = bpack != nullptr;
    if (dtype == 1) {
        if (mode == 1) hipLaunchKernelGGL((cpl::k_recconv_cpl7<1, bf16_t>), dim3(grid), dim3(64), 0, s, (const bf16_t*)x, (bf16_t*)y, wpack, bpack, N, C, hb);
        else hipLaunchKernelGGL((cpl::k_recconv_cpl7<0, bf16_t>), dim3(grid), dim3(64), 0, s, (const bf16_t*)x, (bf16_t*)y, wpack, bpack, N, C, hb);
    } else {
        if (mode == 1) hipLaunchKernelGGL((cpl::k_recconv_cpl7<1, float>), dim3(grid), dim3(64), 0, s, (const float*)x, (float*)y, wpack, bpack, N, C, hb);
        else hipLaunchKernelGGL((cpl::k_recconv_cpl7<0, float>), dim3(grid), dim3(64), 0, s, (const float*)x, (float*)y, wpack, bpack, N, C, hb);
    }
    return hipGetLastError();
}

}  // namespace rcx
