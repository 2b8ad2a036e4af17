// The exact (erf) GELU of the block's channel mixer and of the stem (model/recnext.py:125-146: nn.GELU()), two values at a time on the packed float32 pipe.
#pragma once
#include <hip/hip_runtime.h>

namespace rcx {

typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));

// erf(x) ~ xc Q(xc^2) with xc = x clamped to [-A, A], A = 2.799997: a weighted minimax fit constrained to reach 1 at the clamp, |error| < 7.7e-5; two values
// at a time on the packed pipe, one v_med3 each for the clamp.  A is not 2.8 but the float32 twelve ulps below it at which the float32 evaluation of xc Q(xc^2)
// below (this multiply, these seven FMAs) returns EXACTLY +-1 (round 6): past the clamp 1 + erf is exactly 2 or 0, so gelu(v) is exactly v or 0 for
// large |v| -- at 2.8 the fit returned +-0.9999983 there and a large negative pre-activation left -1.7e-6 |v| instead of 0, an error growing with |v|.
// (Just inside the clamp the fit overshoots 1 by at most 1.9e-6: bounded by 8e-6 in gelu, inside the fit's own error.)
__device__ __forceinline__ gelu_f32x2 gelu2(gelu_f32x2 v)
{
    constexpr float A = 2.799997f;
    const gelu_f32x2 x = v * 0.70710678f;
    const gelu_f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -A, A), __builtin_amdgcn_fmed3f(x.y, -A, A)};
    const gelu_f32x2 s = xc * xc;
    gelu_f32x2 q = {-4.114877470e-07f, -4.114877470e-07f};
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{1.744569090e-05f, 1.744569090e-05f});
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{-3.191421274e-04f, -3.191421274e-04f});
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{3.352143336e-03f, 3.352143336e-03f});
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{-2.280939557e-02f, -2.280939557e-02f});
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{1.079412624e-01f, 1.079412624e-01f});
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{-3.734020293e-01f, -3.734020293e-01f});
    q = __builtin_elementwise_fma(q, s, gelu_f32x2{1.127931833e+00f, 1.127931833e+00f});
    const gelu_f32x2 e = xc * q, hv = v * 0.5f;
    return __builtin_elementwise_fma(hv, e, hv);
}

// The same on N pairs in lockstep: every step of the polynomial is issued for all N pairs before the next one, so that a wave alone on its SIMD (the streamed channel-mixer
// kernels) does not wait out each dependent packed op's latency -- the compiler keeps the N chains of gelu2() calls one after the other.
template <int N>
__device__ __forceinline__ void gelu2_batch(gelu_f32x2 (&v)[N])
{
    constexpr float A = 2.799997f;
    gelu_f32x2 xc[N], s[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const gelu_f32x2 x = v[i] * 0.70710678f;
        xc[i] = gelu_f32x2{__builtin_amdgcn_fmed3f(x.x, -A, A), __builtin_amdgcn_fmed3f(x.y, -A, A)};
    }
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = xc[i] * xc[i];
    constexpr float C[8] = {1.127931833e+00f, -3.734020293e-01f, 1.079412624e-01f, -2.280939557e-02f, 3.352143336e-03f, -3.191421274e-04f, 1.744569090e-05f, -4.114877470e-07f};
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(gelu_f32x2{C[7], C[7]}, s[i], gelu_f32x2{C[6], C[6]});
#pragma unroll
    for (int k = 5; k >= 0; --k)
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], gelu_f32x2{C[k], C[k]});
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const gelu_f32x2 e = xc[i] * q[i], hv = v[i] * 0.5f;
        v[i] = __builtin_elementwise_fma(hv, e, hv);
    }
}

// TWICE the GELU, v (1 + erf(v / sqrt 2)), on N pairs in lockstep, with the 1 / sqrt 2 folded into the coefficients (erf(v / sqrt 2) ~ vc Q'(vc^2), vc = v clamped to
// +-3.95979 = the float32 33 ulps below 2.8 sqrt 2 at which this evaluation returns exactly +-1: see gelu2) and the final 0.5 left to the consumer (the channel mixer's W2 pack is stored halved, which is exact in bf16): 12 vector ops per pair instead of 15.
template <int N>
__device__ __forceinline__ void gelu2x_batch(gelu_f32x2 (&v)[N])
{
    constexpr float A = 3.95979f;
    gelu_f32x2 vc[N], s[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) vc[i] = gelu_f32x2{__builtin_amdgcn_fmed3f(v[i].x, -A, A), __builtin_amdgcn_fmed3f(v[i].y, -A, A)};
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = vc[i] * vc[i];
    constexpr float C[8] = {7.975682616e-01f, -1.320175529e-01f, 1.908149943e-02f, -2.016084734e-03f, 1.481452055e-04f, -7.052111414e-06f, 1.927494679e-07f, -2.273170097e-09f};
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(gelu_f32x2{C[7], C[7]}, s[i], gelu_f32x2{C[6], C[6]});
#pragma unroll
    for (int k = 5; k >= 0; --k)
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], gelu_f32x2{C[k], C[k]});
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_elementwise_fma(v[i], vc[i] * q[i], v[i]);
}

}  // namespace rcx
