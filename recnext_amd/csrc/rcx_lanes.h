// Shared pieces of the register-resident ("lanes") kernels: compile-time loops, DPP lane shifts, raw LDS element access.
#pragma once
#include <utility>

#include "rcx_common.h"

namespace rcx {
namespace lanes {

template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
// compile-time loop: f(IC<0>) ... f(IC<N-1>)
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

#define RCX_INL __attribute__((always_inline))
#ifndef RCX_ROW_FENCE
#define RCX_ROW_FENCE __builtin_amdgcn_sched_barrier(0)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // plain vector loads/stores (HIP's uint4 copies as memcpy)

// ------------------------------------------------------------------------------------------------
// lane shifts inside a 16-lane DPP row; lanes shifted in from outside the row, or from an
// EXEC-disabled lane, read 0
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// value held by the lane N below (LPC == 8: chained single steps, so a value never jumps over the guard lane)
template <int N, int LPC>
__device__ __forceinline__ float from_left(float v)
{
    if constexpr (N == 0) return v;
    else if constexpr (LPC == 16) return dpp_mov<0x110 + N>(v);      // row_shr:N
    else return from_left<N - 1, LPC>(dpp_mov<0x111>(v));
}

template <int N, int LPC>
__device__ __forceinline__ float from_right(float v)
{
    if constexpr (N == 0) return v;
    else if constexpr (LPC == 16) return dpp_mov<0x100 + N>(v);      // row_shl:N
    else return from_right<N - 1, LPC>(dpp_mov<0x101>(v));
}

// ------------------------------------------------------------------------------------------------
template <typename TIO> struct Raw;
template <> struct Raw<float> {
    static __device__ __forceinline__ float ld(const unsigned char* p) { return *reinterpret_cast<const float*>(p); }
    static __device__ __forceinline__ void st(unsigned char* p, float v) { *reinterpret_cast<float*>(p) = v; }
};
template <> struct Raw<bf16_t> {
    static __device__ __forceinline__ float ld(const unsigned char* p) { return bf16_to_f32(*reinterpret_cast<const bf16_t*>(p)); }
    static __device__ __forceinline__ void st(unsigned char* p, float v) { *reinterpret_cast<bf16_t*>(p) = f32_to_bf16(v); }
};

// Pixel slot inside the raw LDS image: within a row, column c = lane*B0 + j is stored at slot j*LA + lane, so the LA lanes
// of a channel read consecutive pixels (PITCH apart: 8 distinct banks) instead of pixels B0*PITCH apart (2 banks).
template <int W0, int B0, int LA>
__device__ __forceinline__ int lds_slot(int p)
{
    const int row = p / W0, col = p - row * W0;
    return row * W0 + (col % B0) * LA + col / B0;
}

}  // namespace lanes
}  // namespace rcx
