// Device-side helpers shared by the gfx950 kernels: typed vector loads/stores for NHWC channel
// vectors, bf16 <-> f32, and the resize index arithmetic (ATen upsample semantics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rcx {

typedef uint16_t bf16_t;   // raw bf16 bits
typedef _Float16 f16_t;    // IEEE half (the reference's autocast dtype, engine.py:48); dtype id RCX_DTYPE_F16 = 2

// dtype id of an element type (include/recnext_amd.h)
template <typename T> struct DtId;
template <> struct DtId<float> { static constexpr int id = 0; };
template <> struct DtId<bf16_t> { static constexpr int id = 1; };
template <> struct DtId<f16_t> { static constexpr int id = 2; };

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even through the compiler's native conversion (v_cvt_pk_bf16_f32 on gfx950,
// NaN stays NaN -- MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

template <int V> struct FVec { float v[V]; };

// ---- channel-vector loads: V consecutive channels starting at p (p aligned to V*sizeof(T)) ----
template <int V>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&out)[V])
{
    if constexpr (V == 1) { out[0] = p[0]; }
    else if constexpr (V == 2) { float2 t = *reinterpret_cast<const float2*>(p); out[0] = t.x; out[1] = t.y; }
    else {
        static_assert(V % 4 == 0, "V");
#pragma unroll
        for (int i = 0; i < V / 4; ++i) {
            float4 t = reinterpret_cast<const float4*>(p)[i];
            out[4 * i] = t.x; out[4 * i + 1] = t.y; out[4 * i + 2] = t.z; out[4 * i + 3] = t.w;
        }
    }
}

template <int V>
__device__ __forceinline__ void load_vec(const bf16_t* __restrict__ p, float (&out)[V])
{
    if constexpr (V == 1) { out[0] = bf16_to_f32(p[0]); }
    else if constexpr (V == 2) {
        uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        out[0] = __uint_as_float(t << 16); out[1] = __uint_as_float(t & 0xffff0000u);
    } else if constexpr (V == 4) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        out[0] = __uint_as_float(t.x << 16); out[1] = __uint_as_float(t.x & 0xffff0000u);
        out[2] = __uint_as_float(t.y << 16); out[3] = __uint_as_float(t.y & 0xffff0000u);
    } else {
        static_assert(V == 8, "V");
        uint4 t = *reinterpret_cast<const uint4*>(p);
        out[0] = __uint_as_float(t.x << 16); out[1] = __uint_as_float(t.x & 0xffff0000u);
        out[2] = __uint_as_float(t.y << 16); out[3] = __uint_as_float(t.y & 0xffff0000u);
        out[4] = __uint_as_float(t.z << 16); out[5] = __uint_as_float(t.z & 0xffff0000u);
        out[6] = __uint_as_float(t.w << 16); out[7] = __uint_as_float(t.w & 0xffff0000u);
    }
}

template <int V>
__device__ __forceinline__ void load_vec(const f16_t* __restrict__ p, float (&out)[V])
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    if constexpr (V == 1) { out[0] = (float)p[0]; }
    else {
        static_assert(V == 2 || V == 4 || V == 8, "V");
        uint32_t raw[V / 2];
        if constexpr (V == 2) raw[0] = *reinterpret_cast<const uint32_t*>(p);
        else if constexpr (V == 4) { uint2 t = *reinterpret_cast<const uint2*>(p); raw[0] = t.x; raw[1] = t.y; }
        else { uint4 t = *reinterpret_cast<const uint4*>(p); raw[0] = t.x; raw[1] = t.y; raw[2] = t.z; raw[3] = t.w; }
#pragma unroll
        for (int i = 0; i < V / 2; ++i) {
            const h2 h = __builtin_bit_cast(h2, raw[i]);
            out[2 * i] = (float)h.x; out[2 * i + 1] = (float)h.y;
        }
    }
}

// round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, h);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <int V>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&in)[V])
{
    if constexpr (V == 1) { p[0] = in[0]; }
    else if constexpr (V == 2) { *reinterpret_cast<float2*>(p) = make_float2(in[0], in[1]); }
    else {
#pragma unroll
        for (int i = 0; i < V / 4; ++i)
            reinterpret_cast<float4*>(p)[i] = make_float4(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]);
    }
}

template <int V>
__device__ __forceinline__ void store_vec(bf16_t* __restrict__ p, const float (&in)[V])
{
    if constexpr (V == 1) { p[0] = f32_to_bf16(in[0]); }
    else if constexpr (V == 2) { *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(in[0], in[1]); }
    else if constexpr (V == 4) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]));
    } else {
        static_assert(V == 8, "V");
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]),
                                                  pack_bf16x2(in[4], in[5]), pack_bf16x2(in[6], in[7]));
    }
}

template <int V>
__device__ __forceinline__ void store_vec(f16_t* __restrict__ p, const float (&in)[V])
{
    if constexpr (V == 1) { p[0] = (f16_t)in[0]; }
    else if constexpr (V == 2) { *reinterpret_cast<uint32_t*>(p) = pack_f16x2(in[0], in[1]); }
    else if constexpr (V == 4) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_f16x2(in[0], in[1]), pack_f16x2(in[2], in[3]));
    } else {
        static_assert(V == 8, "V");
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_f16x2(in[0], in[1]), pack_f16x2(in[2], in[3]),
                                                  pack_f16x2(in[4], in[5]), pack_f16x2(in[6], in[7]));
    }
}

// one element of any supported type as float32
__device__ __forceinline__ float elem_to_f32(float v) { return v; }
__device__ __forceinline__ float elem_to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float elem_to_f32(f16_t v) { return (float)v; }

// ---- resize source indices (per axis), float arithmetic exactly as ATen's upsample kernels ----
struct Lerp { int i0, i1; float lam; };

// bilinear, align_corners=False: src = max(scale*(d+0.5)-0.5, 0); i0=floor(src); i1=i0+(i0<in-1)
__device__ __forceinline__ Lerp bilinear_src(int d, int n_in, float scale)
{
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;                       // src >= 0 so truncation == floor
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    r.lam = src - (float)i0;
    return r;
}

// legacy nearest: i = min(floor(d*scale), in-1)
__device__ __forceinline__ int nearest_src(int d, int n_in, float scale)
{
    int i = (int)((float)d * scale);
    return i < n_in - 1 ? i : n_in - 1;
}

}  // namespace rcx
