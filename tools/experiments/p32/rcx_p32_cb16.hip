// RecConv2d on the 32 x 32 plane, level 2 (32 -> 16 -> 8; RecNeXt-M3's stage 2 on a 512 x 512 detection input: 13 of its 21 blocks --
// BASELINE config 5, model/recnext.py:24-34 with detection/recnext.py's input): ONE launch in which a channel's plane is spread over SIXTEEN lanes.
//
// Why another schedule for a plane the register-resident banded kernel (rcx_lanes16.hip, round 1) already fuses: at the batch this config runs
// (N = 32, C = 256) there are 8 192 channel planes for 1 024 SIMDs; a lane that owns a plane (or a band of it) leaves the chip a quarter full, and the
// banded kernel's DPP row shifts run the SIMD in its slow issue mode (23 - 28 us = 0.15 - 0.18 of the HBM roofline, profiles/r03c_512_*).  Here
//   a WORKGROUP = one image x 32 channels, 512 lanes = 32 channels x 16 parts; everything the block touches lives in LDS, CHANNEL-major
//   ([channel][pixel], rows of 8 / 4 pixels read as 16-byte vectors; the channel pitch is padded so that 8 consecutive lanes cover the 32 banks):
//     x (16-bit, 64.5 KB), F1 / T1 (float32 16 x 16, in place), C1, F2, C2 -- 146.5 KB, one workgroup per CU, N x C / 32 workgroups (256 at N = 32);
//   phases, a barrier between them, every lane a row (or two) of the phase's plane, input-row stationary, float32 accumulation:
//     0 x rows 2p, 2p+1 -> LDS      1 F1 = down(x) row p           2 F2 = down(F1) row p (p < 8)      3 C2 = conv0(F2) row p (p < 8)
//     4 T1 = F1 + resize(C2) row p  5 C1 = conv1(T1) row p         6 y rows 2p, 2p+1 = conv2(x + resize(C1)), 16-bit stores.
// Same sums as every other schedule (float32, taps in ky, kx order per input row), bilinear resize in its closed 2x form (0.25 / 0.75, clamped ends:
// ATen's align_corners = False arithmetic for an exact factor of two) or nearest.  16-bit activations only (a float32 x plane does not fit).
// RCX_P32=0: the banded kernel (A/B).
#include "rcx_common.h"
#include "rcx_launch.h"
#include "rcx_opts.h"
#include <stdlib.h>

namespace rcx {
namespace p32 {

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));

#ifndef RCX_P32_CB
#define RCX_P32_CB 16
#endif
constexpr int CB = RCX_P32_CB, NT = 16 * CB, NPART = NT / CB, RP = 32 / NPART, H0 = 32, H1 = 16, H2 = 8;      // RP: rows of the 32 x 32 plane per lane
constexpr int XP = H0 * H0 * 2 + 16;          // bytes of a channel's x plane (16-bit) + pad: pitch = 516 dwords = 4 mod 32
constexpr int P1 = H1 * H1 * 4 + 16;          // float32 16 x 16 + pad: 260 dwords
constexpr int P2 = H2 * H2 * 4 + 16;          // float32 8 x 8 + pad: 68 dwords
constexpr int OX = 0, OF1 = OX + CB * XP, OC1 = OF1 + CB * P1, OF2 = OC1 + CB * P1, OC2 = OF2 + CB * P2, LDS_BYTES = OC2 + CB * P2;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

template <typename TIO> __device__ __forceinline__ float lo16(unsigned v);
template <typename TIO> __device__ __forceinline__ float hi16(unsigned v);
template <> __device__ __forceinline__ float lo16<bf16_t>(unsigned v) { return __uint_as_float(v << 16); }
template <> __device__ __forceinline__ float hi16<bf16_t>(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
template <> __device__ __forceinline__ float lo16<f16_t>(unsigned v) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(v & 0xffffu)); }
template <> __device__ __forceinline__ float hi16<f16_t>(unsigned v) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(v >> 16)); }

// a row of 32 16-bit pixels of this lane's channel from the LDS image -> float32
template <typename TIO>
__device__ __forceinline__ void read_xrow(const char* lx, int y, float (&out)[H0])
{
    const u32x4p* p = reinterpret_cast<const u32x4p*>(lx + y * (H0 * 2));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4p v = p[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { out[8 * q + 2 * e] = lo16<TIO>(v[e]); out[8 * q + 2 * e + 1] = hi16<TIO>(v[e]); }
    }
}
template <int W>
__device__ __forceinline__ void read_frow(const char* lp, int y, float (&out)[W])
{
    const f32x4p* p = reinterpret_cast<const f32x4p*>(lp + y * (W * 4));
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
        const f32x4p v = p[q];
        out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
}
template <int W>
__device__ __forceinline__ void write_frow(char* lp, int y, const float (&in)[W])
{
    f32x4p* p = reinterpret_cast<f32x4p*>(lp + y * (W * 4));
#pragma unroll
    for (int q = 0; q < W / 4; ++q) p[q] = f32x4p{in[4 * q], in[4 * q + 1], in[4 * q + 2], in[4 * q + 3]};
}

// dst += row r of resize(src) (destination 2 WS pixels wide; src is HS x WS float32 in LDS): the two source rows and their weights, then the row
template <int MODE, int WS>
__device__ __forceinline__ void add_resized_row(const char* lsrc, int r, int HS, float (&dst)[2 * WS])
{
    float v[WS];
    if constexpr (MODE == 1) {
        read_frow<WS>(lsrc, r >> 1, v);
#pragma unroll
        for (int j = 0; j < WS; ++j) { dst[2 * j] += v[j]; dst[2 * j + 1] += v[j]; }
    } else {
        const int i = r >> 1;
        const bool odd = r & 1;
        const int i0 = odd ? i : (i > 0 ? i - 1 : 0), i1 = odd ? (i < HS - 1 ? i + 1 : i) : i;
        const float l1 = odd ? 0.25f : 0.75f, l0 = 1.f - l1;      // destination 2i: 0.25 src[i-1] + 0.75 src[i]; 2i+1: 0.75 src[i] + 0.25 src[i+1]
        {
            float b[WS];
            read_frow<WS>(lsrc, i0, v);
            read_frow<WS>(lsrc, i1, b);
#pragma unroll
            for (int j = 0; j < WS; ++j) v[j] = l0 * v[j] + l1 * b[j];
        }
#pragma unroll
        for (int j = 0; j < WS; ++j) {
            dst[2 * j] += 0.25f * v[j > 0 ? j - 1 : 0] + 0.75f * v[j];
            dst[2 * j + 1] += 0.75f * v[j] + 0.25f * v[j < WS - 1 ? j + 1 : j];
        }
    }
}

// acc[j] += sum_dx w[dy][dx] * row[S j - 2 + dx] (zero outside the row): one input row's share of a 5-tap row of outputs, stride S
template <int S, int WIN, int WOUT>
__device__ __forceinline__ void row_taps(float (&acc)[WOUT], const float (&row)[WIN], const float (&w)[25], int dy)
{
#pragma unroll
    for (int j = 0; j < WOUT; ++j)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int xx = S * j - 2 + dx;
            if (xx >= 0 && xx < WIN) acc[j] = fmaf(w[dy * 5 + dx], row[xx], acc[j]);
        }
}

template <typename TIO, int MODE>
__global__ void __launch_bounds__(NT)          // 16 channels: 256 lanes and 73 KB of LDS a workgroup, two per CU, in different phases
k_recconv_p32(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack, int N, int C)
{
    extern __shared__ __attribute__((aligned(16))) char lds_p[];
    const int ch = threadIdx.x & (CB - 1), part = threadIdx.x / CB;               // a wave = CB channels x 64 / CB consecutive parts
    const int nb = (C + CB - 1) / CB;
    const int n = blockIdx.x / nb, cb = blockIdx.x - n * nb;
    const int c = cb * CB + ch;
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;
    char* const lx = lds_p + OX + ch * XP;
    char* const lf1 = lds_p + OF1 + ch * P1;
    char* const lc1 = lds_p + OC1 + ch * P1;
    char* const lf2 = lds_p + OF2 + ch * P2;
    char* const lc2 = lds_p + OC2 + ch * P2;
    auto taps = [&](float (&w)[25], float& b, int idx) {
#pragma unroll
        for (int t = 0; t < 25; ++t) w[t] = wpack[(size_t)(idx * 25 + t) * C + cc];
        b = bpack ? bpack[(size_t)idx * C + cc] : 0.f;
    };

    // ---- 0. x rows 2 part, 2 part + 1 of this channel -> LDS (two bytes per lane and request: a wave's request is 2 x 64 contiguous bytes)
    const int pix = C * 2;                                              // bytes between horizontally adjacent pixels
    const unsigned OOB = 0x80000000u;
    {
        // the image as a raw buffer: a lane's offset = its row and channel (an invalid channel: out of range, reads 0), the pixel of the row in the
        // scalar offset -- 32-bit address arithmetic, one scalar multiply per request
        const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(x + ((size_t)n * H0 * H0) * C), 0, H0 * H0 * pix, 0x00020000);
#pragma unroll
        for (int rr = 0; rr < RP; ++rr) {
            const int yy = RP * part + rr;
            const unsigned vo = cvalid ? (unsigned)(yy * H0 * pix + c * 2) : OOB;
            unsigned short v[H0];
#pragma unroll
            for (int xx = 0; xx < H0; ++xx) v[xx] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xsrc, vo, xx * pix, 0);
            u32x4p* dst = reinterpret_cast<u32x4p*>(lx + yy * (H0 * 2));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[q] = u32x4p{(unsigned)v[8 * q] | ((unsigned)v[8 * q + 1] << 16), (unsigned)v[8 * q + 2] | ((unsigned)v[8 * q + 3] << 16),
                                (unsigned)v[8 * q + 4] | ((unsigned)v[8 * q + 5] << 16), (unsigned)v[8 * q + 6] | ((unsigned)v[8 * q + 7] << 16)};
        }
    }
    // the next conv's taps are requested a phase ahead (wn), so that their trip to L2 runs behind the current phase
    float w[25], bias, wn[25], biasn;
    taps(w, bias, 0);
    __syncthreads();
    taps(wn, biasn, 1);                                                // convs[0]: the coarsest level (phase 3)

    // ---- 1. F1 row `part` = conv5 stride 2 (x) + b_down
    if (part < H1) {
        float acc[H1];
#pragma unroll
        for (int j = 0; j < H1; ++j) acc[j] = bias;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int yy = 2 * part - 2 + dy;
            if (yy >= 0 && yy < H0) {
                float row[H0];
                read_xrow<TIO>(lx, yy, row);
                row_taps<2, H0, H1>(acc, row, w, dy);
            }
        }
        write_frow<H1>(lf1, part, acc);
    }
    __syncthreads();

    // ---- 2. F2 row `part` (8 rows: the first four waves) = conv5 stride 2 (F1) + b_down
    if (part < H2) {
        float acc[H2];
#pragma unroll
        for (int j = 0; j < H2; ++j) acc[j] = bias;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int yy = 2 * part - 2 + dy;
            if (yy >= 0 && yy < H1) {
                float row[H1];
                read_frow<H1>(lf1, yy, row);
                row_taps<2, H1, H2>(acc, row, w, dy);
            }
        }
        write_frow<H2>(lf2, part, acc);
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = wn[t];
    bias = biasn;
    __syncthreads();
    taps(wn, biasn, 2);                                                // convs[1] (phase 5)

    // ---- 3. C2 row `part` = conv5 (F2) + b_0
    if (part < H2) {
        float acc[H2];
#pragma unroll
        for (int j = 0; j < H2; ++j) acc[j] = bias;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int yy = part - 2 + dy;
            if (yy >= 0 && yy < H2) {
                float row[H2];
                read_frow<H2>(lf2, yy, row);
                row_taps<1, H2, H2>(acc, row, w, dy);
            }
        }
        write_frow<H2>(lc2, part, acc);
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = wn[t];
    bias = biasn;
    __syncthreads();
    taps(wn, biasn, 3);                                                // convs[2]: the final conv (phase 6)

    // ---- 4. T1 row `part` = F1 + resize(C2), in place
    if (part < H1) {
        float f[H1];
        read_frow<H1>(lf1, part, f);
        add_resized_row<MODE, H2>(lc2, part, H2, f);
        write_frow<H1>(lf1, part, f);
    }
    __syncthreads();

    // ---- 5. C1 row `part` = conv5 (T1) + b_1
    if (part < H1) {
        float acc[H1];
#pragma unroll
        for (int j = 0; j < H1; ++j) acc[j] = bias;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int yy = part - 2 + dy;
            if (yy >= 0 && yy < H1) {
                float row[H1];
                read_frow<H1>(lf1, yy, row);
                row_taps<1, H1, H1>(acc, row, w, dy);
            }
        }
        write_frow<H1>(lc1, part, acc);
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = wn[t];
    bias = biasn;
    __syncthreads();

    // ---- 6. y rows 2 part, 2 part + 1 = conv5 (x + resize(C1)) + b_2
    {
        float acc[RP][H0];
#pragma unroll
        for (int o = 0; o < RP; ++o)
#pragma unroll
            for (int j = 0; j < H0; ++j) acc[o][j] = bias;
#pragma unroll
        for (int tt = 0; tt < RP + 4; ++tt) {
            const int t = RP * part - 2 + tt;
            if (t >= 0 && t < H0) {
                float row[H0];
                read_xrow<TIO>(lx, t, row);
                add_resized_row<MODE, H1>(lc1, t, H1, row);
#pragma unroll
                for (int o = 0; o < RP; ++o)
                    if (tt - o >= 0 && tt - o < 5) row_taps<1, H0, H0>(acc[o], row, w, tt - o >= 0 && tt - o < 5 ? tt - o : 0);       // output row RP part + o: dy = tt - o
            }
        }
        const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc((void*)(y + ((size_t)n * H0 * H0) * C), 0, H0 * H0 * pix, 0x00020000);
#pragma unroll
        for (int o = 0; o < RP; ++o) {
            const unsigned vo = cvalid ? (unsigned)((RP * part + o) * H0 * pix + c * 2) : OOB;       // an invalid channel: out of range, not written
#pragma unroll
            for (int j = 0; j < H0; ++j) {
                TIO e[1];
                const float v1[1] = {acc[o][j]};
                store_vec<1>(e, v1);
                __builtin_amdgcn_raw_buffer_store_b16((short)__builtin_bit_cast(unsigned short, e[0]), ysrc, vo, j * pix, 0);
            }
        }
    }
}

template <typename TIO>
static hipError_t launch(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, hipStream_t s)
{
    const unsigned grid = (unsigned)N * (unsigned)((C + CB - 1) / CB);
    if (mode == 1) {
        auto kfn = k_recconv_p32<TIO, 1>;
        RCX_SET_LDS_ONCE(kfn, LDS_BYTES);
        RCX_LAUNCH_TIMED(kfn, dim3(grid), dim3(NT), LDS_BYTES, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C);
    } else {
        auto kfn = k_recconv_p32<TIO, 0>;
        RCX_SET_LDS_ONCE(kfn, LDS_BYTES);
        RCX_LAUNCH_TIMED(kfn, dim3(grid), dim3(NT), LDS_BYTES, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C);
    }
    return hipGetLastError();
}

}  // namespace p32

// never a function of N (a batch shard gives the same rows)
bool p32_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    const char* v = getenv("RCX_P32");
    const char* l = rcx::opt::value(rcx::opt::LANES);
    if ((v && *v == '0') || (l && *l == '0')) return false;
    return N >= 1 && C >= 1 && H == 32 && W == 32 && level == 2 && k == 5 && (dtype == 1 || dtype == 2) &&
           (long long)N * ((C + p32::CB - 1) / p32::CB) < (1ll << 31);
}

int p32_describe(int N, int C, int mode, char* buf, int len)
{
    return snprintf(buf, len, "p32(k_recconv_p32<%d>,cb=%d,nt=%d,blocks=%d,lds=%d)", mode, p32::CB, p32::NT, N * ((C + p32::CB - 1) / p32::CB), p32::LDS_BYTES);
}

hipError_t p32_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s)
{
    return dtype == 1 ? p32::launch<bf16_t>(x, y, wpack, bpack, N, C, mode, s) : p32::launch<f16_t>(x, y, wpack, bpack, N, C, mode, s);
}

}  // namespace rcx
