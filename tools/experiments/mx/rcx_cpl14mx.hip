// Matrix-core RecConv2d for the 14x14 / level 2 block (13 of RecNeXt-M3's 21 token mixers; model/recnext.py:24-34), 16-bit activations
// whose taps may be rounded to the activations' type (rcx_recconv2d_fwd_mx; round 3).
//
// Same idea as rcx_cpl14.hip -- a lane owns a whole (image, channel) plane, every level lives in its registers, nothing goes through LDS
// memory -- but all five 5x5 convs run as banded 4 x 4 x 4 products on the matrix cores (v_mfma_f32_4x4x4_16b_bf16 / _f16; operand maps
// and Toeplitz blocks: rcx_cpt_kernel.h, "matrix-core variant"; the A operands come ready-made from the matrix pack):
//   block = channel, the four lanes of a block = FOUR IMAGES (n .. n+3) of that channel; a wave = 16 channels x 4 images = 64 planes;
//   a row of a plane is held as "K blocks" of four adjacent pixels (two registers of two 16-bit values); an output block of four columns
//   takes two (stride 1) or three (stride 2) K blocks per tap row; K blocks that lie wholly in the zero padding are not issued.
// Lane maps: the matrix instruction wants M (lane = 4 * channel + image).  x comes in with 16-byte loads (a lane = 8 channels of one pixel
// of one image, four images per lane: 28 instructions for the wave's 25 KB, all in flight at once -- the first build's 196 two-byte
// loads per lane filled the 6-bit memory counter four rows at a time and the wave spent 7 us waiting for them), is interleaved in
// registers (16 v_perm_b32 per lane and step) into an LDS image [pixel][channel][image], and four adjacent pixels of a channel of all
// four images then are ONE ds_read_b64_tr_b16 that lands in map M as it is.  y leaves from map A (lane = 16 * image + channel: 16
// consecutive lanes = 32 contiguous bytes of a pixel) through 7 ds_bpermute_b32 per row.
// The four waves of a workgroup are four consecutive channel groups of the same four images: they read the four quarters of the same
// 128-byte lines at about the same time.
// Numerics: float32 accumulation; the conv INPUTS are rounded to the activations' type (x is exact; F1, F2, F1 + resize(C2) and
// x + resize(C1) once each) -- fewer roundings than the reference's own 16-bit run, which rounds after every operator (:27-34).
// Against the vector kernel (k_recconv_cpl14: 3 780 FMA instructions of 4 677 vector instructions per wave, 18.6 us at 256 x 256):
// ~ 800 matrix instructions + ~ 1 900 vector instructions, 19.0 us -- one dependent chain per SIMD either way.
// A band-split variant (the four lanes of a block = four row bands of ONE plane, every plane in the wave's LDS image as 8-byte K blocks,
// no barrier, 4 096 one-wave workgroups) was built and measured in round 3 and is not kept: a wave alone took 6.6 us instead of 14 (as
// designed), the full problem 36.6 us -- 27 KB of LDS per wave leave 1.5 waves per SIMD, and every level is an LDS write -> read ->
// matrix -> convert -> write chain that nothing hides (profiles/r03_mx14b_band_split.txt).
#include "rcx_cpt_kernel_mx.h"
#include "rcx_opts.h"

namespace rcx {
namespace mx14 {

using cpt::f32x4;
using cpt::i32x4;
using cpt::mx444;
using cpt::MxTaps;
using cpt::pk16;
using cpt::u32x2;
using lanes::IC;
using lanes::sfor;
using lanes::vtab;
using lanes::VT;

// a packed pair of 16-bit values as two float32
template <typename TIO> __device__ __forceinline__ float lo_f32(uint32_t p)
{
    if constexpr (std::is_same<TIO, f16_t>::value) return (float)__builtin_bit_cast(_Float16, (uint16_t)(p & 0xffffu));
    else return __uint_as_float(p << 16);
}
template <typename TIO> __device__ __forceinline__ float hi_f32(uint32_t p)
{
    if constexpr (std::is_same<TIO, f16_t>::value) return (float)__builtin_bit_cast(_Float16, (uint16_t)(p >> 16));
    else return __uint_as_float(p & 0xffff0000u);
}

// The K blocks of a row of N <= 7 pixels (columns -2 .. 9; N = 4: columns -2 .. 5) given as float32, rounded to TIO: K0 = (0, 0, c0, c1), K1 =
// (c2 .. c5), K2 = (c6, 0, 0, 0); columns >= N are zero (what a product's unused fourth output column left in a register is dropped here)
template <typename TIO, int N>
__device__ __forceinline__ void kblocks_small(const float (&c)[8], u32x2 (&K)[3])
{
    auto at = [&](int i) -> float { return i < N ? c[i] : 0.f; };
    K[0] = u32x2{0u, pk16<TIO>(at(0), at(1))};
    K[1] = u32x2{pk16<TIO>(at(2), at(3)), N > 4 ? pk16<TIO>(at(4), at(5)) : 0u};
    K[2] = u32x2{N > 6 ? pk16<TIO>(at(6), 0.f) : 0u, 0u};
}

constexpr int WLDS = 198 * 128;                                  // bytes of LDS per wave

template <int MODE, typename TIO>
__global__ __launch_bounds__(256, 1)
void k_recconv_mx14(const TIO* __restrict__ x, TIO* __restrict__ y, const void* __restrict__ mxpack, const float* __restrict__ bpack,
                    int N, int C, int has_bias)
{
    constexpr int W = 14, W1 = 7, W2 = 4;
    const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int ng = (C + 15) / 16, nq = (N + 3) / 4;
    const int unit = (int)blockIdx.x * 4 + wave;
    if (unit >= nq * ng) return;                                  // whole waves only: every remaining lane runs the matrix instructions
    const int iq = unit / ng, cg = unit - iq * ng;
    const int pix = C * 2;                                        // bytes between horizontally adjacent pixels
    // lane maps (see the head of the file)
    const int nA = 4 * iq + (lane >> 4), cA = 16 * cg + (lane & 15);
    const bool validA = nA < N && cA < C;
    const unsigned OOB = 0x80000000u;
    const unsigned voA = validA ? (unsigned)nA * (unsigned)(W * W * pix) + (unsigned)cA * 2u : OOB;   // out of range: loads read 0, stores are dropped
    const int cM = 16 * cg + (lane >> 2), ccM = cM < C ? cM : C - 1;
    const int permMA = 4 * (4 * (lane & 15) + (lane >> 4));      // ds_bpermute address: lane (A) takes its value from lane (M)
    i32x4 ysrc;
    {
        const unsigned long long b = (unsigned long long)y;
        const long long bytes = (long long)N * W * W * pix;
        ysrc.x = (int)(unsigned)b; ysrc.y = (int)(unsigned)(b >> 32) & 0xffff; ysrc.z = (int)bytes; ysrc.w = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc((void*)mxpack, 0, 4 * 5 * cpt::MXP_SLOTS * 4 * C * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrc_ = __builtin_amdgcn_make_buffer_rsrc((void*)bpack, 0, has_bias ? 4 * C * 4 : 0, 0x00020000);   // zero records without a bias

    // ---- x -> LDS image of this wave: [pixel 0 .. 197][channel 0 .. 15][image 0 .. 3] of 16-bit values (128 bytes per pixel; pixels 196, 197
    // are only ever read into registers nobody uses)
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];
    unsigned char* const wl = xlds + wave * WLDS;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long long)N * W * W * pix), 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    sfor<7>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int slot = 64 * i + lane, px = slot >> 1, oct = slot & 1;
        const bool on = slot < 2 * W * W;
        u32x4 L[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = 4 * iq + e;
            const unsigned vo = (on && n < N) ? (unsigned)(n * (W * W) + px) * (unsigned)pix + (unsigned)((16 * cg + 8 * oct) * 2) : OOB;
            L[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)vo, 0, 0));
        }
        if (on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // channels 2k, 2k + 1 of the octet: (image 0, 1) (image 2, 3) each
                u32x4 o;
                o.x = __builtin_amdgcn_perm(L[1][k], L[0][k], 0x05040100u);
                o.y = __builtin_amdgcn_perm(L[3][k], L[2][k], 0x05040100u);
                o.z = __builtin_amdgcn_perm(L[1][k], L[0][k], 0x07060302u);
                o.w = __builtin_amdgcn_perm(L[3][k], L[2][k], 0x07060302u);
                *reinterpret_cast<u32x4*>(wl + px * 128 + oct * 64 + k * 16) = o;
            }
        }
    });
    // a wave's LDS operations execute in order: its reads below see its writes above (no barrier: the image is this wave's own)
    uint32_t Xp[W][7];                                            // x as pairs (columns 2k, 2k+1), map M: kept for the final conv
    // lane 4 q + p of a 16-lane group g supplies row q (pixel) and 8-byte chunk p (channel 4 g + p, four images) of the transposed block
    const unsigned char* const trp = wl + ((lane >> 2) & 3) * 128 + (4 * (lane >> 4) + (lane & 3)) * 8;
    MxTaps<TIO, 2> ad;
    cpt::load_mxtaps(ad, msrc, bsrc_, 0, C, ccM, lane & 3);
    f32x4 F1[W1][2];                                              // columns 0 .. 7 of each row (column 7 is not a pixel)
    {
        f32x4 facc[3][2];
        const f32x4 b4 = f32x4{ad.bias, ad.bias, ad.bias, ad.bias};
        sfor<W>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                        // pixels 4k .. 4k+3 of the row (14, 15 belong to the next row: not used)
                const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cpt::s16x4*)(trp + (W * r + 4 * k) * 128)));
                Xp[r][2 * k] = v.x;
                if (k < 3) Xp[r][2 * k + 1] = v.y;
            }
            // K blocks of the row: columns -2 .. 17 (K4 = columns 14 .. 17 is zero: not issued)
            const u32x2 B[4] = {u32x2{0u, Xp[r][0]}, u32x2{Xp[r][1], Xp[r][2]}, u32x2{Xp[r][3], Xp[r][4]}, u32x2{Xp[r][5], Xp[r][6]}};
#pragma unroll
            for (int o = 0; o < W1; ++o) {
                const int u = r - 2 * o + 2;
                if (u < 0 || u > 4) continue;
                const bool first = u == 0 || r == 0;             // the first input row that reaches output row o carries the bias in
                f32x4(&a)[2] = facc[o % 3];
#pragma unroll
                for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (2 * m + kb > 3) continue;            // K4
                        a[m] = mx444<TIO>(ad.a[u][kb], B[2 * m + kb], (first && kb == 0) ? b4 : a[m]);
                    }
                if (u == 4 || r == W - 1) { F1[o][0] = a[0]; F1[o][1] = a[1]; }
            }
        });
    }

    // ---- the level-1 block on the 7x7 plane: C1 = conv_1(F1 + resize(conv_0(down(F1))))                          (:27-33)
    f32x4 C1[W1][2];
    {
        // F2 = down(F1): 7 -> 4, one output block, K blocks 0 .. 2 of the F1 rows
        f32x4 F2[W2];
        {
            const f32x4 b4 = f32x4{ad.bias, ad.bias, ad.bias, ad.bias};
            u32x2 K[W1][3];
#pragma unroll
            for (int r = 0; r < W1; ++r) {
                const float c[8] = {F1[r][0][0], F1[r][0][1], F1[r][0][2], F1[r][0][3], F1[r][1][0], F1[r][1][1], F1[r][1][2], 0.f};
                kblocks_small<TIO, W1>(c, K[r]);
            }
#pragma unroll
            for (int o = 0; o < W2; ++o) {
                bool first = true;
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int r = 2 * o + u - 2;
                    if (r < 0 || r >= W1) continue;
#pragma unroll
                    for (int kb = 0; kb < 3; ++kb) { F2[o] = mx444<TIO>(ad.a[u][kb], K[r][kb], first ? b4 : F2[o]); first = false; }
                }
            }
        }
        // C2 = conv_0(F2) on 4x4: K blocks 0, 1 (columns -2 .. 5)
        f32x4 C2[W2];
        {
            MxTaps<TIO, 1> a0;
            cpt::load_mxtaps(a0, msrc, bsrc_, 1, C, ccM, lane & 3);
            const f32x4 b4 = f32x4{a0.bias, a0.bias, a0.bias, a0.bias};
            u32x2 K[W2][3];
#pragma unroll
            for (int r = 0; r < W2; ++r) {
                const float c[8] = {F2[r][0], F2[r][1], F2[r][2], F2[r][3], 0.f, 0.f, 0.f, 0.f};
                kblocks_small<TIO, W2>(c, K[r]);
            }
#pragma unroll
            for (int o = 0; o < W2; ++o) {
                bool first = true;
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int r = o + u - 2;
                    if (r < 0 || r >= W2) continue;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) { C2[o] = mx444<TIO>(a0.a[u][kb], K[r][kb], first ? b4 : C2[o]); first = false; }
                }
            }
        }
        // T1 = F1 + resize(C2): 4 -> 7 with ATen's index arithmetic (vtab), rows of C2 resized horizontally first
        float T1[W1][8];
        {
            float H[W2][W1];
#pragma unroll
            for (int i = 0; i < W2; ++i)
#pragma unroll
                for (int q = 0; q < W1; ++q) {
                    const VT t = vtab(MODE, W2, W1, q);
                    H[i][q] = (MODE == 1 || t.i0 == t.i1) ? C2[i][t.i0] : fmaf(t.l, C2[i][t.i1], (1.f - t.l) * C2[i][t.i0]);
                }
#pragma unroll
            for (int r = 0; r < W1; ++r) {
                const VT t = vtab(MODE, W2, W1, r);
#pragma unroll
                for (int q = 0; q < W1; ++q) {
                    const float f = q < 4 ? F1[r][0][q] : F1[r][1][q - 4];
                    T1[r][q] = (MODE == 1 || t.i0 == t.i1) ? f + H[t.i0][q] : fmaf(t.l, H[t.i1][q], fmaf(1.f - t.l, H[t.i0][q], f));
                }
                T1[r][7] = 0.f;
            }
        }
        // C1 = conv_1(T1) on 7x7: two output blocks, K blocks m, m + 1
        {
            MxTaps<TIO, 1> a1;
            cpt::load_mxtaps(a1, msrc, bsrc_, 2, C, ccM, lane & 3);
            const f32x4 b4 = f32x4{a1.bias, a1.bias, a1.bias, a1.bias};
            u32x2 K[W1][3];
#pragma unroll
            for (int r = 0; r < W1; ++r) kblocks_small<TIO, W1>(T1[r], K[r]);
#pragma unroll
            for (int o = 0; o < W1; ++o) {
                bool first = true;
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int r = o + u - 2;
                    if (r < 0 || r >= W1) continue;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int m = 0; m < 2; ++m) C1[o][m] = mx444<TIO>(a1.a[u][kb], K[r][m + kb], (first && kb == 0) ? b4 : C1[o][m]);
                    first = false;
                }
            }
        }
    }

    // ---- y = conv_2(x + resize(C1)): input-row stationary, five accumulator rows in flight                         (:34)
    {
        MxTaps<TIO, 1> a2;
        cpt::load_mxtaps(a2, msrc, bsrc_, 3, C, ccM, lane & 3);
        const f32x4 b4 = f32x4{a2.bias, a2.bias, a2.bias, a2.bias};
        float H1[W1][W];                                          // C1 rows resized horizontally, each computed just before its first use
        f32x4 acc[5][4];
        sfor<W>([&](auto tc_) {
            constexpr int t = decltype(tc_)::value;
            constexpr VT vt = vtab(MODE, W1, W, t);
            sfor<W1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr bool used = i == vt.i0 || i == vt.i1;
                constexpr bool before = t > 0 && (i == vtab(MODE, W1, W, t > 0 ? t - 1 : 0).i0 || i == vtab(MODE, W1, W, t > 0 ? t - 1 : 0).i1);
                if constexpr (used && !before) {
#pragma unroll
                    for (int q = 0; q < W; ++q) {
                        const VT h = vtab(MODE, W1, W, q);
                        const float c0 = h.i0 < 4 ? C1[i][0][h.i0] : C1[i][1][h.i0 - 4], c1 = h.i1 < 4 ? C1[i][0][h.i1] : C1[i][1][h.i1 - 4];
                        H1[i][q] = (MODE == 1 || h.i0 == h.i1) ? c0 : fmaf(h.l, c1, (1.f - h.l) * c0);
                    }
                }
            });
            // T0 row t, rounded to pairs; K blocks: columns -2 .. 17 (K4 is zero)
            uint32_t p[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const float x0 = lo_f32<TIO>(Xp[t][k]), x1 = hi_f32<TIO>(Xp[t][k]);
                float t0, t1;
                if (MODE == 1 || vt.i0 == vt.i1) { t0 = x0 + H1[vt.i0][2 * k]; t1 = x1 + H1[vt.i0][2 * k + 1]; }
                else {
                    t0 = fmaf(vt.l, H1[vt.i1][2 * k], fmaf(1.f - vt.l, H1[vt.i0][2 * k], x0));
                    t1 = fmaf(vt.l, H1[vt.i1][2 * k + 1], fmaf(1.f - vt.l, H1[vt.i0][2 * k + 1], x1));
                }
                p[k] = pk16<TIO>(t0, t1);
            }
            const u32x2 B[4] = {u32x2{0u, p[0]}, u32x2{p[1], p[2]}, u32x2{p[3], p[4]}, u32x2{p[5], p[6]}};
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int o = t - u + 2;
                if (o < 0 || o >= W) continue;
                const bool first = u == 0 || t == 0;             // the first input row that reaches output row o carries the bias in
                f32x4(&a)[4] = acc[o % 5];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (m + kb > 3) continue;                // K4
                        a[m] = mx444<TIO>(a2.a[u][kb], B[m + kb], (first && kb == 0) ? b4 : a[m]);
                    }
            }
            // output rows that have seen their last input row: t - 2, and at the bottom edge the last two
            sfor<3>([&](auto dc) {
                constexpr int o = t - 2 + decltype(dc)::value;
                if constexpr (o >= 0 && o < W && (decltype(dc)::value == 0 || t == W - 1)) {
                    uint32_t q7[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j)
                        q7[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(permMA, (int)pk16<TIO>(acc[o % 5][j >> 1][2 * (j & 1)], acc[o % 5][j >> 1][2 * (j & 1) + 1]));
                    cpt::RowSt<TIO, 0>::st_packed(q7, voA, ysrc, o * W * pix, pix);
                }
            });
        });
    }
}

template <int MODE, typename TIO>
static hipError_t launch(const void* x, void* y, const void* mxpack, const float* bpack, int N, int C, hipStream_t s)
{
    const int units = ((N + 3) / 4) * ((C + 15) / 16);
    auto kfn = k_recconv_mx14<MODE, TIO>;
    RCX_SET_LDS_ONCE(kfn, 4 * WLDS);
    hipLaunchKernelGGL(kfn, dim3((units + 3) / 4), dim3(256), 4 * WLDS, s, (const TIO*)x, (TIO*)y, mxpack, bpack, N, C, bpack != nullptr);
    return hipGetLastError();
}

}  // namespace mx14

// RCX_CPL14_MX=0: off (A/B)
bool cpl14mx_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    const char* v = rcx::opt::value(rcx::opt::CPL14_MX);
    if (v && *v == '0') return false;
    const char* l = rcx::opt::value(rcx::opt::LANES);
    if (l && *l == '0') return false;
    // 32-bit byte offsets inside the activation buffer
    // C % 8: 16-byte loads of eight channels; 32-bit byte offsets inside the activation buffer
    return k == 5 && H == 14 && W == 14 && level == 2 && (dtype == 1 || dtype == 2) && C >= 8 && C % 8 == 0 && (long long)N * 196 * C * 2 < (1ll << 31);
}

int cpl14mx_describe(int N, int C, int mode, int dtype, char* buf, int len)
{
    return snprintf(buf, len, "cpl14_mx(k_recconv_mx14<%d, %s>,units=%d,nt=256)", mode, dtype == 2 ? "f16" : "bf16", ((N + 3) / 4) * ((C + 15) / 16));
}

hipError_t cpl14mx_recconv(const void* x, void* y, const void* mxpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s)
{
    if (dtype == 1) return mode == 1 ? mx14::launch<1, bf16_t>(x, y, mxpack, bpack, N, C, s) : mx14::launch<0, bf16_t>(x, y, mxpack, bpack, N, C, s);
    if (dtype == 2) return mode == 1 ? mx14::launch<1, f16_t>(x, y, mxpack, bpack, N, C, s) : mx14::launch<0, f16_t>(x, y, mxpack, bpack, N, C, s);
    return hipErrorInvalidConfiguration;
}

}  // namespace rcx
