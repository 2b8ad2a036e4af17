// Shared pieces of the register-resident ("lanes") kernels: compile-time loops, DPP lane shifts, raw LDS element access,
// the 5x5 row-streaming convolutions, the resize weights and the in-register pyramid levels.
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

// active lanes of a group of LPC: 7 of 8 / 14 of 16 on the 7*2^k planes (the rest are EXEC-disabled guard lanes), all 16 on
// the 16*2^k planes (there a group fills a DPP row, whose boundary zero-fills)
constexpr int lanes_active(int w0, int lpc) { return (w0 % 16 == 0) ? 16 : (lpc == 8 ? 7 : 14); }

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

// ------------------------------------------------------------------------------------------------
// compile-time geometry
constexpr int down_size5(int h) { return (h + 4 - 5) / 2 + 1; }

struct VT { int i0, i1; float l; };
// vertical resize table entry, float arithmetic as ATen (and rcx_common.h bilinear_src / nearest_src)
constexpr VT vtab(int mode, int n_in, int n_out, int d)
{
    const float scale = (float)n_in / (float)n_out;
    if (mode == 1) {
        int i = (int)((float)d * scale);
        i = i < n_in - 1 ? i : n_in - 1;
        return VT{i, i, 0.f};
    }
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    return VT{i0, i0 + (i0 < n_in - 1 ? 1 : 0), src - (float)i0};
}

// ------------------------------------------------------------------------------------------------
// per-lane context
struct Ctx {
    int lane_in_group;       // 0 .. LPC-1
    int mode;                // 0 bilinear, 1 nearest
};

// row with two halo columns each side: ext[0]=col-2, ext[1]=col-1, ext[2..B+1]=own, ext[B+2], ext[B+3]
template <int LPC, int B, int D>
__device__ __forceinline__ void make_ext(const float (&row)[B], float (&ext)[B + 4])
{
#pragma unroll
    for (int j = 0; j < B; ++j) ext[2 + j] = row[j];
    if constexpr (B >= 2) {
        ext[1] = from_left<1, LPC>(row[B - 1]);
        ext[0] = from_left<1, LPC>(row[B - 2]);
        ext[B + 2] = from_right<1, LPC>(row[0]);
        ext[B + 3] = from_right<1, LPC>(row[1]);
    } else {
        ext[1] = from_left<D, LPC>(row[0]);
        ext[0] = from_left<D, LPC>(ext[1]);
        ext[3] = from_right<D, LPC>(row[0]);
        ext[4] = from_right<D, LPC>(ext[3]);
    }
}

// 5x5 depthwise, stride 1, pad 2.  in_row(IC<r>, float(&)[B]) yields input row r (called once per row, in
// order); out_row(IC<o>, const float(&)[B]) receives output row o as soon as it is complete.
template <int LPC, int H, int B, int D, class InRow, class OutRow>
__device__ __forceinline__ void conv5_s1(const float (&w)[25], float bias, InRow&& in_row, OutRow&& out_row)
{
    float acc[H][B];
    sfor<H>([&](auto R) RCX_INL {
        constexpr int r = decltype(R)::value;
        float row[B], ext[B + 4];
        in_row(R, row);
        make_ext<LPC, B, D>(row, ext);
        sfor<5>([&](auto U) RCX_INL {
            constexpr int u = decltype(U)::value;
            constexpr int o = r + 2 - u;
            if constexpr (o >= 0 && o < H) {
                constexpr bool first = (u == 0) || (r == 0);             // input row max(o-2, 0) is the first to reach output row o
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    float a = first ? bias : acc[o][j];
#pragma unroll
                    for (int v = 0; v < 5; ++v) a = fmaf(ext[j + v], w[u * 5 + v], a);
                    acc[o][j] = a;
                }
            }
        });
        if constexpr (r >= 2) out_row(IC<r - 2>{}, acc[r - 2]);
        RCX_ROW_FENCE;
        if constexpr (r == H - 1) {
            if constexpr (H >= 2) out_row(IC<H - 2>{}, acc[H - 2]);
            out_row(IC<H - 1>{}, acc[H - 1]);
        }
    });
}

// 5x5 depthwise, stride 2, pad 2: (HI, BI) -> (HO, BO).  BI >= 2: BO = BI/2, same lanes.  BI == 1: the result is
// valid in the lanes that are multiples of 2*D (horizontal stride-1 evaluation, every other lane is unused).
template <int LPC, int HI, int BI, int D, int HO, int BO, class InRow>
__device__ __forceinline__ void conv5_s2(const float (&w)[25], float bias, InRow&& in_row, float (&out)[HO][BO])
{
    sfor<HI>([&](auto R) RCX_INL {
        constexpr int r = decltype(R)::value;
        float row[BI], ext[BI + 4];
        in_row(R, row);
        make_ext<LPC, BI, D>(row, ext);
        sfor<5>([&](auto U) RCX_INL {
            constexpr int u = decltype(U)::value;
            constexpr int t = r + 2 - u;                                 // = 2 * o
            if constexpr (t >= 0 && (t % 2) == 0 && (t / 2) < HO) {
                constexpr int o = t / 2;
                constexpr bool is_first = (r == (2 * o - 2 > 0 ? 2 * o - 2 : 0));   // first input row that reaches output row o
#pragma unroll
                for (int i = 0; i < BO; ++i) {
                    float a = is_first ? bias : out[o][i];
#pragma unroll
                    for (int v = 0; v < 5; ++v) a = fmaf(ext[(BI >= 2 ? 2 * i : 0) + v], w[u * 5 + v], a);
                    out[o][i] = a;
                }
            }
        });
        RCX_ROW_FENCE;
    });
}

// horizontal resize weights of the compact 2x step: fine column j of the lane reads coarse columns
// cext[m + (j&1)] and cext[m + (j&1) + 1] (m = j/2, cext[0] = the lane's column -1)
template <int BC, int BF>
__device__ __forceinline__ void hweights_2x(const Ctx& c, int wc, int wf, float (&wt)[BF][2])
{
    const float scale = (float)wc / (float)wf;
#pragma unroll
    for (int j = 0; j < BF; ++j) {
        const int xf = c.lane_in_group * BF + j;
        int i0, i1;
        float lam;
        if (c.mode == 1) { i0 = i1 = nearest_src(xf, wc, scale); lam = 0.f; }
        else { Lerp s = bilinear_src(xf, wc, scale); i0 = s.i0; i1 = s.i1; lam = s.lam; }
        const int ca = c.lane_in_group * BC + j / 2 - 1 + (j & 1);
        wt[j][0] = (ca == i0 ? 1.f - lam : 0.f) + (ca == i1 ? lam : 0.f);
        wt[j][1] = (ca + 1 == i0 ? 1.f - lam : 0.f) + (ca + 1 == i1 ? lam : 0.f);
    }
}

// horizontal resize weights towards a B = 1 level (fine lane stride DF, coarse lane stride 2*DF):
// hrow = sum_{o=-2..2} wt[o+2] * value of the coarse row in lane (own + o*DF)
template <int DF>
__device__ __forceinline__ void hweights_off(const Ctx& c, int wc, int wf, float (&wt)[5])
{
    const float scale = (float)wc / (float)wf;
    const int xf = c.lane_in_group / DF;
    const bool lane_ok = (c.lane_in_group % DF) == 0 && xf < wf;
    int i0, i1;
    float lam;
    if (c.mode == 1) { i0 = i1 = nearest_src(xf, wc, scale); lam = 0.f; }
    else { Lerp s = bilinear_src(xf, wc, scale); i0 = s.i0; i1 = s.i1; lam = s.lam; }
#pragma unroll
    for (int o = -2; o <= 2; ++o) {
        const int t = xf + o;                                            // coarse column * 2
        const bool ok = lane_ok && t >= 0 && (t & 1) == 0;
        const int i = t >> 1;
        wt[o + 2] = ok ? ((i == i0 ? 1.f - lam : 0.f) + (i == i1 ? lam : 0.f)) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// taps of one conv from the workgroup's LDS copy [26][CBW] (25 taps + bias row); tl already points at the lane's channel.
// The empty asm makes the values opaque: otherwise the compiler re-reads a tap from LDS right before each use and
// every FMA group waits out an LDS round trip.
template <int CBW>
__device__ __forceinline__ void load_taps(const float* __restrict__ tl, float (&w)[25], float& b)
{
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = tl[t * CBW];
    b = tl[25 * CBW];
#pragma unroll
    for (int t = 0; t < 25; ++t) asm volatile("" : "+v"(w[t]));
    asm volatile("" : "+v"(b));
}

// one coarse row (BC columns per lane) resized horizontally to BF = 2*BC columns per lane
template <int LPC, int BC, int BF>
__device__ __forceinline__ void hresize_row(const float (&cr)[BC], const float (&wt)[BF][2], float (&out)[BF])
{
    float cext[BC + 2];
    cext[0] = from_left<1, LPC>(cr[BC - 1]);
#pragma unroll
    for (int i = 0; i < BC; ++i) cext[1 + i] = cr[i];
    cext[BC + 1] = from_right<1, LPC>(cr[0]);
#pragma unroll
    for (int j = 0; j < BF; ++j) out[j] = fmaf(wt[j][1], cext[j / 2 + (j & 1) + 1], wt[j][0] * cext[j / 2 + (j & 1)]);
}

// Level l of the pyramid.  run_io() is the general form: in_row(IC<r>, row) yields row r of F_l (it is called
// TWICE per row when l < LEVEL: once for the stride-2 conv, once to build T_l = F_l + resize(C_{l+1})), and
// out_row(IC<o>, row) receives C_l = conv_{LEVEL-l}(T_l) row by row.  run() is the all-in-registers wrapper
// used below level 0.  W is this level's width, (B, D) its lane layout.
template <int LPC, int MODE, int LVL, int LEVEL, int W, int B, int D, int CBW>
struct Level {
    static constexpr int H = W;
    static constexpr int WN = down_size5(W);
    static constexpr int BN = B >= 2 ? B / 2 : 1;
    static constexpr int DN = B >= 2 ? 1 : 2 * D;

    template <class InRow, class OutRow>
    static __device__ __forceinline__ void run_io(InRow&& in_row, OutRow&& out_row, const float* __restrict__ taps, const Ctx& c)
    {
        float w[25], b;
        if constexpr (LVL < LEVEL) {
            float Cn[WN][BN];
            {
                float Fn[WN][BN];
                load_taps<CBW>(taps, w, b);                                // conv 0 of the pack = the shared `down`
                conv5_s2<LPC, H, B, D, WN, BN>(w, b, in_row, Fn);
                Level<LPC, MODE, LVL + 1, LEVEL, WN, BN, DN, CBW>::run(Fn, Cn, taps, c);
            }
            float hrow[WN][B];
            hresize(Cn, hrow, c);
            load_taps<CBW>(taps + (1 + LEVEL - LVL) * 26 * CBW, w, b);
            conv5_s1<LPC, H, B, D>(w, b,
                [&](auto R, float (&row)[B]) RCX_INL {
                    in_row(R, row);
                    constexpr VT t = vtab(MODE, WN, H, decltype(R)::value);
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        if constexpr (MODE == 1 || t.i0 == t.i1) row[j] += hrow[t.i0][j];
                        else row[j] += fmaf(t.l, hrow[t.i1][j], (1.f - t.l) * hrow[t.i0][j]);
                    }
                },
                out_row);
        } else {
            load_taps<CBW>(taps + (1 + LEVEL - LVL) * 26 * CBW, w, b);
            conv5_s1<LPC, H, B, D>(w, b, in_row, out_row);
        }
    }

    static __device__ __forceinline__ void run(const float (&F)[H][B], float (&Cout)[H][B], const float* __restrict__ taps, const Ctx& c)
    {
        run_io(
            [&](auto R, float (&row)[B]) RCX_INL {
#pragma unroll
                for (int j = 0; j < B; ++j) row[j] = F[decltype(R)::value][j];
            },
            [&](auto O, const float (&acc)[B]) RCX_INL {
#pragma unroll
                for (int j = 0; j < B; ++j) Cout[decltype(O)::value][j] = acc[j];
            },
            taps, c);
    }

    // hrow = C_{l+1} resized horizontally to this level's columns (rows still coarse)
    static __device__ __forceinline__ void hresize(const float (&Cn)[WN][BN], float (&hrow)[WN][B], const Ctx& c)
    {
        if constexpr (B >= 2) {
            float wt[B][2];
            hweights_2x<BN, B>(c, WN, W, wt);
            sfor<WN>([&](auto R) RCX_INL { hresize_row<LPC, BN, B>(Cn[decltype(R)::value], wt, hrow[decltype(R)::value]); });
        } else {
            float wt[5];
            hweights_off<D>(c, WN, W, wt);
            sfor<WN>([&](auto R) RCX_INL {
                constexpr int r = decltype(R)::value;
                const float v = Cn[r][0];
                const float l1 = from_left<D, LPC>(v), l2 = from_left<D, LPC>(l1);
                const float r1 = from_right<D, LPC>(v), r2 = from_right<D, LPC>(r1);
                float a = wt[0] * l2;
                a = fmaf(wt[1], l1, a);
                a = fmaf(wt[2], v, a);
                a = fmaf(wt[3], r1, a);
                a = fmaf(wt[4], r2, a);
                hrow[r][0] = a;
            });
        }
    }
};

}  // namespace lanes
}  // namespace rcx
