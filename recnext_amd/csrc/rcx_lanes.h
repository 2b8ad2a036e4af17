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

#ifndef RCX_PK_FMA
#define RCX_PK_FMA 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <> struct Raw<float> {
    // x element into a register that is only ever loaded this way (see the bf16 form)
    static __device__ __forceinline__ void ld_into(const unsigned char* p, float& v) { v = *reinterpret_cast<const float*>(p); }
    template <int OFF> static __device__ __forceinline__ void ld_hi(const unsigned char* base, float& v) { v = *reinterpret_cast<const float*>(base + OFF); }
    template <int N> static __device__ __forceinline__ void settle(float (&)[N]) {}
    typedef float raw_t;
    static __device__ __forceinline__ raw_t ldr(const unsigned char* p) { return *reinterpret_cast<const float*>(p); }
    static __device__ __forceinline__ float cvt(raw_t r) { return r; }
    static __device__ __forceinline__ float ld(const unsigned char* p) { return *reinterpret_cast<const float*>(p); }
    static __device__ __forceinline__ void st(unsigned char* p, float v) { *reinterpret_cast<float*>(p) = v; }
};
template <> struct Raw<bf16_t> {
    // bf16 -> f32 without a VALU instruction: the 16 bits land in the upper half of v (ds_read_u16_d16_hi), whose lower
    // half must already be zero -- true for a register that was zeroed once and is only ever loaded through here
    static __device__ __forceinline__ void ld_into(const unsigned char* p, float& v)
    {
        u16x2 r = __builtin_bit_cast(u16x2, v);
        r.y = *reinterpret_cast<const bf16_t*>(p);
        v = __builtin_bit_cast(float, r);
    }
    // The same as one instruction the compiler does not select on gfx950 (its D16 loads do not preserve the other half under
    // SRAM-ECC; with the lower half already zero either behaviour gives the f32 value).  The compiler does not count an
    // LDS operation issued from inline asm, so settle() waits for it explicitly before the row is used; its own counts stay
    // safe because LDS operations complete in order (it can only wait longer than it thinks).
    template <int OFF> static __device__ __forceinline__ void ld_hi(const unsigned char* base, float& v)
    {
        static_assert(OFF >= 0 && OFF < 65536, "LDS instruction offset field");
        const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)base;
        asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
    }
    template <int N> static __device__ __forceinline__ void settle(float (&v)[N])
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < N; ++j) asm volatile("" : "+v"(v[j]));
    }
    typedef bf16_t raw_t;
    static __device__ __forceinline__ raw_t ldr(const unsigned char* p) { return *reinterpret_cast<const bf16_t*>(p); }
    static __device__ __forceinline__ float cvt(raw_t r) { return bf16_to_f32(r); }
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
typedef __attribute__((address_space(3))) float lds_float;     // explicit LDS pointers: a generic float* member would become flat_load

// per-lane context
struct Ctx {
    int lane_in_group;       // 0 .. LPC-1
    int mode;                // 0 bilinear, 1 nearest
    lds_float* xl[3];        // RCX_XCH_LDS: the lane's exchange lines for B = 1, 2, 4 columns per lane (lowest address it touches)
};

// ------------------------------------------------------------------------------------------------
// Neighbour exchange through LDS instead of DPP (compiled in with -DRCX_XCH_LDS=1; measured, not the default).
// On gfx950 ANY DPP/SDWA instruction drops the SIMD out of its 2-cycle VALU issue mode for the next ~100
// instructions (tools/ubench/xlane.hip, dpp_window.hip: 5 v_fmac + 1 v_mov_dpp run at 4.1 cycles per instruction
// however many waves are resident; the same stream with ds_swizzle or LDS reads instead runs at 2.2-2.7), so a kernel
// with one DPP move per dozen FMAs never leaves the slow mode.  Here a row goes through a per-channel line in LDS
// instead: the lane writes its B values and reads the neighbours' (one wave owns the whole channel and its LDS
// operations execute in order, so no barrier is involved).  Result on MI355X: bit-identical output, no DPP left, but
// the LDS pipe becomes the limiter (LDS busy 26 % -> 58 %, SQ_WAIT_INST_LDS x7.5) and the kernels are 5-12 % SLOWER
// (14x14: 32.3 vs 30.8 us, 28x28: 71 vs 62, 56x56: 157 vs 141), so the DPP form stays the default.  DESIGN.md 6.
// Layout of a channel's lines:
//   B = 1:  [XCH_PAD zeros][LPC slots][XCH_PAD zeros]
//   B >= 2: B planes of [zero][LPC slots][zero], plane i holds column i of every lane
// The zeros (and the slots of the EXEC-disabled guard lanes, which are never written) are the horizontal padding.
#ifndef RCX_XCH_LDS
#define RCX_XCH_LDS 0
#endif
constexpr int XCH_PAD = 8;                                   // zeros each side of the B = 1 line: 2 * the largest lane stride (4)
constexpr int xch_dmax(int b0, int level)
{
    int b = b0, d = 1;
    for (int l = 0; l < level; ++l) { if (b >= 2) b /= 2; else d *= 2; }
    return d;
}
constexpr int xch_pl(int lpc) { return lpc + 2; }
constexpr int xch_line1(int lpc) { return lpc + 2 * XCH_PAD; }
// floats per channel, padded so that the channels of a 32-lane half start on different banks
constexpr int xch_stride(int lpc, int b0)
{
    int n = xch_line1(lpc) + (b0 >= 2 ? 2 * xch_pl(lpc) : 0) + (b0 >= 4 ? 4 * xch_pl(lpc) : 0);
    while (n % 16 != lpc % 16 || n % 32 == 0) ++n;            // lpc 8: stride = 8 or 24 (mod 32); lpc 16: 16 (mod 32)
    return n;
}
// the lane's three line pointers inside the workgroup's exchange area; every access is at a non-negative immediate offset:
// B = 1 line: own slot at +XCH_PAD; planes: own slot at +1 (the left neighbour's at +0, the right one's at +2)
template <int LPC>
__device__ __forceinline__ void xch_setup(Ctx& c, float* area, int ch, int b0)
{
    lds_float* base = (lds_float*)area + ch * xch_stride(LPC, b0) + c.lane_in_group;
    c.xl[0] = base;
    c.xl[1] = base + xch_line1(LPC);
    c.xl[2] = c.xl[1] + 2 * xch_pl(LPC);
}
__device__ __forceinline__ void xch_put(lds_float* p, float v) { *reinterpret_cast<volatile lds_float*>(p) = v; }
__device__ __forceinline__ float xch_get(const lds_float* p) { return *reinterpret_cast<const volatile lds_float*>(p); }

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

// the same through the channel's LDS exchange line (RCX_XCH_LDS), else the DPP form above
template <int LPC, int B, int D>
__device__ __forceinline__ void make_ext(const float (&row)[B], float (&ext)[B + 4], const Ctx& c)
{
#if RCX_XCH_LDS
    if constexpr (RCX_XCH_LDS == 2 && B == 1) { make_ext<LPC, B, D>(row, ext); return; }   // hybrid: DPP below the B >= 2 levels
    static_assert(B == 1 || B == 2 || B == 4, "exchange lines exist for 1, 2 and 4 columns per lane");
    constexpr int PL = xch_pl(LPC);
#pragma unroll
    for (int j = 0; j < B; ++j) ext[2 + j] = row[j];
    if constexpr (B >= 2) {
        lds_float* p = c.xl[B == 2 ? 1 : 2];
#pragma unroll
        for (int i = 0; i < B; ++i) xch_put(p + i * PL + 1, row[i]);
        ext[0] = xch_get(p + (B - 2) * PL);
        ext[1] = xch_get(p + (B - 1) * PL);
        ext[B + 2] = xch_get(p + 2);
        ext[B + 3] = xch_get(p + PL + 2);
    } else {
        static_assert(2 * D <= XCH_PAD, "lane stride beyond the exchange line's padding");
        lds_float* p = c.xl[0] + XCH_PAD;
        xch_put(p, row[0]);
        ext[0] = xch_get(p - 2 * D);
        ext[1] = xch_get(p - D);
        ext[3] = xch_get(p + D);
        ext[4] = xch_get(p + 2 * D);
    }
#else
    make_ext<LPC, B, D>(row, ext);
#endif
}

// The 25 taps + bias of one conv as 15 aligned register pairs, three per tap row: (w0,w1) (w2,w3) (w4,-), the bias in the
// spare half of the last one.  v_pk_fma_f32 takes a tap splat out of either half of a pair through op_sel (two columns
// per instruction) or a whole pair (two taps per instruction); the scalar FMAs read the halves as plain registers.
struct Taps {
    f32x2 p[15];
    __device__ __forceinline__ float get(int u, int v) const { return (v & 1) ? p[3 * u + (v >> 1)].y : p[3 * u + (v >> 1)].x; }
    __device__ __forceinline__ f32x2 splat(int u, int v) const
    {
        const f32x2 q = p[3 * u + (v >> 1)];
        return (v & 1) ? __builtin_shufflevector(q, q, 1, 1) : __builtin_shufflevector(q, q, 0, 0);
    }
    __device__ __forceinline__ f32x2 pair(int u, int k) const { return p[3 * u + k]; }     // taps 2k, 2k+1 of row u (k = 0, 1)
    __device__ __forceinline__ float bias() const { return p[14].y; }
};

// taps of one conv from the workgroup's LDS copy [26][CBW] (25 taps + bias row); tl already points at the lane's channel.
// The empty asm makes the values opaque: otherwise the compiler re-reads a tap from LDS right before each use and
// every FMA group waits out an LDS round trip.
template <int CBW>
__device__ __forceinline__ void load_taps(const float* __restrict__ tl, Taps& w)
{
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        w.p[3 * u] = f32x2{tl[(5 * u) * CBW], tl[(5 * u + 1) * CBW]};
        w.p[3 * u + 1] = f32x2{tl[(5 * u + 2) * CBW], tl[(5 * u + 3) * CBW]};
        w.p[3 * u + 2] = f32x2{tl[(5 * u + 4) * CBW], u == 4 ? tl[25 * CBW] : 0.f};
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) asm volatile("" : "+v"(w.p[k]));
}

// 5x5 depthwise, stride 1, pad 2.  in_row(IC<r>, float(&)[B]) yields input row r (called once per row, in
// order); out_row(IC<o>, const float(&)[B]) receives output row o as soon as it is complete.
template <int LPC, int H, int B, int D, bool PK = true, class InRow, class OutRow>
__device__ __forceinline__ void conv5_s1(const Taps& w, InRow&& in_row, OutRow&& out_row, const Ctx& c)
{
    const float bias = w.bias();
    // B odd: two taps per v_pk_fma_f32 into two partial sums per output (taps 0,2,4 / 1,3), added when the row is complete
    constexpr bool TP = RCX_PK_FMA && PK && (B % 2) == 1;
    float acc[H][B];
    f32x2 acc2[TP ? H : 1][B];
    auto emit = [&](auto O) RCX_INL {
        constexpr int o = decltype(O)::value;
        if constexpr (TP) {
            float done[B];
#pragma unroll
            for (int j = 0; j < B; ++j) done[j] = acc2[o][j].x + acc2[o][j].y;
            out_row(O, done);
        } else {
            out_row(O, acc[o]);
        }
    };
#if RCX_XCH_LDS
    float ahead[B + 4];                                               // row r+1 goes through the exchange under row r's FMAs
    {
        float row[B];
        in_row(IC<0>{}, row);
        make_ext<LPC, B, D>(row, ahead, c);
    }
#endif
    sfor<H>([&](auto R) RCX_INL {
        constexpr int r = decltype(R)::value;
        float ext[B + 4];
#if RCX_XCH_LDS
#pragma unroll
        for (int k = 0; k < B + 4; ++k) ext[k] = ahead[k];
        if constexpr (r + 1 < H) {
            float row[B];
            in_row(IC<r + 1>{}, row);
            make_ext<LPC, B, D>(row, ahead, c);
        }
#else
        float row[B];
        in_row(R, row);
        make_ext<LPC, B, D>(row, ext, c);
#endif
        sfor<5>([&](auto U) RCX_INL {
            constexpr int u = decltype(U)::value;
            constexpr int o = r + 2 - u;
            if constexpr (o >= 0 && o < H) {
                constexpr bool first = (u == 0) || (r == 0);             // input row max(o-2, 0) is the first to reach output row o
                if constexpr (RCX_PK_FMA && PK && B % 2 == 0) {
                    // two adjacent columns per v_pk_fma_f32 (the tap is splat through op_sel): same products, same order of
                    // summation as the scalar form, half the instructions -- and a packed FMA costs the same 4 cycles as
                    // any other VALU instruction once a DPP move has put the SIMD into its slow issue mode
#pragma unroll
                    for (int q = 0; q < B / 2; ++q) {
                        f32x2 a = first ? f32x2{bias, bias} : f32x2{acc[o][2 * q], acc[o][2 * q + 1]};
#pragma unroll
                        for (int v = 0; v < 5; ++v)
                            a = __builtin_elementwise_fma(f32x2{ext[2 * q + v], ext[2 * q + v + 1]}, w.splat(u, v), a);
                        acc[o][2 * q] = a.x;
                        acc[o][2 * q + 1] = a.y;
                    }
                } else if constexpr (TP) {
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        f32x2 a = first ? f32x2{bias, 0.f} : acc2[o][j];
                        a = __builtin_elementwise_fma(f32x2{ext[j], ext[j + 1]}, w.pair(u, 0), a);
                        a = __builtin_elementwise_fma(f32x2{ext[j + 2], ext[j + 3]}, w.pair(u, 1), a);
                        a.x = fmaf(ext[j + 4], w.get(u, 4), a.x);
                        acc2[o][j] = a;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        float a = first ? bias : acc[o][j];
#pragma unroll
                        for (int v = 0; v < 5; ++v) a = fmaf(ext[j + v], w.get(u, v), a);
                        acc[o][j] = a;
                    }
                }
            }
        });
        if constexpr (r >= 2) emit(IC<r - 2>{});
        RCX_ROW_FENCE;
        if constexpr (r == H - 1) {
            if constexpr (H >= 2) emit(IC<H - 2>{});
            emit(IC<H - 1>{});
        }
    });
}

// 5x5 depthwise, stride 2, pad 2: (HI, BI) -> (HO, BO).  BI >= 2: BO = BI/2, same lanes.  BI == 1: the result is
// valid in the lanes that are multiples of 2*D (horizontal stride-1 evaluation, every other lane is unused).
template <int LPC, int HI, int BI, int D, int HO, int BO, bool PK = true, class InRow>
__device__ __forceinline__ void conv5_s2(const Taps& w, InRow&& in_row, float (&out)[HO][BO], const Ctx& c)
{
    const float bias = w.bias();
    // two taps per v_pk_fma_f32 into two partial sums per output (taps 0,2,4 / 1,3), added when the output row is complete
    constexpr bool TP = RCX_PK_FMA && PK;
    f32x2 out2[TP ? HO : 1][BO];
#if RCX_XCH_LDS
    float ahead[BI + 4];
    {
        float row[BI];
        in_row(IC<0>{}, row);
        make_ext<LPC, BI, D>(row, ahead, c);
    }
#endif
    sfor<HI>([&](auto R) RCX_INL {
        constexpr int r = decltype(R)::value;
        float ext[BI + 4];
#if RCX_XCH_LDS
#pragma unroll
        for (int k = 0; k < BI + 4; ++k) ext[k] = ahead[k];
        if constexpr (r + 1 < HI) {
            float row[BI];
            in_row(IC<r + 1>{}, row);
            make_ext<LPC, BI, D>(row, ahead, c);
        }
#else
        float row[BI];
        in_row(R, row);
        make_ext<LPC, BI, D>(row, ext, c);
#endif
        sfor<5>([&](auto U) RCX_INL {
            constexpr int u = decltype(U)::value;
            constexpr int t = r + 2 - u;                                 // = 2 * o
            if constexpr (t >= 0 && (t % 2) == 0 && (t / 2) < HO) {
                constexpr int o = t / 2;
                constexpr bool is_first = (r == (2 * o - 2 > 0 ? 2 * o - 2 : 0));   // first input row that reaches output row o
                constexpr bool is_last = (r == (2 * o + 2 < HI - 1 ? 2 * o + 2 : HI - 1));
#pragma unroll
                for (int i = 0; i < BO; ++i) {
                    const int e0 = BI >= 2 ? 2 * i : 0;
                    if constexpr (TP) {
                        f32x2 a = is_first ? f32x2{bias, 0.f} : out2[o][i];
                        a = __builtin_elementwise_fma(f32x2{ext[e0], ext[e0 + 1]}, w.pair(u, 0), a);
                        a = __builtin_elementwise_fma(f32x2{ext[e0 + 2], ext[e0 + 3]}, w.pair(u, 1), a);
                        a.x = fmaf(ext[e0 + 4], w.get(u, 4), a.x);
                        out2[o][i] = a;
                        if constexpr (is_last) out[o][i] = a.x + a.y;
                    } else {
                        float a = is_first ? bias : out[o][i];
#pragma unroll
                        for (int v = 0; v < 5; ++v) a = fmaf(ext[e0 + v], w.get(u, v), a);
                        out[o][i] = a;
                    }
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

template <int LPC, int BC, int BF>
__device__ __forceinline__ void hresize_row(const float (&cr)[BC], const float (&wt)[BF][2], float (&out)[BF], const Ctx& c)
{
#if RCX_XCH_LDS
    if constexpr (RCX_XCH_LDS == 2 && BF == 1) { hresize_row<LPC, BC, BF>(cr, wt, out); return; }
    static_assert(BC == 1 || BC == 2 || BC == 4, "exchange lines exist for 1, 2 and 4 columns per lane");
    constexpr int PL = xch_pl(LPC);
    float cext[BC + 2];
#pragma unroll
    for (int i = 0; i < BC; ++i) cext[1 + i] = cr[i];
    if constexpr (BC == 1) {
        lds_float* p = c.xl[0] + XCH_PAD;
        xch_put(p, cr[0]);
        cext[0] = xch_get(p - 1);
        cext[2] = xch_get(p + 1);
    } else {
        lds_float* p = c.xl[BC == 2 ? 1 : 2];
        xch_put(p + 1, cr[0]);
        xch_put(p + (BC - 1) * PL + 1, cr[BC - 1]);
        cext[0] = xch_get(p + (BC - 1) * PL);
        cext[BC + 1] = xch_get(p + 2);
    }
#pragma unroll
    for (int j = 0; j < BF; ++j) out[j] = fmaf(wt[j][1], cext[j / 2 + (j & 1) + 1], wt[j][0] * cext[j / 2 + (j & 1)]);
#else
    hresize_row<LPC, BC, BF>(cr, wt, out);
#endif
}

// Level l of the pyramid.  run_io() is the general form: in_row(IC<r>, row) yields row r of F_l (it is called
// TWICE per row when l < LEVEL: once for the stride-2 conv, once to build T_l = F_l + resize(C_{l+1})), and
// out_row(IC<o>, row) receives C_l = conv_{LEVEL-l}(T_l) row by row.  run() is the all-in-registers wrapper
// used below level 0.  W is this level's width, (B, D) its lane layout.
template <int LPC, int MODE, int LVL, int LEVEL, int W, int B, int D, int CBW, bool PK = true>
struct Level {
    static constexpr int H = W;
    static constexpr int WN = down_size5(W);
    static constexpr int BN = B >= 2 ? B / 2 : 1;
    static constexpr int DN = B >= 2 ? 1 : 2 * D;

    template <class InRow, class OutRow>
    static __device__ __forceinline__ void run_io(InRow&& in_row, OutRow&& out_row, const float* __restrict__ taps, const Ctx& c)
    {
        Taps w;
        if constexpr (LVL < LEVEL) {
            float Cn[WN][BN];
            {
                float Fn[WN][BN];
                load_taps<CBW>(taps, w);                                   // conv 0 of the pack = the shared `down`
                conv5_s2<LPC, H, B, D, WN, BN, PK>(w, in_row, Fn, c);
                Level<LPC, MODE, LVL + 1, LEVEL, WN, BN, DN, CBW, PK>::run(Fn, Cn, taps, c);
            }
            float hrow[WN][B];
            hresize(Cn, hrow, c);
            load_taps<CBW>(taps + (1 + LEVEL - LVL) * 26 * CBW, w);
            conv5_s1<LPC, H, B, D, PK>(w,
                [&](auto R, float (&row)[B]) RCX_INL {
                    in_row(R, row);
                    constexpr VT t = vtab(MODE, WN, H, decltype(R)::value);
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        if constexpr (MODE == 1 || t.i0 == t.i1) row[j] += hrow[t.i0][j];
                        else row[j] += fmaf(t.l, hrow[t.i1][j], (1.f - t.l) * hrow[t.i0][j]);
                    }
                },
                out_row, c);
        } else {
            load_taps<CBW>(taps + (1 + LEVEL - LVL) * 26 * CBW, w);
            conv5_s1<LPC, H, B, D, PK>(w, in_row, out_row, c);
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
            sfor<WN>([&](auto R) RCX_INL { hresize_row<LPC, BN, B>(Cn[decltype(R)::value], wt, hrow[decltype(R)::value], c); });
        } else {
            float wt[5];
            hweights_off<D>(c, WN, W, wt);
            sfor<WN>([&](auto R) RCX_INL {
                constexpr int r = decltype(R)::value;
                const float v = Cn[r][0];
#if RCX_XCH_LDS == 1
                static_assert(2 * D <= XCH_PAD, "lane stride beyond the exchange line's padding");
                lds_float* p = c.xl[0] + XCH_PAD;
                xch_put(p, v);
                const float l2 = xch_get(p - 2 * D), l1 = xch_get(p - D);
                const float r1 = xch_get(p + D), r2 = xch_get(p + 2 * D);
#else
                const float l1 = from_left<D, LPC>(v), l2 = from_left<D, LPC>(l1);
                const float r1 = from_right<D, LPC>(v), r2 = from_right<D, LPC>(r1);
#endif
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
