// Building blocks of the channel-per-lane kernels (rcx_cpl14.hip: forward; rcx_cplbwd.hip: backward): one lane owns a whole
// (image, channel) plane held as float32 pairs of horizontally adjacent pixels.  See rcx_cpl14.hip for the layout's rationale.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "rcx_common.h"
#include "rcx_lanes.h"
#include "rcx_launch.h"

namespace rcx {
namespace cpl14 {

using lanes::f32x2;
using lanes::vtab;
using lanes::VT;

#define RCX_FENCE __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
// (a.y, b.x): the pair that starts one pixel to the right of a -- one v_pk_mov_b32
__device__ __forceinline__ f32x2 shift1(f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 1, 2); }

// ---- addressing: uniform base in an SGPR pair (made opaque, so that the compiler keeps "scalar base + lane offset + immediate"
// and does not re-associate towards one vector base plus vector adds) + this lane's byte offset
typedef const __attribute__((address_space(1))) char* gcptr;        // explicit global address space: global_*, never flat_*
typedef __attribute__((address_space(1))) char* gptr;
__device__ __forceinline__ gcptr opaque(gcptr p) { asm volatile("" : "+s"(p)); return p; }
template <typename T> __device__ __forceinline__ T gload(gcptr p) { return *reinterpret_cast<const __attribute__((address_space(1))) T*>(p); }
// A load that stays where it is written: a relaxed wavefront-scope atomic load is the same global_load instruction (no cache
// bits at this scope) but, being ordered, is neither sunk towards its first use nor hoisted -- the x rows must be requested
// AHEAD rows before they are touched, and the compiler still counts it in its own s_waitcnt bookkeeping.
template <typename T> __device__ __forceinline__ T gload_here(gcptr p)
{
    return __hip_atomic_load(reinterpret_cast<const __attribute__((address_space(1))) T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <typename T> __device__ __forceinline__ void gstore(gcptr p, T v) { *(__attribute__((address_space(1))) T*)(p) = v; }

// Pins.  sched_barrier orders only what the machine scheduler sees; the instruction selector before it is free to float pure
// arithmetic across the barrier (and it does: without pins the FMAs of all fourteen rows sink below the loads, conversions and
// stash moves of all fourteen rows, and 196 converted values are live at once).  An empty volatile asm that reads and writes
// a value is ordered against the barriers and ties the value's producers above it and its consumers below it: no instruction.
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
template <int A> __device__ __forceinline__ void pin(f32x2 (&v)[A]) {
#pragma unroll
    for (int i = 0; i < A; ++i) pin(v[i]);
}
template <int A, int B> __device__ __forceinline__ void pin(f32x2 (&v)[A][B]) {
#pragma unroll
    for (int i = 0; i < A; ++i) pin(v[i]);
}

// first touch of a row of hand-issued loads: wait until at most PENDING younger memory operations are outstanding
template <int PENDING>
__device__ __forceinline__ void pin_row(uint32_t (&v)[14])
{
    asm volatile("s_waitcnt vmcnt(%14)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]) : "n"(PENDING));
}
template <int PENDING>
__device__ __forceinline__ void pin_row(uint32_t (&v)[7])
{
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) : "n"(PENDING));
}

// AGPR stash: the accumulator half of the unified register file holds x between its two uses (one VALU move each way)
// (volatile: they stay in the row they are written in)
__device__ __forceinline__ float stash(float v) { float a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ float unstash(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

// Row access.  CT > 0: the channel count is a compile-time constant, a row needs one scalar base (two for float32 I/O) and the
// columns are immediates; CT == 0: one scalar base per row and 14 per-lane column offsets computed once.
template <int W_, int CT, typename TIO>
struct RowAddr {
    static constexpr int W = W_;
    static constexpr int PIXB = CT * (int)sizeof(TIO);
    static constexpr int GROUP = CT > 0 ? ((8192 / (PIXB > 0 ? PIXB : 1) >= W) ? W : 8192 / (PIXB > 0 ? PIXB : 1)) : W;   // columns per base
    unsigned col[CT > 0 ? 1 : W];                                                             // CT == 0: vo + q * pix
    size_t pix;
    __device__ __forceinline__ RowAddr(unsigned vo, size_t pix_) : pix(pix_)
    {
        if constexpr (CT > 0) {
            static_assert(GROUP >= 2 && (GROUP % 2 == 0 || GROUP >= W), "channel count too large for immediate addressing");
            col[0] = vo;
        } else {
#pragma unroll
            for (int q = 0; q < W; ++q) col[q] = vo + (unsigned)q * (unsigned)pix_;
        }
    }
    // f(IC<q>, base, voff, IC<imm>): element (r, q) of this lane lives at base (uniform) + voff (this lane's 32-bit offset) + imm
    template <class F>
    __device__ __forceinline__ void row(gcptr plane, int r, F&& f) const
    {
        if constexpr (CT > 0) {
            lanes::sfor<(W + GROUP - 1) / GROUP>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int q0 = g * GROUP, q1 = q0 + GROUP < W ? q0 + GROUP : W, mid = q0 + GROUP / 2;
                const gcptr base = opaque(plane + (size_t)(r * W + mid) * PIXB);
                lanes::sfor<q1 - q0>([&](auto qc) {
                    constexpr int q = q0 + decltype(qc)::value;
                    f(lanes::IC<q>{}, base, col[0], lanes::IC<(q - mid) * PIXB>{});
                });
            });
        } else {
            const gcptr base = opaque(plane + (size_t)(r * W) * pix);
            lanes::sfor<W>([&](auto qc) { f(qc, base, col[decltype(qc)::value], lanes::IC<0>{}); });
        }
    }
};

// x loads are issued by hand (inline asm): the compiler sinks an ordinary load towards its first use and widens or converts an
// ordered one right behind it, and either way the prefetch distance collapses; issued here they stay AHEAD rows in front, and
// the row's first touch, pin_row(), carries the one counted wait the row needs.  The compiler does not count these loads: its own
// waits (for the tap loads) can only come out longer than necessary, never shorter.
template <typename TIO> struct PixLd;
template <> struct PixLd<float> {
    template <int IMM> static __device__ __forceinline__ void ld(uint32_t& dst, gcptr base, unsigned voff)
    {
        asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM));
    }
    static __device__ __forceinline__ float cvt(uint32_t r) { return __uint_as_float(r); }
};
template <> struct PixLd<bf16_t> {
    // bf16 -> float32 without an instruction: the D16 "hi" load puts the 16 bits into the upper half of the register and, on
    // gfx950 (SRAM-ECC: D16 loads do not preserve the other half), ZEROES the lower half -- measured, tools/ubench/d16_probe.hip
    // (a register preset to 0xAAAAAAAA reads 0x12030000 after loading 0x1203); the compiler never selects this form itself.
    template <int IMM> static __device__ __forceinline__ void ld(uint32_t& dst, gcptr base, unsigned voff)
    {
        asm volatile("global_load_short_d16_hi %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM));
    }
    static __device__ __forceinline__ float cvt(uint32_t r) { return __uint_as_float(r); }
};

template <> struct PixLd<f16_t> {
    // float16: zero-extended 16-bit load, one v_cvt_f32_f16 per element
    template <int IMM> static __device__ __forceinline__ void ld(uint32_t& dst, gcptr base, unsigned voff)
    {
        asm volatile("global_load_ushort %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM));
    }
    static __device__ __forceinline__ float cvt(uint32_t r) { return (float)__builtin_bit_cast(_Float16, (uint16_t)r); }
};

// Loads the compiler counts itself.  The forward kernels issue their x rows from inline asm and wait for them by hand; that is only
// sound while the register allocator never copies a destination register between the load and the wait (it believes the value is
// there), and in these kernels it does (v_mov_b64 of two in-flight registers right behind the loads: measured, wrong gradients).
// An ordered (relaxed, wavefront-scope atomic) load stays where it is written as well, and every move or use of its result gets
// the compiler's own counted s_waitcnt: correct by construction; the pin AHEAD rows later is where the wait lands.
template <typename TIO> struct SafeLd;
template <> struct SafeLd<float> {
    static __device__ __forceinline__ uint32_t ld(gcptr p) { return gload_here<uint32_t>(p); }
    static __device__ __forceinline__ float cvt(uint32_t r) { return __uint_as_float(r); }
};
template <> struct SafeLd<bf16_t> {
    static __device__ __forceinline__ uint32_t ld(gcptr p) { return (uint32_t)gload_here<uint16_t>(p); }
    static __device__ __forceinline__ float cvt(uint32_t r) { return __uint_as_float(r << 16); }
};
template <> struct SafeLd<f16_t> {
    static __device__ __forceinline__ uint32_t ld(gcptr p) { return (uint32_t)gload_here<uint16_t>(p); }
    static __device__ __forceinline__ float cvt(uint32_t r) { return (float)__builtin_bit_cast(_Float16, (uint16_t)r); }
};
template <int A> __device__ __forceinline__ void pin_raw(uint32_t (&v)[A])
{
#pragma unroll
    for (int i = 0; i < A; ++i) asm volatile("" : "+v"(v[i]));
}

// a pair of horizontally adjacent output pixels: converted once, stored as two elements (the lanes of a wave write 128
// contiguous bytes per instruction)
template <typename TIO> struct PixSt;
template <> struct PixSt<float> {
    typedef f32x2 packed;
    static __device__ __forceinline__ packed prep(f32x2 v) { return v; }
    static __device__ __forceinline__ void st(gcptr p, packed v, int half) { gstore<float>(p, half ? v.y : v.x); }
};
template <> struct PixSt<bf16_t> {
    typedef uint32_t packed;
    static __device__ __forceinline__ packed prep(f32x2 v)
    {
        uint32_t pk;                                                   // one conversion for the two pixels (RNE, NaN stays NaN)
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v.x), "v"(v.y));
        return pk;
    }
    static __device__ __forceinline__ void st(gcptr p, packed v, int half)
    {
        gstore<bf16_t>(p, half ? (bf16_t)(v >> 16) : (bf16_t)v);                    // global_store_short / global_store_short_d16_hi
    }
};

template <> struct PixSt<f16_t> {
    typedef uint32_t packed;
    static __device__ __forceinline__ packed prep(f32x2 v)
    {
        uint32_t pk;                                                   // RNE
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v.x), "v"(v.y));
        return pk;
    }
    static __device__ __forceinline__ void st(gcptr p, packed v, int half)
    {
        gstore<bf16_t>(p, half ? (bf16_t)(v >> 16) : (bf16_t)v);                    // 16 raw bits either way
    }
};

// The 25 taps of conv `conv` of the tap-major pack for this lane's channel, as three register pairs per tap row:
// (w0,w1) (w2,w3) (w4,-).  Stride-2 convs use the pairs as they are; stride-1 convs splat one half (an op_sel modifier).
struct Taps {
    f32x2 p[5][3];
    float bias;
    __device__ __forceinline__ float at(int u, int v) const { return (v & 1) ? p[u][v >> 1].y : p[u][v >> 1].x; }
};

// CT > 0: one scalar base per tap row, the five taps of a row are immediates (C * 4 bytes apart)
template <int CT>
__device__ __forceinline__ void load_taps(Taps& t, const float* __restrict__ wpack, const float* __restrict__ bpack, int conv, int C,
                                          unsigned vow, int has_bias)
{
    const gcptr wb = (gcptr)(wpack + (size_t)conv * 25 * C);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const gcptr rowb = opaque(wb + (size_t)(u * 5 + 2) * C * 4);       // the middle tap: -2C*4 .. +2C*4 fit the immediate field
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float w = gload<float>(rowb + (ptrdiff_t)(v - 2) * (CT > 0 ? CT : C) * 4 + vow);
            if (v & 1) t.p[u][v >> 1].y = w;
            else t.p[u][v >> 1].x = w;
        }
        t.p[u][2].y = 0.f;
    }
    t.bias = has_bias ? gload<float>((gcptr)(bpack + (size_t)conv * C) + vow) : 0.f;
}

// ---- stride-2 5x5 conv, pad 2: in = NI x NI plane as pairs in[NI][(NI+1)/2] (odd NI: the last pair's .y is 0), out NO x NO
// pairs.  Tap pairs: out(o,i) = sum_u [ in[r](2i-2,2i-1).(w0,w1) + in[r](2i,2i+1).(w2,w3) + in[r](2i+2)*w4 ], r = 2o+u-2.
template <int NI, int NO>
__device__ __forceinline__ void down5(const f32x2 (&in)[NI][(NI + 1) / 2], f32x2 (&out)[NO][(NO + 1) / 2], const Taps& t)
{
    constexpr int PI = (NI + 1) / 2;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        f32x2 acc[NO];
#pragma unroll
        for (int i = 0; i < NO; ++i) acc[i] = f32x2{t.bias, 0.f};
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int r = 2 * o + u - 2;
            if (r < 0 || r >= NI) continue;
            // one tap part at a time over all outputs: consecutive instructions never touch the same accumulator (a packed
            // FMA's result cannot be forwarded to the very next instruction; the compiler pads adjacent dependent ones with s_nop)
#pragma unroll
            for (int i = 1; i < NO; ++i) acc[i] = pfma(in[r][i - 1], t.p[u][0], acc[i]);
#pragma unroll
            for (int i = 0; i < NO; ++i) if (i < PI) acc[i] = pfma(in[r][i], t.p[u][1], acc[i]);
#pragma unroll
            for (int i = 0; i < NO; ++i) if (2 * i + 2 < NI) acc[i].x = fmaf(in[r][i + 1].x, t.p[u][2].x, acc[i].x);
        }
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const float v = acc[i].x + acc[i].y;
            if (i & 1) out[o][i >> 1].y = v;
            else out[o][i >> 1].x = v;
        }
        if (NO & 1) out[o][NO >> 1].y = 0.f;
        pin(out[o]);
        RCX_FENCE;
    }
}

// one input row of a stride-1 5x5 conv scattered into the accumulator rows it feeds.  row = the NP pairs of an N-wide row
// (odd N: last .y is 0); acc_of(o) gives the accumulator row of output row o.
template <int N, class AccOf>
__device__ __forceinline__ void conv5_row(const f32x2 (&row)[(N + 1) / 2], int t, const Taps& w, AccOf&& acc_of)
{
    constexpr int NP = (N + 1) / 2;
    // the row shifted by one pixel: odd[j] = (x[2j-1], x[2j]), j = 0..NP (x[-1] = x[N] = 0)
    f32x2 odd[NP + 1];
    const f32x2 zero = f32x2{0.f, 0.f};
    odd[0] = shift1(zero, row[0]);
#pragma unroll
    for (int j = 1; j < NP; ++j) odd[j] = shift1(row[j - 1], row[j]);
    odd[NP] = shift1(row[NP - 1], zero);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int o = t - u + 2;
        if (o < 0 || o >= N) continue;
        f32x2(&a)[NP] = acc_of(o);
        // output pair (2j, 2j+1); tap v = 0..4 reads input pixels 2j+v-2, 2j+v-1.  One tap at a time over all pairs, so that
        // consecutive instructions never touch the same accumulator (see down5)
#pragma unroll
        for (int j = 1; j < NP; ++j) a[j] = pfma(row[j - 1], splat(w.at(u, 0)), a[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) a[j] = pfma(odd[j], splat(w.at(u, 1)), a[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) a[j] = pfma(row[j], splat(w.at(u, 2)), a[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) if (2 * j + 1 < N) a[j] = pfma(odd[j + 1], splat(w.at(u, 3)), a[j]);   // (x[2j+1], x[2j+2]); nothing when both are padding
#pragma unroll
        for (int j = 0; j + 1 < NP; ++j) a[j] = pfma(row[j + 1], splat(w.at(u, 4)), a[j]);
    }
}

// whole stride-1 conv of a small plane (N <= 8): out = conv(in), all rows resident
template <int N>
__device__ __forceinline__ void conv5_plane(const f32x2 (&in)[N][(N + 1) / 2], f32x2 (&out)[N][(N + 1) / 2], const Taps& w)
{
    constexpr int NP = (N + 1) / 2;
#pragma unroll
    for (int o = 0; o < N; ++o)
#pragma unroll
        for (int j = 0; j < NP; ++j) out[o][j] = f32x2{w.bias, (2 * j + 1 < N) ? w.bias : 0.f};
#pragma unroll
    for (int t = 0; t < N; ++t) {
        conv5_row<N>(in[t], t, w, [&](int o) -> f32x2(&)[NP] { return out[o]; });
#pragma unroll
        for (int o = 0; o < N; ++o) if (o >= t - 2 && o <= t + 2) pin(out[o]);
        RCX_FENCE;
    }
    if (N & 1) {        // the padding column collected products of real pixels: clear it, later stages read it as zero padding
#pragma unroll
        for (int o = 0; o < N; ++o) out[o][NP - 1].y = 0.f;
    }
}

// the pair (x[i], x[i+1]) of a row held as aligned pairs: the pair itself (i even) or one v_pk_mov_b32 (i odd)
template <int NP>
__device__ __forceinline__ f32x2 pair_at(const f32x2 (&in)[NP], int i)
{
    return (i & 1) ? shift1(in[i >> 1], in[(i >> 1) + 1 < NP ? (i >> 1) + 1 : i >> 1]) : in[i >> 1];
}

// horizontal resize of one row, NI -> NO pixels (ATen index arithmetic, rcx_lanes.h vtab).  Two output pixels per instruction
// where their source pixels are adjacent (every interior pair of an exact 2x step): out = W1 * (x[i1], x[i1']) + W0 * (x[i0], x[i0']).
template <int MODE, int NI, int NO>
__device__ __forceinline__ void resize_row(const f32x2 (&in)[(NI + 1) / 2], f32x2 (&out)[(NO + 1) / 2])
{
    constexpr int NPI = (NI + 1) / 2;
    auto px = [&](int i) -> float { return (i & 1) ? in[i >> 1].y : in[i >> 1].x; };
    auto one = [&](int q) -> float {
        const VT t = vtab(MODE, NI, NO, q);
        return (MODE == 1 || t.i0 == t.i1) ? px(t.i0) : fmaf(t.l, px(t.i1), (1.f - t.l) * px(t.i0));
    };
#pragma unroll
    for (int j = 0; j < (NO + 1) / 2; ++j) {
        const VT a = vtab(MODE, NI, NO, 2 * j), b = vtab(MODE, NI, NO, 2 * j + 1 < NO ? 2 * j + 1 : 2 * j);
        const bool paired = MODE == 0 && 2 * j + 1 < NO && a.i0 != a.i1 && b.i0 != b.i1 && a.i0 + 1 == b.i0 && a.i1 + 1 == b.i1 &&
                            b.i1 < NI;
        if (paired) out[j] = pfma(f32x2{a.l, b.l}, pair_at<NPI>(in, a.i1), f32x2{1.f - a.l, 1.f - b.l} * pair_at<NPI>(in, a.i0));
        else out[j] = f32x2{one(2 * j), 2 * j + 1 < NO ? one(2 * j + 1) : 0.f};
    }
}

// dst += vertical resize: dst(row d of NO) += (1-l) * h[i0] + l * h[i1]
template <int MODE, int NI, int NO, int NP>
__device__ __forceinline__ void add_resized_row(f32x2 (&dst)[NP], const f32x2 (&h)[NI][NP], int d)
{
    const VT t = vtab(MODE, NI, NO, d);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (MODE == 1 || t.i0 == t.i1) dst[j] = dst[j] + h[t.i0][j];
        else dst[j] = pfma(splat(t.l), h[t.i1][j], pfma(splat(1.f - t.l), h[t.i0][j], dst[j]));
    }
}


// ---- training forward (rcx_recconv2d_fwd_train): the same launch also leaves the float32 pyramid the backward reads -- F_l and C_l,
// l = 1 .. level, each N x h_l x w_l x C (rcx_api.hip, TrainLadder) -- instead of one launch per ladder step.  base == nullptr: inference.
struct SavedPyr {
    float* base;
    unsigned long long f_off[2], c_off[2];                    // byte offsets of F_1, F_2 / C_1, C_2 (level 1: index 0 only)
};
// one plane of NW x NW pixels held as pairs, for this lane's (image, channel)
template <int NW>
__device__ __forceinline__ void save_plane(float* base, unsigned long long off, int n, int C, int c, const f32x2 (&p)[NW][(NW + 1) / 2])
{
    float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) + ((size_t)n * NW * NW) * C + c;
#pragma unroll
    for (int o = 0; o < NW; ++o)
#pragma unroll
        for (int i = 0; i < NW; ++i) q[(size_t)(o * NW + i) * C] = (i & 1) ? p[o][i >> 1].y : p[o][i >> 1].x;
}

}  // namespace cpl14
}  // namespace rcx
