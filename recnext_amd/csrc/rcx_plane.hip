// Fused single-launch RecConv2d forward for gfx950: "plane" schedule.
//
// One workgroup owns one (image, channel block) and carries the whole recursion
// (model/recnext.py:24-34) on chip:
//
//   pass 1   x (HBM, NHWC) --row bands--> LDS band --5x5 stride 2--> F_1 (LDS, fp32)
//   ladder   F_{l+1} = down(F_l)                                   LDS -> LDS   (:27-29)
//   up       C_l = conv_j(T_l);  T_{l-1} = F_{l-1} + resize(C_l)   LDS -> LDS   (:31-33)
//   pass 2   x (L2/MALL re-read) + resize(C_1) --row bands--> LDS band --5x5--> y (HBM)   (:34)
//
// HBM sees x once and y once (the algorithmic bytes); every intermediate lives in LDS as fp32.
// When the zero-bordered plane fits the band (small planes), pass 2 adds resize(C_1) onto the
// band left by pass 1 in place and x is read exactly once.
//
// Lane mapping
//   compute: NHWC puts channels innermost, so a lane owns one channel PAIR (float2 / packed-FMA math)
//            and LPP = CB/2 consecutive lanes cover one pixel of the block's CB channels; each thread
//            produces TW horizontally adjacent outputs from the zero-bordered circular row band, so a
//            band element is read ~1.6x instead of 25x and there is no bounds logic in the hot loop.
//   staging: one lane moves 16 bytes (8 bf16 / 4 fp32 channels) of one pixel; loads for the band
//            D bands ahead are issued before the current band is convolved (register FIFO with
//            static slots), which is what keeps HBM latency off the critical path at 1-2 waves/SIMD.
#include "rcx_common.h"
#include "rcx_opts.h"
#include "rcx_launch.h"

#include <cstdio>
#include <cstdlib>

namespace rcx {

// Diagnostic build only (-DRCX_STAMPS): wave 0 of the first workgroups writes s_memtime at phase
// boundaries into a debug buffer that nothing else reads. The shipped library has no stamps.
#ifdef RCX_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define RCX_STAMP(id)                                                                                   \
    do {                                                                                                \
        if (threadIdx.x == 0 && g_stamp_buf && blockIdx.x < 256)                                        \
            g_stamp_buf[blockIdx.x * 64 + (id)] = __builtin_readcyclecounter();                         \
    } while (0)
#define RCX_ABLATE(a, bit) (((a).ablate >> (bit)) & 1)
#else
#define RCX_STAMP(id) do { } while (0)
#define RCX_ABLATE(a, bit) 0
#endif

constexpr int PL_MAXL = 8;
constexpr int PL_K = 5;
constexpr int PL_P = 2;
constexpr int PL_TW = 7;       // outputs per thread along x in the stride-1 strip convs
constexpr int PL_TW2 = 3;      // ... and in the stride-2 pass (window spans 2*TW+3 columns; odd keeps bank conflicts at 2-way)
constexpr int PL_NT = 1024;    // max threads per workgroup (16 waves; the kernels need < 128 VGPRs)
constexpr int PL_IPB = 2;      // max 16-byte staging items per thread per band
constexpr int PL_DIRECT = 8;   // 16-byte loads in flight per thread when a whole plane is staged at once

struct PlaneArgs {
    int N, C, H, W, level;
    int h[PL_MAXL + 1], w[PL_MAXL + 1];
    int f_off[PL_MAXL + 1];   // float2 offsets of F_l (l >= 1) in LDS
    int c_off[PL_MAXL + 1];   // float2 offsets of C_l (l >= 1) in LDS
    int band_off;             // float2 offset of the row band
    int taps_off;             // float2 offset of the (level+2) tap sets
    int tab_off;              // float2 offset of the resize tables
    int tr_off[PL_MAXL + 1], tc_off[PL_MAXL + 1];   // AxisTab index of the row / column table of level l (from level l+1)
    int band_rows;            // NR: rows in the circular band
    int band_wp;              // padded row width in pixels
    int B2;                   // output rows per band, pass 2 (stride 1)
    int B1;                   // output rows per band, pass 1 (stride 2)
    int nblk;                 // channel blocks per image
    int has_bias;
    int mode;                 // 0 bilinear, 1 nearest
    int single;               // 1: the whole zero-bordered plane lives in the band; pass 2 reuses it in place
    int ablate;               // diagnostic build only: bit mask of work to skip (results are then wrong)
};

// q / d for 0 <= q < 2^20 via a float reciprocal: (q + 0.5) / d is never within rounding distance of an integer
__device__ __forceinline__ int fast_div(int q, float inv_d) { return (int)(((float)q + 0.5f) * inv_d); }
__device__ __forceinline__ int wrap(int v, int n) { return v >= n ? v - n : v; }
__device__ __forceinline__ int mod_pos(int r, int n) { int m = r % n; return m < 0 ? m + n : m; }

// Explicit 2-wide vectors: llvm.fma.v2f32 selects v_pk_fma_f32 on gfx950. Left to the SLP vectoriser
// the dominant stride-1 loop came out as scalar v_fma_f32 (twice the VALU instructions).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f as_v2f(float2 a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ float2 as_f2(v2f a) { return __builtin_bit_cast(float2, a); }
__device__ __forceinline__ float2 fma2(float2 a, float2 b, float2 c)
{
    return as_f2(__builtin_elementwise_fma(as_v2f(a), as_v2f(b), as_v2f(c)));
}
__device__ __forceinline__ float2 add2(float2 a, float2 b) { return as_f2(as_v2f(a) + as_v2f(b)); }
__device__ __forceinline__ float2 lerp2(float2 a, float2 b, float w0, float w1) { return make_float2(w0 * a.x + w1 * b.x, w0 * a.y + w1 * b.y); }

// ---- 16-byte global chunks: EPL channels = EPL/2 channel pairs ----
template <typename T> struct IO;
template <> struct IO<float> {
    static constexpr int CPL = 2;                  // channel pairs per 16-byte load
    static __device__ __forceinline__ void unpack(const uint4& r, float2 (&o)[CPL])
    {
        o[0] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
        o[1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
    }
    static __device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
};
template <> struct IO<bf16_t> {
    static constexpr int CPL = 4;
    static __device__ __forceinline__ void unpack(const uint4& r, float2 (&o)[CPL])
    {
        o[0] = make_float2(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u));
        o[1] = make_float2(__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
        o[2] = make_float2(__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u));
        o[3] = make_float2(__uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u));
    }
    static __device__ __forceinline__ void st2(bf16_t* p, float2 v) { *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(v.x, v.y); }
};

// Taps live in LDS for the whole kernel: set i (0 = down, 1+j = convs[j]) at taps + i*26*LPP, laid out
// [tap 0..24][cp] followed by one bias row [cp].  A conv loop reads the 5 taps of a window row right
// before using them (lanes with the same channel pair broadcast), so taps cost 10 transient VGPRs
// instead of 50 persistent ones and a phase change costs no global round trip.
constexpr int PL_TAPROWS = PL_K * PL_K + 1;

template <int LPP>
__device__ __forceinline__ void preload_taps(float2* __restrict__ taps, const float* __restrict__ wpack, const float* __restrict__ bpack,
                                             int C, int c0, int nsets)
{
    const int total = nsets * PL_TAPROWS * LPP;
    constexpr int B = 8;                                   // loads in flight per thread
    for (int base = threadIdx.x; base < total; base += B * blockDim.x) {
        float2 v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int i = base + j * blockDim.x;
            v[j] = make_float2(0.f, 0.f);
            if (i < total) {
                const int cp = i % LPP;
                const int row = (i / LPP) % PL_TAPROWS;
                const int set = i / (LPP * PL_TAPROWS);
                if (row < PL_K * PL_K) v[j] = *reinterpret_cast<const float2*>(wpack + ((set * PL_K * PL_K + row) * C + c0 + 2 * cp));
                else if (bpack) v[j] = *reinterpret_cast<const float2*>(bpack + (set * C + c0 + 2 * cp));
            }
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int i = base + j * blockDim.x;
            if (i < total) taps[i] = v[j];
        }
    }
}

// Per-axis resize tables, built once per workgroup in LDS: entry d of an axis holds the two source
// indices and their weights (bilinear: (1-lam, lam); nearest: (1, 0) with i1 == i0).  Index and lambda
// arithmetic is ATen's (float), done H+W times per level instead of once per pixel.
struct __attribute__((aligned(16))) AxisTab { int i0, i1; float w0, w1; };

// `pitch` = 1 for a column table, = coarse row width for a row table (entries are then row offsets in pixels)
__device__ __forceinline__ void build_axis(AxisTab* __restrict__ tab, int n_out, int n_in, int mode, int pitch)
{
    const float scale = (float)n_in / (float)n_out;
    for (int d = threadIdx.x; d < n_out; d += blockDim.x) {
        AxisTab t;
        if (mode == 1) { t.i0 = t.i1 = nearest_src(d, n_in, scale); t.w0 = 1.f; t.w1 = 0.f; }
        else { const Lerp l = bilinear_src(d, n_in, scale); t.i0 = l.i0; t.i1 = l.i1; t.w0 = 1.f - l.lam; t.w1 = l.lam; }
        t.i0 *= pitch; t.i1 *= pitch;
        tab[d] = t;
    }
}

// Source taps of resize(src (hs x ws) -> dst) at (y, x): four corner offsets (in pixels) and weights.
struct UpTap { int o00, o01, o10, o11; float w00, w01, w10, w11; };

__device__ __forceinline__ UpTap up_tap(const AxisTab* __restrict__ rows, const AxisTab* __restrict__ cols, int /*ws*/, int y, int x)
{
    const AxisTab ry = rows[y], cx = cols[x];
    UpTap t;
    t.o00 = ry.i0 + cx.i0; t.o01 = ry.i0 + cx.i1;          // row entries are already multiplied by ws
    t.o10 = ry.i1 + cx.i0; t.o11 = ry.i1 + cx.i1;
    t.w00 = ry.w0 * cx.w0; t.w01 = ry.w0 * cx.w1; t.w10 = ry.w1 * cx.w0; t.w11 = ry.w1 * cx.w1;
    return t;
}

// ATen evaluates w00*v00 + w01*v01 + w10*v10 + w11*v11 with the four products of the 1-D weights.
template <int LPP>
__device__ __forceinline__ float2 up_sample(const float2* __restrict__ src, const UpTap& t, int cp)
{
    const float2 a = src[t.o00 * LPP + cp], b = src[t.o01 * LPP + cp], d = src[t.o10 * LPP + cp], e = src[t.o11 * LPP + cp];
    const v2f r = as_v2f(a) * t.w00 + as_v2f(b) * t.w01 + as_v2f(d) * t.w10 + as_v2f(e) * t.w11;
    return as_f2(r);
}

// ---------------- staging: x rows -> zero-bordered circular band ----------------
// A staging item is (row dr of the range, padded column px, 16-byte group g); consecutive threads take
// consecutive groups, so a wave-load covers 1 KiB of consecutive channel chunks / pixels.
template <int LPP, typename TIO>
struct Stage {
    static constexpr int CPL = IO<TIO>::CPL;
    static constexpr int G = LPP / CPL;            // 16-byte groups per pixel of this block
    static_assert(LPP % CPL == 0, "channel block narrower than one 16-byte chunk");

    const TIO* xn; int c0; int H, W, C, wp, nr; int ablate;

    __device__ __forceinline__ bool decode(int it, int r0, int& r, int& px, int& g, float inv_row) const
    {
        const int dr = fast_div(it, inv_row);      // inv_row = 1 / (wp * G)
        const int rem = it - dr * (wp * G);
        px = rem / G; g = rem % G;                 // G is a power of two (compile time)
        r = r0 + dr;
        const int ix = px - PL_P;
        return r >= 0 && r < H && ix >= 0 && ix < W;
    }
    __device__ __forceinline__ uint4 load(int r, int px, int g) const
    {
        const TIO* p = xn + ((r * W + (px - PL_P)) * C + c0 + g * (2 * CPL));     // per-image offsets fit 32 bits (checked on the host)
        return *reinterpret_cast<const uint4*>(p);
    }
};

// issue the loads of rows [r0, r1) into a FIFO slot
template <int LPP, typename TIO>
__device__ __forceinline__ void stage_issue(const Stage<LPP, TIO>& sg, uint4 (&pre)[PL_IPB], int r0, int r1)
{
    const int per_row = sg.wp * Stage<LPP, TIO>::G;
    const int items = (r1 - r0) * per_row;
    const float inv_row = 1.0f / (float)per_row;
#pragma unroll
    for (int j = 0; j < PL_IPB; ++j) {
        const int it = threadIdx.x + j * blockDim.x;
        int r, px, g;
        pre[j] = make_uint4(0u, 0u, 0u, 0u);
        if (it < items && sg.decode(it, r0, r, px, g, inv_row) && !RCX_ABLATE(sg, 5)) pre[j] = sg.load(r, px, g);
    }
}

// convert a FIFO slot (+ resize(coarse) when HAS_COARSE) and write it to the band
template <int LPP, bool HAS_COARSE, typename TIO>
__device__ __forceinline__ void stage_write(const Stage<LPP, TIO>& sg, float2* __restrict__ band, const uint4 (&pre)[PL_IPB],
                                            int r0, int r1, const float2* __restrict__ coarse, int wc,
                                            const AxisTab* __restrict__ trow, const AxisTab* __restrict__ tcol)
{
    constexpr int CPL = Stage<LPP, TIO>::CPL;
    const int per_row = sg.wp * Stage<LPP, TIO>::G;
    const int items = (r1 - r0) * per_row;
    const float inv_row = 1.0f / (float)per_row;
    const int slot0 = mod_pos(r0, sg.nr);
#pragma unroll
    for (int j = 0; j < PL_IPB; ++j) {
        const int it = threadIdx.x + j * blockDim.x;
        if (it >= items) break;
        int r, px, g;
        const bool inside = sg.decode(it, r0, r, px, g, inv_row);
        float2 v[CPL];
        IO<TIO>::unpack(pre[j], v);                          // zeros when outside (issue stored zeros)
        if constexpr (HAS_COARSE) {
            if (inside && !RCX_ABLATE(sg, 3)) {
                const UpTap t = up_tap(trow, tcol, wc, r, px - PL_P);
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[i] = add2(v[i], up_sample<LPP>(coarse, t, g * CPL + i));
            }
        }
        const int slot = wrap(slot0 + (r - r0), sg.nr);
        float2* dst = band + ((slot * sg.wp + px) * LPP + g * CPL);
        if (RCX_ABLATE(sg, 4) && v[0].x != 123456.f) continue;
#pragma unroll
        for (int i = 0; i < CPL; i += 2) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i].x, v[i].y, v[i + 1].x, v[i + 1].y);
    }
}

// Whole-plane / prologue staging: any number of rows, PL_DIRECT 16-byte loads in flight per thread, so a
// small plane costs one HBM round trip.  Split into issue / write halves so the caller can put other
// latency (tap and table preload) under the same round trip.
struct DirectBatch { uint4 raw[PL_DIRECT]; int rr[PL_DIRECT], pp[PL_DIRECT], gg[PL_DIRECT]; };

template <int LPP, typename TIO>
__device__ __forceinline__ void direct_issue(const Stage<LPP, TIO>& sg, DirectBatch& b, int r0, int r1, int base)
{
    const int per_row = sg.wp * Stage<LPP, TIO>::G;
    const int items = (r1 - r0) * per_row;
    const float inv_row = 1.0f / (float)per_row;
#pragma unroll
    for (int j = 0; j < PL_DIRECT; ++j) {
        const int it = base + j * blockDim.x;
        b.raw[j] = make_uint4(0u, 0u, 0u, 0u);
        b.rr[j] = r0; b.pp[j] = 0; b.gg[j] = 0;
        if (it < items && sg.decode(it, r0, b.rr[j], b.pp[j], b.gg[j], inv_row)) b.raw[j] = sg.load(b.rr[j], b.pp[j], b.gg[j]);
    }
}

template <int LPP, typename TIO>
__device__ __forceinline__ void direct_write(const Stage<LPP, TIO>& sg, float2* __restrict__ band, const DirectBatch& b, int r0, int r1, int base)
{
    constexpr int CPL = Stage<LPP, TIO>::CPL;
    const int items = (r1 - r0) * sg.wp * Stage<LPP, TIO>::G;
    const int slot0 = mod_pos(r0, sg.nr);
#pragma unroll
    for (int j = 0; j < PL_DIRECT; ++j) {
        const int it = base + j * blockDim.x;
        if (it < items) {
            float2 v[CPL];
            IO<TIO>::unpack(b.raw[j], v);
            int slot = slot0 + (b.rr[j] - r0);
            while (slot >= sg.nr) slot -= sg.nr;
            float2* dst = band + ((slot * sg.wp + b.pp[j]) * LPP + b.gg[j] * CPL);
#pragma unroll
            for (int i = 0; i < CPL; i += 2) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i].x, v[i].y, v[i + 1].x, v[i + 1].y);
        }
    }
}

// batches starting at item `first_base + threadIdx.x`
template <int LPP, typename TIO>
__device__ __forceinline__ void stage_direct(const Stage<LPP, TIO>& sg, float2* __restrict__ band, int r0, int r1, int first_base = 0)
{
    const int items = (r1 - r0) * sg.wp * Stage<LPP, TIO>::G;
    for (int base = first_base + threadIdx.x; base < items; base += PL_DIRECT * blockDim.x) {
        DirectBatch b;
        direct_issue<LPP, TIO>(sg, b, r0, r1, base);
        direct_write<LPP, TIO>(sg, band, b, r0, r1, base);
    }
}

// band[r][x] += resize(coarse)(r, x) over the real plane (single-band mode, pass 2)
template <int LPP>
__device__ __forceinline__ void band_upadd(float2* __restrict__ band, const PlaneArgs& a, const float2* __restrict__ coarse, int wc,
                                           const AxisTab* __restrict__ trow, const AxisTab* __restrict__ tcol)
{
    const int nq = a.H * a.W;
    const float inv_w = 1.0f / (float)a.W;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
#pragma unroll 4
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int y = fast_div(q, inv_w);
        const int x = q - y * a.W;
        const UpTap t = up_tap(trow, tcol, wc, y, x);
        const int slot = mod_pos(y, a.band_rows);
        float2* p = band + ((slot * a.band_wp + x + PL_P) * LPP + cp);
        *p = add2(*p, up_sample<LPP>(coarse, t, cp));
    }
}

// ---------------- strip convs ----------------
struct NoHook { __device__ __forceinline__ void operator()() const {} };


// 5x5 conv of band rows -> output rows [o0, o1) (stride S): each thread produces a strip of TW outputs.
// No bounds logic: the band carries 2 zero columns left and enough right padding for the last strip.
// `hook` runs exactly once per call, after the FMAs of the thread's first item and before its stores
// (or after the loop for a thread without items): the banded passes use it to retire the prefetched
// loads of the next band while the only younger memory operations are old ones.
template <int LPP, int S, bool TO_GLOBAL, typename TIO, typename Hook = NoHook>
__device__ __forceinline__ void conv_band(const float2* __restrict__ band, const PlaneArgs& a, const float2* __restrict__ tp,
                                          int o0, int o1, int Wo, float2* __restrict__ dst_lds, TIO* __restrict__ yn, int c0,
                                          Hook hook = Hook())
{
    constexpr int TW = S == 1 ? PL_TW : PL_TW2;
    constexpr int SPAN = (TW - 1) * S + PL_K;
    bool hooked = false;
    const int wp = a.band_wp;
    const int strips = (Wo + TW - 1) / TW;
    const int nq = (o1 - o0) * strips;
    const float inv_strips = 1.0f / (float)strips;
    const int slot_base = mod_pos(o0 * S - PL_P, a.band_rows);
    const int row_stride = wp * LPP, ring_span = a.band_rows * wp * LPP;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    int q = threadIdx.x / LPP;
    if (q < nq) {
        const float2 bias = tp[PL_K * PL_K * LPP + cp];
        for (; q < nq; q += qstep) {
            const int dg = fast_div(q, inv_strips);
            const int st = q - dg * strips;
            const int oy = o0 + dg;
            const int ox0 = st * TW;
            int slot = slot_base + dg * S;
            while (slot >= a.band_rows) slot -= a.band_rows;
            int roff = (slot * wp + ox0 * S) * LPP + cp;   // LDS index of the window's first element; advanced row by row
            float2 acc[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[j] = bias;
            // padded column of input ix is ix + 2, so the window of output ox starts at padded column ox*S.
            // The 5 window rows are a ROLLED loop (taps come from LDS, so nothing needs a static row index):
            // one row + the accumulators are live (~70 VGPRs), and latency is hidden by occupancy
            // (4 waves/SIMD), not by hoisting five rows of loads into 250 registers.
#pragma unroll 1
            for (int u = 0; u < PL_K; ++u) {
                const float2* row = band + roff;
                const float2* trow_ = tp + (u * PL_K * LPP + cp);
                float2 v[SPAN], tw[PL_K];
                if (!RCX_ABLATE(a, 2)) {
#pragma unroll
                    for (int s = 0; s < SPAN; ++s) v[s] = row[s * LPP];
                } else {
#pragma unroll
                    for (int s = 0; s < SPAN; ++s) v[s] = make_float2(1.f, 1.f);
                }
#pragma unroll
                for (int i = 0; i < PL_K; ++i) tw[i] = trow_[i * LPP];
                if (!RCX_ABLATE(a, 1)) {
#pragma unroll
                    for (int s = 0; s < SPAN; ++s) {
#pragma unroll
                        for (int j = 0; j < TW; ++j) {
                            const int tap = s - j * S;
                            if (tap >= 0 && tap < PL_K) acc[j] = fma2(tw[tap], v[s], acc[j]);
                        }
                    }
                }
                ++slot;
                roff += row_stride;
                if (slot == a.band_rows) { slot = 0; roff -= ring_span; }
            }
            if (!hooked) { hook(); hooked = true; }
            if (!(RCX_ABLATE(a, 0) && acc[0].x != 123456.f)) {
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int ox = ox0 + j;
                    if (ox < Wo) {
                        if constexpr (TO_GLOBAL) IO<TIO>::st2(yn + ((oy * Wo + ox) * a.C + c0 + 2 * cp), acc[j]);
                        else dst_lds[(oy * Wo + ox) * LPP + cp] = acc[j];
                    }
                }
            }
        }
    }
    if (!hooked) hook();
}

// LDS plane -> LDS plane 5x5 conv, stride S, strips of TW outputs, zero padding by predication.
// All 5 window rows are loaded (predicated) before any of them is used, so a thread pays one LDS
// round trip per item instead of one per tap: these planes are small and latency, not bandwidth,
// is what they cost.
template <int LPP, int S, int TW>
__device__ __forceinline__ void conv_lds_tw(const float2* __restrict__ src, int hs, int ws, float2* __restrict__ dst, int hd, int wd,
                                            const float2* __restrict__ tp)
{
    constexpr int SPAN = (TW - 1) * S + PL_K;
    const int strips = (wd + TW - 1) / TW;
    const int nq = hd * strips;
    const float inv_strips = 1.0f / (float)strips;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int oy = fast_div(q, inv_strips);
        const int ox0 = (q - oy * strips) * TW;
        float2 acc[TW];
        {
            const float2 bias = tp[PL_K * PL_K * LPP + cp];
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[j] = bias;
        }
        bool colok[SPAN];
#pragma unroll
        for (int s = 0; s < SPAN; ++s) { const int ix = ox0 * S + s - PL_P; colok[s] = ix >= 0 && ix < ws; }
        const int base = ((oy * S - PL_P) * ws + ox0 * S - PL_P) * LPP + cp;      // may be negative: only dereferenced where valid
        // rolled window rows: one row + accumulators live, occupancy hides the LDS latency (see conv_band)
#pragma unroll 1
        for (int u = 0; u < PL_K; ++u) {
            const int iy = oy * S + u - PL_P;
            if (iy < 0 || iy >= hs) continue;              // zero padding: the whole row contributes nothing
            const float2* row = src + (base + u * ws * LPP);
            const float2* trow_ = tp + (u * PL_K * LPP + cp);
            float2 v[SPAN], tw[PL_K];
#pragma unroll
            for (int s = 0; s < SPAN; ++s) {
                v[s] = make_float2(0.f, 0.f);
                if (colok[s]) v[s] = row[s * LPP];
            }
#pragma unroll
            for (int i = 0; i < PL_K; ++i) tw[i] = trow_[i * LPP];
#pragma unroll
            for (int s = 0; s < SPAN; ++s) {
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int tap = s - j * S;
                    if (tap >= 0 && tap < PL_K) acc[j] = fma2(tw[tap], v[s], acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TW; ++j)
            if (ox0 + j < wd) dst[(oy * wd + ox0 + j) * LPP + cp] = acc[j];
    }
}

// Small planes have few outputs: a wide strip would leave most threads idle behind one long dependent
// chain per active thread.  Pick the widest strip that still gives every thread an item.
template <int LPP, int S>
__device__ __forceinline__ void conv_lds(const float2* __restrict__ src, int hs, int ws, float2* __restrict__ dst, int hd, int wd,
                                         const float2* __restrict__ tp)
{
    const int lanes = blockDim.x / LPP;                    // strip items that run concurrently
    const int rows = hd;
    // Odd widths only: consecutive lane groups then sit an odd number of pixels apart, which spreads their
    // 8-byte reads over all LDS banks (an even pitch folds them onto the same banks: 2- to 4-way conflicts).
    if (S == 1 && rows * ((wd + 6) / 7) >= lanes) conv_lds_tw<LPP, S, 7>(src, hs, ws, dst, hd, wd, tp);
    else if (rows * ((wd + 4) / 5) >= lanes) conv_lds_tw<LPP, S, 5>(src, hs, ws, dst, hd, wd, tp);
    else if (rows * ((wd + 2) / 3) >= lanes) conv_lds_tw<LPP, S, 3>(src, hs, ws, dst, hd, wd, tp);
    else conv_lds_tw<LPP, S, 1>(src, hs, ws, dst, hd, wd, tp);
}

template <int LPP>
__device__ __forceinline__ void upadd_plane(const float2* __restrict__ coarse, int wc, float2* __restrict__ fine, int hf, int wf,
                                            const AxisTab* __restrict__ trow, const AxisTab* __restrict__ tcol)
{
    const int nq = hf * wf;
    const float inv_wf = 1.0f / (float)wf;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
#pragma unroll 4
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int y = fast_div(q, inv_wf);
        const int x = q - y * wf;
        const UpTap t = up_tap(trow, tcol, wc, y, x);
        fine[q * LPP + cp] = add2(fine[q * LPP + cp], up_sample<LPP>(coarse, t, cp));
    }
}

// ---------------- one banded pass over x ----------------
// S == 2: F_1 = down(x) into dst_lds.   S == 1: y = conv(x + resize(coarse)) to global.
// Band b produces output rows [b*Bo, (b+1)*Bo) and needs input rows up to last(b); rows are staged
// in order, `first(b)`..`first(b+1)` being band b's new rows (Bo*S of them), so slot FIFO[b % D]
// always holds exactly one band's worth of loads.
template <int LPP, int S, bool HAS_COARSE, typename TIO>
struct Pass {
    const PlaneArgs& a;
    const Stage<LPP, TIO>& sg;
    float2* band;
    const float2* tp;
    const float2* coarse; int wc; const AxisTab* trow; const AxisTab* tcol;
    float2* dst_lds; TIO* yn; int c0;
    int Ho, Wo, Bo, nb;

    // first new input row of band b (rows before first(0) are staged by the prologue)
    __device__ __forceinline__ int first(int b) const { return b * Bo * S + (S == 2 ? 1 : PL_P); }
    __device__ __forceinline__ int row_end(int b) const { const int e = first(b + 1); const int lim = a.H + PL_P; return e < lim ? e : lim; }

    __device__ __forceinline__ void run() const
    {
        uint4 pre[PL_IPB];        // loads in flight: the NEXT band's new rows
        uint4 cur[PL_IPB];        // the band about to be written to LDS (already arrived)
        // rows [-2, first(0)): zero rows and the first real row(s)
        if constexpr (HAS_COARSE) {
            stage_issue<LPP, TIO>(sg, cur, -PL_P, first(0));
            stage_write<LPP, true, TIO>(sg, band, cur, -PL_P, first(0), coarse, wc, trow, tcol);
        } else {
            stage_direct<LPP, TIO>(sg, band, -PL_P, first(0));
        }
        stage_issue<LPP, TIO>(sg, cur, first(0), row_end(0));
        for (int b = 0; b < nb; ++b) {
            __syncthreads();                              // band b-1's readers are done with the slots we overwrite
            if (S == 1 && b < 8) RCX_STAMP(8 + 3 * b);
            // one band ahead: these loads have the whole of band b's staging + conv to arrive
            if (b + 1 < nb) stage_issue<LPP, TIO>(sg, pre, first(b + 1), row_end(b + 1));
            stage_write<LPP, HAS_COARSE, TIO>(sg, band, cur, first(b), row_end(b), coarse, wc, trow, tcol);
            __syncthreads();
            if (S == 1 && b < 8) RCX_STAMP(9 + 3 * b);
            const int o0 = b * Bo, o1 = min(o0 + Bo, Ho);
            // Retire the prefetch between this band's FMAs and its stores: the wait then covers loads issued
            // a whole band ago and no store that was issued a moment ago (vmcnt counts both on gfx950).
            conv_band<LPP, S, S == 1, TIO>(band, a, tp, o0, o1, Wo, dst_lds, yn, c0, [&]() {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < PL_IPB; ++j) cur[j] = pre[j];
#pragma unroll
                for (int j = 0; j < PL_IPB; ++j) asm volatile("" : "+v"(cur[j].x), "+v"(cur[j].y), "+v"(cur[j].z), "+v"(cur[j].w));
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
};

template <int LPP, typename TIO>
__global__ void __launch_bounds__(PL_NT)
k_recconv_plane(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                PlaneArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);

    // blocks b and b+8 share an XCD (observed round-robin dispatch): keep the channel blocks of one
    // image on one XCD and adjacent in dispatch order so they share its L2 lines. Speed only.
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int blk = q % a.nblk;
    const int n = (q / a.nblk) * 8 + xcd;
    if (n >= a.N) return;
    constexpr int CB = 2 * LPP;
    const int c0 = blk * CB;
    const TIO* xn = x + (size_t)n * a.H * a.W * a.C;
    TIO* yn = y + (size_t)n * a.H * a.W * a.C;
    float2* band = lds + a.band_off;
    float2* taps = lds + a.taps_off;
    const int L = a.level;
    auto taps_of = [&](int i) { return taps + i * PL_TAPROWS * LPP; };
    const Stage<LPP, TIO> sg{xn, c0, a.H, a.W, a.C, a.band_wp, a.band_rows, a.ablate};

    // Whole-plane mode: put the first batch of x loads in flight before anything else, so the tap / table
    // preload below (L2 round trips) hides under the same HBM round trip.
    DirectBatch first;
    if (a.single) direct_issue<LPP, TIO>(sg, first, -PL_P, a.H + PL_P, threadIdx.x);
    // all (L+2) tap sets of this channel block and the resize tables, once; their first readers are
    // behind the first staging barrier
    RCX_STAMP(0);
    preload_taps<LPP>(taps, wpack, a.has_bias ? bpack : nullptr, a.C, c0, L + 2);
    AxisTab* tabs = reinterpret_cast<AxisTab*>(lds + a.tab_off);
    for (int l = 0; l < L; ++l) {
        build_axis(tabs + a.tr_off[l], a.h[l], a.h[l + 1], a.mode, a.w[l + 1]);
        build_axis(tabs + a.tc_off[l], a.w[l], a.w[l + 1], a.mode, 1);
    }
    if (a.single) {
        direct_write<LPP, TIO>(sg, band, first, -PL_P, a.H + PL_P, threadIdx.x);
        stage_direct<LPP, TIO>(sg, band, -PL_P, a.H + PL_P, PL_DIRECT * blockDim.x);     // the rest, if the plane is larger
    }

    if (L >= 1) {
        // ---- pass 1: F_1 = down(x) ----
        float2* F1 = lds + a.f_off[1];
        if (a.single) {
            __syncthreads();
            conv_band<LPP, 2, false, TIO>(band, a, taps_of(0), 0, a.h[1], a.w[1], F1, nullptr, c0);
        } else {
            const Pass<LPP, 2, false, TIO> p1{a, sg, band, taps_of(0), nullptr, 0, nullptr, nullptr, F1, nullptr, c0,
                                              a.h[1], a.w[1], a.B1, (a.h[1] + a.B1 - 1) / a.B1};
            p1.run();
        }
        __syncthreads();
        RCX_STAMP(1);
        // ---- ladder: F_{l+1} = down(F_l) ----
        for (int l = 1; l < L; ++l) {
            conv_lds<LPP, 2>(lds + a.f_off[l], a.h[l], a.w[l], lds + a.f_off[l + 1], a.h[l + 1], a.w[l + 1], taps_of(0));
            __syncthreads();
        }
        RCX_STAMP(2);
        // ---- up recursion, coarsest first: C_l = conv_j(T_l); T_{l-1} = F_{l-1} + resize(C_l) ----
        for (int l = L, j = 0; l >= 1; --l, ++j) {
            if (l == 1) RCX_STAMP(3);
            conv_lds<LPP, 1>(lds + a.f_off[l], a.h[l], a.w[l], lds + a.c_off[l], a.h[l], a.w[l], taps_of(1 + j));
            __syncthreads();
            if (l > 1) {
                upadd_plane<LPP>(lds + a.c_off[l], a.w[l], lds + a.f_off[l - 1], a.h[l - 1], a.w[l - 1],
                                 tabs + a.tr_off[l - 1], tabs + a.tc_off[l - 1]);
                __syncthreads();
            }
        }
    }
    RCX_STAMP(4);
    // ---- pass 2: y = conv_L(x + resize(C_1)) ----
    const float2* C1 = L >= 1 ? lds + a.c_off[1] : nullptr;
    const int wc = L >= 1 ? a.w[1] : 1;
    const AxisTab* trow = tabs + a.tr_off[0];
    const AxisTab* tcol = tabs + a.tc_off[0];
    if (a.single) {
        if (L >= 1) band_upadd<LPP>(band, a, C1, wc, trow, tcol);       // the band still holds x (staged at kernel start)
        __syncthreads();                                                // level 0: y = conv_0(x) straight from the staged plane
        conv_band<LPP, 1, true, TIO>(band, a, taps_of(1 + L), 0, a.H, a.W, nullptr, yn, c0);
    } else if (L >= 1) {
        const Pass<LPP, 1, true, TIO> p2{a, sg, band, taps_of(1 + L), C1, wc, trow, tcol, nullptr, yn, c0,
                                         a.H, a.W, a.B2, (a.H + a.B2 - 1) / a.B2};
        p2.run();
    } else {
        const Pass<LPP, 1, false, TIO> p2{a, sg, band, taps_of(1 + L), nullptr, 0, nullptr, nullptr, nullptr, yn, c0,
                                          a.H, a.W, a.B2, (a.H + a.B2 - 1) / a.B2};
        p2.run();
    }
    RCX_STAMP(5);
}

// Whole-plane planes only (the zero-bordered plane is the band): same phases as k_recconv_plane's `single`
// path, but compiled on its own so the register allocator is not dragged to 200+ VGPRs by the banded
// machinery (capping it at 128 VGPRs for 4 waves/SIMD still spills into the hot loops: see DESIGN.md section 6).
#ifndef RCX_WHOLE_WAVES
#define RCX_WHOLE_WAVES 2
#endif
template <int LPP, typename TIO>
__global__ void __launch_bounds__(PL_NT, RCX_WHOLE_WAVES)
k_recconv_whole(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                PlaneArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int blk = q % a.nblk;
    const int n = (q / a.nblk) * 8 + xcd;
    if (n >= a.N) return;
    const int c0 = blk * 2 * LPP;
    const TIO* xn = x + (size_t)n * a.H * a.W * a.C;
    TIO* yn = y + (size_t)n * a.H * a.W * a.C;
    float2* band = lds + a.band_off;
    float2* taps = lds + a.taps_off;
    const int L = a.level;
    auto taps_of = [&](int i) { return taps + i * PL_TAPROWS * LPP; };
    const Stage<LPP, TIO> sg{xn, c0, a.H, a.W, a.C, a.band_wp, a.band_rows, 0};
    {
        DirectBatch first;
        direct_issue<LPP, TIO>(sg, first, -PL_P, a.H + PL_P, threadIdx.x);
        preload_taps<LPP>(taps, wpack, a.has_bias ? bpack : nullptr, a.C, c0, L + 2);
        direct_write<LPP, TIO>(sg, band, first, -PL_P, a.H + PL_P, threadIdx.x);
    }
    AxisTab* tabs = reinterpret_cast<AxisTab*>(lds + a.tab_off);
    for (int l = 0; l < L; ++l) {
        build_axis(tabs + a.tr_off[l], a.h[l], a.h[l + 1], a.mode, a.w[l + 1]);
        build_axis(tabs + a.tc_off[l], a.w[l], a.w[l + 1], a.mode, 1);
    }
    stage_direct<LPP, TIO>(sg, band, -PL_P, a.H + PL_P, PL_DIRECT * blockDim.x);
    __syncthreads();
    if (L >= 1) {
        conv_band<LPP, 2, false, TIO>(band, a, taps_of(0), 0, a.h[1], a.w[1], lds + a.f_off[1], nullptr, c0);
        __syncthreads();
        for (int l = 1; l < L; ++l) {
            conv_lds<LPP, 2>(lds + a.f_off[l], a.h[l], a.w[l], lds + a.f_off[l + 1], a.h[l + 1], a.w[l + 1], taps_of(0));
            __syncthreads();
        }
        for (int l = L, j = 0; l >= 1; --l, ++j) {
            conv_lds<LPP, 1>(lds + a.f_off[l], a.h[l], a.w[l], lds + a.c_off[l], a.h[l], a.w[l], taps_of(1 + j));
            __syncthreads();
            if (l > 1) {
                upadd_plane<LPP>(lds + a.c_off[l], a.w[l], lds + a.f_off[l - 1], a.h[l - 1], a.w[l - 1],
                                 tabs + a.tr_off[l - 1], tabs + a.tc_off[l - 1]);
                __syncthreads();
            }
        }
        band_upadd<LPP>(band, a, lds + a.c_off[1], a.w[1], tabs + a.tr_off[0], tabs + a.tc_off[0]);
        __syncthreads();
    }
    conv_band<LPP, 1, true, TIO>(band, a, taps_of(1 + L), 0, a.H, a.W, nullptr, yn, c0);
}

// ---------------- host side ----------------
struct PlanePlan {
    bool ok;
    int lpp;
    int nt;
    size_t lds_bytes;
    PlaneArgs args;
};

static inline int down_size5(int h) { return (h + 2 * PL_P - PL_K) / 2 + 1; }

static int env_int(rcx::opt::Id id, int dflt)
{
    const char* v = rcx::opt::value(id);
    return v && *v ? atoi(v) : dflt;
}

static void fill_args(PlaneArgs& a, int N, int C, int H, int W, int level, int lpp, int B2cand, size_t& lds_bytes)
{
    a.N = N; a.C = C; a.H = H; a.W = W; a.level = level;
    a.h[0] = H; a.w[0] = W;
    for (int l = 1; l <= level; ++l) { a.h[l] = down_size5(a.h[l - 1]); a.w[l] = down_size5(a.w[l - 1]); }
    // band geometry first: the small levels may alias it
    a.single = B2cand >= H ? 1 : 0;
    a.B2 = a.single ? H : B2cand;
    a.band_rows = a.B2 + 2 * PL_P;
    a.B1 = (a.band_rows - 3) / 2;
    if (a.B1 < 1) a.B1 = 1;
    if (a.band_rows < 2 * a.B1 + 3) a.band_rows = 2 * a.B1 + 3;
    // padded row: 2 zero columns left, and enough on the right for the last partial strip of either pass
    const int strips2 = (W + PL_TW - 1) / PL_TW;
    int wp = strips2 * PL_TW + 2 * PL_P;
    if (level >= 1) {
        const int strips1 = (a.w[1] + PL_TW2 - 1) / PL_TW2;
        const int wp1 = 2 * strips1 * PL_TW2 + 3;
        if (wp1 > wp) wp = wp1;
    }
    a.band_wp = wp;
    const size_t band_px = (size_t)a.band_rows * wp;

    size_t pix = 0;                                // LDS pixels (each lpp float2 wide)
    size_t small = 0;                              // F_2 .. F_level
    for (int l = 2; l <= level; ++l) small += (size_t)a.h[l] * a.w[l];
    // F_1 (which becomes T_1) is live from pass 1 to the end of the up recursion: its own region.
    if (level >= 1) { a.f_off[1] = 0; pix += (size_t)a.h[1] * a.w[1]; }
    // C region: C_1 from its start; C_2, C_3, ... packed from its start too (all dead before C_1 is written)
    const size_t cbase = pix;
    size_t csmall = 0;
    for (int l = 2; l <= level; ++l) { a.c_off[l] = (int)((cbase + csmall) * lpp); csmall += (size_t)a.h[l] * a.w[l]; }
    size_t creg = 0;
    if (level >= 1) { a.c_off[1] = (int)(cbase * lpp); creg = (size_t)a.h[1] * a.w[1]; }
    if (csmall > creg) creg = csmall;
    pix += creg;
    a.band_off = (int)(pix * lpp);
    pix += band_px;
    // F_2 .. F_level are only live between the two passes: they overlay the (then idle) band, unless the
    // band is the whole plane that pass 2 reuses in place
    size_t fbase = pix;
    if (!a.single && small <= band_px) fbase = (size_t)a.band_off / lpp;
    else pix += small;
    size_t fo = 0;
    for (int l = 2; l <= level; ++l) { a.f_off[l] = (int)((fbase + fo) * lpp); fo += (size_t)a.h[l] * a.w[l]; }
    a.taps_off = (int)(pix * lpp);
    pix += (size_t)(level + 2) * PL_TAPROWS;
    // resize tables: 16-byte entries = 2 float2 each
    a.tab_off = (int)(pix * lpp);
    int ent = 0;
    for (int l = 0; l < level; ++l) { a.tr_off[l] = ent; ent += a.h[l]; a.tc_off[l] = ent; ent += a.w[l]; }
    lds_bytes = pix * lpp * sizeof(float2) + (size_t)ent * 16;
}

// 16-byte staging items of the largest band (prologue included); must fit PL_IPB per thread
static int band_stage_items(const PlaneArgs& a, int lpp, int cpl)
{
    int rows = a.B2 > 2 * a.B1 ? a.B2 : 2 * a.B1;
    if (rows < 2 * PL_P) rows = 2 * PL_P;              // the prologue stages rows [-2, 2)
    return rows * a.band_wp * (lpp / cpl);
}

static int pick_threads(const PlaneArgs& a, int lpp, int cpl)
{
    // Measured (profiles/archive/r01b_plane_knob_sweep_*): these kernels are latency-bound, so about two threads per
    // strip item of the widest conv (every item then has a partner wave to hide its LDS round trips), and in
    // banded mode enough threads that one band's staging fits PL_IPB 16-byte items per thread.
    const int stage_items = band_stage_items(a, lpp, cpl);
    const int conv_items = (a.single ? a.H : a.B2) * ((a.W + PL_TW - 1) / PL_TW) * lpp;
    int nt = 128;
    while (nt < PL_NT && nt < conv_items) nt *= 2;
    if (nt < PL_NT && (!a.single || conv_items <= 256)) nt *= 2;
    while (nt < PL_NT && !a.single && stage_items > nt * PL_IPB) nt *= 2;
    if (nt < lpp) nt = lpp;
    return nt;
}

PlanePlan plan_plane(int N, int C, int H, int W, int level, int k, int dtype)
{
    PlanePlan none{};
    none.ok = false;
    if (k != PL_K || level < 0 || level > PL_MAXL || (C % 8) != 0) return none;
    if (dtype > 1) return none;                       // float16 I/O: the channel-per-lane kernels and the generic schedule (rcx_api.hip)
    if ((long long)H * W > (1 << 18)) return none;        // fast_div range
    if ((long long)H * W * C >= (1LL << 30)) return none; // per-image element offsets are 32-bit in the kernels
    const size_t LDS_CU = 160 * 1024;
    const int cpl = dtype == 1 ? 4 : 2;                    // channel pairs per 16-byte chunk
    const int force_lpp = env_int(rcx::opt::PLANE_LPP, 0), force_b2 = env_int(rcx::opt::PLANE_B2, 0);
    const int force_nt = env_int(rcx::opt::PLANE_NT, 0);
    static const int lpps[] = {32, 16, 8, 4};
    constexpr int WHOLE = 1 << 20;

    auto try_cfg = [&](int lpp, int B2, PlanePlan& out) -> bool {
        if (C % (2 * lpp) || lpp < cpl) return false;
        if (force_lpp && lpp != force_lpp) return false;
        if (force_b2 && (B2 == WHOLE ? H : B2) != force_b2) return false;
        if (B2 != WHOLE && B2 >= H) return false;           // same thing as the whole-plane candidate
        PlaneArgs a{};
        size_t bytes = 0;
        fill_args(a, N, C, H, W, level, lpp, B2, bytes);
        if (bytes > LDS_CU) return false;
        a.nblk = C / (2 * lpp);
        out.ok = true; out.lpp = lpp; out.lds_bytes = bytes; out.args = a;
        out.nt = force_nt ? force_nt : pick_threads(a, lpp, cpl);
        if (out.nt < 64 || out.nt > PL_NT || out.nt % 64 || out.nt % lpp) out.nt = 256;
        if (!a.single && band_stage_items(a, lpp, cpl) > out.nt * PL_IPB) return false;   // FIFO slot too small
        return true;
    };

    // Measured on MI355X (tools/sweep_plane.py, profiles/archive/r01b_plane_knob_sweep_*): planes up to 16x16 run best
    // whole (x read once, no band loop) with the widest channel block that still lets two workgroups share a
    // CU; larger planes run banded with the widest block that fits and the tallest band that fits with it.
    PlanePlan p{};
    const bool small = (long long)H * W <= 256 || force_b2 == H;
    if (small) {
        for (int lpp : lpps)
            if (try_cfg(lpp, WHOLE, p) && p.lds_bytes <= LDS_CU / 2) return p;
        for (int lpp : lpps)
            if (try_cfg(lpp, WHOLE, p)) return p;
    }
    static const int b2s[] = {8, 4, 2};
    for (int lpp : lpps)
        for (int B2 : b2s)
            if (try_cfg(lpp, B2, p)) return p;
    for (int lpp : lpps)                                   // planes that only fit whole
        if (try_cfg(lpp, WHOLE, p)) return p;
    return none;
}

template <int LPP, typename TIO>
static hipError_t launch_plane_t(const void* x, void* y, const float* wpack, const float* bpack, const PlanePlan& p, int mode, hipStream_t s)
{
    auto kfn = p.args.single ? k_recconv_whole<LPP, TIO> : k_recconv_plane<LPP, TIO>;
    RCX_SET_LDS_ONCE(kfn, p.lds_bytes);
    PlaneArgs a = p.args;
    a.has_bias = bpack != nullptr;
    a.mode = mode;
#ifdef RCX_STAMPS
    a.ablate = env_int(rcx::opt::PLANE_ABLATE, 0);
#else
    a.ablate = 0;
#endif
    const int groups = (a.N + 7) / 8;
    const unsigned grid = (unsigned)(groups * 8 * a.nblk);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(p.nt), p.lds_bytes, s, (const TIO*)x, (TIO*)y, wpack, bpack, a);
    return hipGetLastError();
}

template <typename TIO>
static hipError_t launch_plane_l(const void* x, void* y, const float* wpack, const float* bpack, const PlanePlan& p, int mode, hipStream_t s)
{
    switch (p.lpp) {
    case 32: return launch_plane_t<32, TIO>(x, y, wpack, bpack, p, mode, s);
    case 16: return launch_plane_t<16, TIO>(x, y, wpack, bpack, p, mode, s);
    case 8: return launch_plane_t<8, TIO>(x, y, wpack, bpack, p, mode, s);
    default:
        if constexpr (IO<TIO>::CPL <= 4) return launch_plane_t<4, TIO>(x, y, wpack, bpack, p, mode, s);
        return hipErrorInvalidConfiguration;
    }
}

bool plane_applicable(int N, int C, int H, int W, int level, int k, int dtype) { return plan_plane(N, C, H, W, level, k, dtype).ok; }

// human-readable description of the chosen schedule (logs / benchmarks)
int plane_describe(int N, int C, int H, int W, int level, int k, int dtype, char* buf, int len)
{
    const PlanePlan p = plan_plane(N, C, H, W, level, k, dtype);
    if (!p.ok) return 0;
    return snprintf(buf, len, "plane(cb=%d,%s,nt=%d,lds=%zu)", 2 * p.lpp,
                    p.args.single ? "whole-plane" : (p.args.B2 == 8 ? "band8" : (p.args.B2 == 4 ? "band4" : "band2")), p.nt, p.lds_bytes);
}

#ifdef RCX_STAMPS
hipError_t set_stamp_buffer(void* p)
{
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &q, sizeof(q));
}
#endif

hipError_t plane_recconv(const void* x, void* y, const float* wpack, const float* bpack,
                         int N, int C, int H, int W, int level, int k, int mode, int dtype, hipStream_t s)
{
    const PlanePlan p = plan_plane(N, C, H, W, level, k, dtype);
    if (!p.ok) return hipErrorInvalidConfiguration;
    if (dtype == 1) return launch_plane_l<bf16_t>(x, y, wpack, bpack, p, mode, s);
    return launch_plane_l<float>(x, y, wpack, bpack, p, mode, s);
}

}  // namespace rcx
