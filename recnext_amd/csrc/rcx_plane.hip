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
#include "rcx_launch.h"

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
#else
#define RCX_STAMP(id) do { } while (0)
#endif

constexpr int PL_MAXL = 8;
constexpr int PL_K = 5;
constexpr int PL_P = 2;
constexpr int PL_TW = 7;       // outputs per thread along x in the stride-1 strip convs
constexpr int PL_TW2 = 4;      // ... and in the stride-2 pass (its window spans 2*TW+3 columns)
constexpr int PL_NT = 512;     // max threads per workgroup
constexpr int PL_D = 3;        // prefetch depth in bands
constexpr int PL_IPB = 2;      // max 16-byte staging items per thread per band

struct PlaneArgs {
    int N, C, H, W, level;
    int h[PL_MAXL + 1], w[PL_MAXL + 1];
    int f_off[PL_MAXL + 1];   // float2 offsets of F_l (l >= 1) in LDS
    int c_off[PL_MAXL + 1];   // float2 offsets of C_l (l >= 1) in LDS
    int band_off;             // float2 offset of the row band
    int taps_off;             // float2 offset of the (level+2) tap sets
    int band_rows;            // NR: rows in the circular band
    int band_wp;              // padded row width in pixels
    int B2;                   // output rows per band, pass 2 (stride 1)
    int B1;                   // output rows per band, pass 1 (stride 2)
    int nblk;                 // channel blocks per image
    int has_bias;
    int mode;                 // 0 bilinear, 1 nearest
    int single;               // 1: the whole zero-bordered plane lives in the band; pass 2 reuses it in place
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
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int cp = i % LPP;
        const int row = (i / LPP) % PL_TAPROWS;
        const int set = i / (LPP * PL_TAPROWS);
        float2 v = make_float2(0.f, 0.f);
        if (row < PL_K * PL_K) v = *reinterpret_cast<const float2*>(wpack + ((size_t)set * PL_K * PL_K + row) * C + c0 + 2 * cp);
        else if (bpack) v = *reinterpret_cast<const float2*>(bpack + (size_t)set * C + c0 + 2 * cp);
        taps[i] = v;
    }
}

// Source taps of resize(src (hs x ws) -> dst) at (y, x): four corner offsets (in pixels) and weights.
struct UpTap { int o00, o01, o10, o11; float w00, w01, w10, w11; };

__device__ __forceinline__ UpTap up_tap(int mode, int hs, int ws, float sy, float sx, int y, int x)
{
    UpTap t;
    if (mode == 1) {
        const int cy = nearest_src(y, hs, sy), cx = nearest_src(x, ws, sx);
        t.o00 = t.o01 = t.o10 = t.o11 = cy * ws + cx;
        t.w00 = 1.f; t.w01 = t.w10 = t.w11 = 0.f;
    } else {
        const Lerp ly = bilinear_src(y, hs, sy), lx = bilinear_src(x, ws, sx);
        t.o00 = ly.i0 * ws + lx.i0; t.o01 = ly.i0 * ws + lx.i1;
        t.o10 = ly.i1 * ws + lx.i0; t.o11 = ly.i1 * ws + lx.i1;
        const float wx1 = lx.lam, wx0 = 1.f - lx.lam, wy1 = ly.lam, wy0 = 1.f - ly.lam;
        t.w00 = wy0 * wx0; t.w01 = wy0 * wx1; t.w10 = wy1 * wx0; t.w11 = wy1 * wx1;
    }
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

    const TIO* xn; int c0; int H, W, C, wp, nr;

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
        const TIO* p = xn + ((size_t)r * W + (px - PL_P)) * C + c0 + g * (2 * CPL);
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
        if (it < items && sg.decode(it, r0, r, px, g, inv_row)) pre[j] = sg.load(r, px, g);
    }
}

// convert a FIFO slot (+ resize(coarse) when HAS_COARSE) and write it to the band
template <int LPP, bool HAS_COARSE, typename TIO>
__device__ __forceinline__ void stage_write(const Stage<LPP, TIO>& sg, float2* __restrict__ band, const uint4 (&pre)[PL_IPB],
                                            int r0, int r1, const float2* __restrict__ coarse, int hc, int wc,
                                            float sy, float sx, int mode)
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
            if (inside) {
                const UpTap t = up_tap(mode, hc, wc, sy, sx, r, px - PL_P);
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[i] = add2(v[i], up_sample<LPP>(coarse, t, g * CPL + i));
            }
        }
        const int slot = wrap(slot0 + (r - r0), sg.nr);
        float2* dst = band + ((size_t)slot * sg.wp + px) * LPP + g * CPL;
#pragma unroll
        for (int i = 0; i < CPL; i += 2) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i].x, v[i].y, v[i + 1].x, v[i + 1].y);
    }
}

// any number of rows, loads batched PL_IPB deep (prologue rows and the single-band mode)
template <int LPP, typename TIO>
__device__ __forceinline__ void stage_direct(const Stage<LPP, TIO>& sg, float2* __restrict__ band, int r0, int r1)
{
    constexpr int CPL = Stage<LPP, TIO>::CPL;
    const int per_row = sg.wp * Stage<LPP, TIO>::G;
    const int items = (r1 - r0) * per_row;
    const float inv_row = 1.0f / (float)per_row;
    const int slot0 = mod_pos(r0, sg.nr);
    for (int base = threadIdx.x; base < items; base += PL_IPB * blockDim.x) {
        uint4 raw[PL_IPB];
        int rr[PL_IPB], pp[PL_IPB], gg[PL_IPB];
#pragma unroll
        for (int j = 0; j < PL_IPB; ++j) {
            const int it = base + j * blockDim.x;
            raw[j] = make_uint4(0u, 0u, 0u, 0u);
            rr[j] = r0; pp[j] = 0; gg[j] = 0;
            if (it < items && sg.decode(it, r0, rr[j], pp[j], gg[j], inv_row)) raw[j] = sg.load(rr[j], pp[j], gg[j]);
        }
#pragma unroll
        for (int j = 0; j < PL_IPB; ++j) {
            const int it = base + j * blockDim.x;
            if (it < items) {
                float2 v[CPL];
                IO<TIO>::unpack(raw[j], v);
                int slot = slot0 + (rr[j] - r0);
                while (slot >= sg.nr) slot -= sg.nr;
                float2* dst = band + ((size_t)slot * sg.wp + pp[j]) * LPP + gg[j] * CPL;
#pragma unroll
                for (int i = 0; i < CPL; i += 2) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i].x, v[i].y, v[i + 1].x, v[i + 1].y);
            }
        }
    }
}

// band[r][x] += resize(coarse)(r, x) over the real plane (single-band mode, pass 2)
template <int LPP>
__device__ __forceinline__ void band_upadd(float2* __restrict__ band, const PlaneArgs& a, const float2* __restrict__ coarse,
                                           int hc, int wc, float sy, float sx)
{
    const int nq = a.H * a.W;
    const float inv_w = 1.0f / (float)a.W;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int y = fast_div(q, inv_w);
        const int x = q - y * a.W;
        const UpTap t = up_tap(a.mode, hc, wc, sy, sx, y, x);
        const int slot = mod_pos(y, a.band_rows);
        float2* p = band + ((size_t)slot * a.band_wp + x + PL_P) * LPP + cp;
        *p = add2(*p, up_sample<LPP>(coarse, t, cp));
    }
}

// ---------------- strip convs ----------------
// 5x5 conv of band rows -> output rows [o0, o1) (stride S), TW outputs per thread, no bounds logic:
// the band carries 2 zero columns left and enough right padding for the last partial strip.
template <int LPP, int S, bool TO_GLOBAL, typename TIO>
__device__ __forceinline__ void conv_band(const float2* __restrict__ band, const PlaneArgs& a, const float2* __restrict__ tp,
                                          int o0, int o1, int Wo, float2* __restrict__ dst_lds, TIO* __restrict__ yn, int c0)
{
    constexpr int TW = S == 1 ? PL_TW : PL_TW2;
    constexpr int SPAN = (TW - 1) * S + PL_K;
    const int wp = a.band_wp;
    const int strips = (Wo + TW - 1) / TW;
    const int nq = (o1 - o0) * strips;
    const float inv_strips = 1.0f / (float)strips;
    const int slot_base = mod_pos(o0 * S - PL_P, a.band_rows);
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int dr = fast_div(q, inv_strips);
        const int st = q - dr * strips;
        const int oy = o0 + dr;
        const int ox0 = st * TW;
        int slot = wrap(wrap(slot_base + dr * S, a.band_rows), a.band_rows);
        float2 acc[TW];
        {
            const float2 bias = tp[PL_K * PL_K * LPP + cp];
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[j] = bias;
        }
        // padded column of input ix is ix + 2, so the window of output ox starts at padded column ox*S.
        // Rows are software-pipelined by hand (row u+1 is loaded while row u is multiplied) and the
        // order is pinned: left alone, the scheduler hoists all 5 rows of loads and spills.
        float2 cur[SPAN], nxt[SPAN];
        {
            const float2* row = band + ((size_t)slot * wp + ox0 * S) * LPP + cp;
#pragma unroll
            for (int s = 0; s < SPAN; ++s) cur[s] = row[s * LPP];
        }
#pragma unroll
        for (int u = 0; u < PL_K; ++u) {
            if (u + 1 < PL_K) {
                slot = wrap(slot + 1, a.band_rows);
                const float2* row = band + ((size_t)slot * wp + ox0 * S) * LPP + cp;
#pragma unroll
                for (int s = 0; s < SPAN; ++s) nxt[s] = row[s * LPP];
            }
            float2 tw[PL_K];
#pragma unroll
            for (int i = 0; i < PL_K; ++i) tw[i] = tp[(u * PL_K + i) * LPP + cp];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SPAN; ++s) {
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int tap = s - j * S;
                    if (tap >= 0 && tap < PL_K) acc[j] = fma2(tw[tap], cur[s], acc[j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SPAN; ++s) cur[s] = nxt[s];
        }
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int ox = ox0 + j;
            if (ox < Wo) {
                if constexpr (TO_GLOBAL) IO<TIO>::st2(yn + ((size_t)oy * Wo + ox) * a.C + c0 + 2 * cp, acc[j]);
                else dst_lds[(oy * Wo + ox) * LPP + cp] = acc[j];
            }
        }
    }
}

// LDS plane -> LDS plane, stride 1, TW-wide strips, zero padding by predication (C_l = conv_j(T_l))
template <int LPP>
__device__ __forceinline__ void conv_plane_strips(const float2* __restrict__ src, int hs, int ws, float2* __restrict__ dst,
                                                  const float2* __restrict__ tp)
{
    const int strips = (ws + PL_TW - 1) / PL_TW;
    const int nq = hs * strips;
    const float inv_strips = 1.0f / (float)strips;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int oy = fast_div(q, inv_strips);
        const int ox0 = (q - oy * strips) * PL_TW;
        float2 acc[PL_TW];
        {
            const float2 bias = tp[PL_K * PL_K * LPP + cp];
#pragma unroll
            for (int j = 0; j < PL_TW; ++j) acc[j] = bias;
        }
#pragma unroll
        for (int u = 0; u < PL_K; ++u) {
            const int iy = oy + u - PL_P;
            if (iy < 0 || iy >= hs) continue;
            const float2* row = src + (size_t)iy * ws * LPP + cp;
            float2 tw[PL_K];
#pragma unroll
            for (int i = 0; i < PL_K; ++i) tw[i] = tp[(u * PL_K + i) * LPP + cp];
#pragma unroll
            for (int s = 0; s < PL_TW + PL_K - 1; ++s) {
                const int ix = ox0 + s - PL_P;
                float2 v = make_float2(0.f, 0.f);
                if (ix >= 0 && ix < ws) v = row[ix * LPP];
#pragma unroll
                for (int j = 0; j < PL_TW; ++j) {
                    const int tap = s - j;
                    if (tap >= 0 && tap < PL_K) acc[j] = fma2(tw[tap], v, acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PL_TW; ++j)
            if (ox0 + j < ws) dst[(oy * ws + ox0 + j) * LPP + cp] = acc[j];
    }
}

// LDS plane -> LDS plane, stride 2, one output per thread (the small levels of the down ladder)
template <int LPP>
__device__ __forceinline__ void down_plane(const float2* __restrict__ src, int hs, int ws, float2* __restrict__ dst, int hd, int wd,
                                           const float2* __restrict__ tp)
{
    const int nq = hd * wd;
    const float inv_wd = 1.0f / (float)wd;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int oy = fast_div(q, inv_wd);
        const int ox = q - oy * wd;
        float2 acc = tp[PL_K * PL_K * LPP + cp];
#pragma unroll
        for (int u = 0; u < PL_K; ++u) {
            const int iy = oy * 2 + u - PL_P;
            if (iy < 0 || iy >= hs) continue;
#pragma unroll
            for (int v = 0; v < PL_K; ++v) {
                const int ix = ox * 2 + v - PL_P;
                if (ix < 0 || ix >= ws) continue;
                acc = fma2(tp[(u * PL_K + v) * LPP + cp], src[(iy * ws + ix) * LPP + cp], acc);
            }
        }
        dst[q * LPP + cp] = acc;
    }
}

template <int LPP>
__device__ __forceinline__ void upadd_plane(const float2* __restrict__ coarse, int hc, int wc, float2* __restrict__ fine, int hf, int wf, int mode)
{
    const float sy = (float)hc / (float)hf, sx = (float)wc / (float)wf;
    const int nq = hf * wf;
    const float inv_wf = 1.0f / (float)wf;
    const int cp = threadIdx.x % LPP;
    const int qstep = blockDim.x / LPP;
    for (int q = threadIdx.x / LPP; q < nq; q += qstep) {
        const int y = fast_div(q, inv_wf);
        const int x = q - y * wf;
        const UpTap t = up_tap(mode, hc, wc, sy, sx, y, x);
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
    const float2* coarse; int hc, wc; float sy, sx;
    float2* dst_lds; TIO* yn; int c0;
    int Ho, Wo, Bo, nb;

    // first new input row of band b (rows before first(0) are staged by the prologue)
    __device__ __forceinline__ int first(int b) const { return b * Bo * S + (S == 2 ? 1 : PL_P); }
    __device__ __forceinline__ int row_end(int b) const { const int e = first(b + 1); const int lim = a.H + PL_P; return e < lim ? e : lim; }

    template <int SLOT>
    __device__ __forceinline__ void step(uint4 (&fifo)[PL_D][PL_IPB], int b) const
    {
        if (b >= nb) return;
        __syncthreads();                                  // band b-1's readers are done with the slots we overwrite
        if (S == 1 && b < 8) RCX_STAMP(8 + 3 * b);
        stage_write<LPP, HAS_COARSE, TIO>(sg, band, fifo[SLOT], first(b), row_end(b), coarse, hc, wc, sy, sx, a.mode);
        if (S == 1 && b == 2) RCX_STAMP(40);
        if (b + PL_D < nb) stage_issue<LPP, TIO>(sg, fifo[SLOT], first(b + PL_D), row_end(b + PL_D));
        if (S == 1 && b == 2) RCX_STAMP(41);
        __syncthreads();
        if (S == 1 && b < 8) RCX_STAMP(9 + 3 * b);
        const int o0 = b * Bo, o1 = min(o0 + Bo, Ho);
        conv_band<LPP, S, S == 1, TIO>(band, a, tp, o0, o1, Wo, dst_lds, yn, c0);
        if (S == 1 && b == 2) RCX_STAMP(42);
    }

    __device__ __forceinline__ void run() const
    {
        uint4 fifo[PL_D][PL_IPB];
        // rows [-2, first(0)) synchronously (zero rows + the first real row(s)), then prime the FIFO
        if constexpr (HAS_COARSE) {
            // the prologue rows need resize(coarse) too: route them through issue/write
            stage_issue<LPP, TIO>(sg, fifo[0], -PL_P, first(0));
            stage_write<LPP, true, TIO>(sg, band, fifo[0], -PL_P, first(0), coarse, hc, wc, sy, sx, a.mode);
        } else {
            stage_direct<LPP, TIO>(sg, band, -PL_P, first(0));
        }
        stage_issue<LPP, TIO>(sg, fifo[0], first(0), row_end(0));
        if (1 < nb) stage_issue<LPP, TIO>(sg, fifo[1], first(1), row_end(1));
        if (2 < nb) stage_issue<LPP, TIO>(sg, fifo[2], first(2), row_end(2));
        static_assert(PL_D == 3, "FIFO priming is written for depth 3");
        for (int b = 0; b < nb; b += PL_D) {
            step<0>(fifo, b);
            step<1>(fifo, b + 1);
            step<2>(fifo, b + 2);
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
    auto taps_of = [&](int i) { return taps + (size_t)i * PL_TAPROWS * LPP; };
    const Stage<LPP, TIO> sg{xn, c0, a.H, a.W, a.C, a.band_wp, a.band_rows};

    // all (L+2) tap sets of this channel block, once; first read is behind the first staging barrier
    RCX_STAMP(0);
    preload_taps<LPP>(taps, wpack, a.has_bias ? bpack : nullptr, a.C, c0, L + 2);

    if (L >= 1) {
        // ---- pass 1: F_1 = down(x) ----
        float2* F1 = lds + a.f_off[1];
        if (a.single) {
            stage_direct<LPP, TIO>(sg, band, -PL_P, a.H + PL_P);
            __syncthreads();
            conv_band<LPP, 2, false, TIO>(band, a, taps_of(0), 0, a.h[1], a.w[1], F1, nullptr, c0);
        } else {
            const Pass<LPP, 2, false, TIO> p1{a, sg, band, taps_of(0), nullptr, 0, 0, 0.f, 0.f, F1, nullptr, c0,
                                              a.h[1], a.w[1], a.B1, (a.h[1] + a.B1 - 1) / a.B1};
            p1.run();
        }
        __syncthreads();
        RCX_STAMP(1);
        // ---- ladder: F_{l+1} = down(F_l) ----
        for (int l = 1; l < L; ++l) {
            down_plane<LPP>(lds + a.f_off[l], a.h[l], a.w[l], lds + a.f_off[l + 1], a.h[l + 1], a.w[l + 1], taps_of(0));
            __syncthreads();
        }
        RCX_STAMP(2);
        // ---- up recursion, coarsest first: C_l = conv_j(T_l); T_{l-1} = F_{l-1} + resize(C_l) ----
        for (int l = L, j = 0; l >= 1; --l, ++j) {
            if (l == 1) RCX_STAMP(3);
            conv_plane_strips<LPP>(lds + a.f_off[l], a.h[l], a.w[l], lds + a.c_off[l], taps_of(1 + j));
            __syncthreads();
            if (l > 1) {
                upadd_plane<LPP>(lds + a.c_off[l], a.h[l], a.w[l], lds + a.f_off[l - 1], a.h[l - 1], a.w[l - 1], a.mode);
                __syncthreads();
            }
        }
    }
    RCX_STAMP(4);
    // ---- pass 2: y = conv_L(x + resize(C_1)) ----
    const float2* C1 = L >= 1 ? lds + a.c_off[1] : nullptr;
    const int hc = L >= 1 ? a.h[1] : 1, wc = L >= 1 ? a.w[1] : 1;
    const float sy = (float)hc / (float)a.H, sx = (float)wc / (float)a.W;
    if (L >= 1 && a.single) {
        band_upadd<LPP>(band, a, C1, hc, wc, sy, sx);
        __syncthreads();
        conv_band<LPP, 1, true, TIO>(band, a, taps_of(1 + L), 0, a.H, a.W, nullptr, yn, c0);
    } else if (L >= 1) {
        const Pass<LPP, 1, true, TIO> p2{a, sg, band, taps_of(1 + L), C1, hc, wc, sy, sx, nullptr, yn, c0,
                                         a.H, a.W, a.B2, (a.H + a.B2 - 1) / a.B2};
        p2.run();
    } else {
        const Pass<LPP, 1, false, TIO> p2{a, sg, band, taps_of(1 + L), nullptr, 0, 0, 0.f, 0.f, nullptr, yn, c0,
                                          a.H, a.W, a.B2, (a.H + a.B2 - 1) / a.B2};
        p2.run();
    }
    RCX_STAMP(5);
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

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

static void fill_args(PlaneArgs& a, int N, int C, int H, int W, int level, int lpp, int B2cand, size_t& lds_bytes)
{
    a.N = N; a.C = C; a.H = H; a.W = W; a.level = level;
    a.h[0] = H; a.w[0] = W;
    size_t pix = 0;                                // LDS pixels (each lpp float2 wide)
    for (int l = 1; l <= level; ++l) {
        a.h[l] = down_size5(a.h[l - 1]); a.w[l] = down_size5(a.w[l - 1]);
        a.f_off[l] = (int)(pix * lpp);
        pix += (size_t)a.h[l] * a.w[l];
    }
    // C region: C_1 from its start; C_2, C_3, ... packed from its start too (all dead before C_1 is written)
    const size_t cbase = pix;
    size_t csmall = 0;
    for (int l = 2; l <= level; ++l) { a.c_off[l] = (int)((cbase + csmall) * lpp); csmall += (size_t)a.h[l] * a.w[l]; }
    size_t creg = 0;
    if (level >= 1) { a.c_off[1] = (int)(cbase * lpp); creg = (size_t)a.h[1] * a.w[1]; }
    if (csmall > creg) creg = csmall;
    pix += creg;
    a.band_off = (int)(pix * lpp);
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
    pix += (size_t)a.band_rows * wp;
    a.taps_off = (int)(pix * lpp);
    pix += (size_t)(level + 2) * PL_TAPROWS;
    lds_bytes = pix * lpp * sizeof(float2);
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
    // enough threads that one band's staging fits PL_IPB 16-byte items per thread, and that the
    // widest strip conv has about one item per thread
    const int stage_items = band_stage_items(a, lpp, cpl);
    const int conv_items = (a.single ? a.H : a.B2) * ((a.W + PL_TW - 1) / PL_TW) * lpp;
    int nt = 64;
    while (nt < PL_NT && (conv_items > nt || (!a.single && stage_items > nt * PL_IPB))) nt *= 2;
    if (nt < lpp) nt = lpp;
    return nt;
}

PlanePlan plan_plane(int N, int C, int H, int W, int level, int k, int dtype)
{
    PlanePlan p{};
    p.ok = false;
    if (k != PL_K || level < 0 || level > PL_MAXL || (C % 8) != 0) return p;
    if ((long long)H * W > (1 << 18)) return p;           // fast_div range
    const size_t LDS_CU = 160 * 1024;
    const int cpl = dtype == 1 ? 4 : 2;                    // channel pairs per 16-byte chunk
    const int force_lpp = env_int("RCX_PLANE_LPP", 0), force_b2 = env_int("RCX_PLANE_B2", 0);
    const int force_nt = env_int("RCX_PLANE_NT", 0);
    static const int lpps[] = {32, 16, 8, 4};
    static const int b2s[] = {1 << 20, 8, 4, 2};          // first candidate: the whole plane in the band
    PlanePlan best{};
    best.ok = false;
    double best_score = -1.0;
    for (int lpp : lpps) {
        if (C % (2 * lpp) || lpp < cpl) continue;
        if (force_lpp && lpp != force_lpp) continue;
        for (int B2 : b2s) {
            if (force_b2 && (B2 > 64 ? H : B2) != force_b2) continue;
            if (B2 < 64 && B2 >= H) continue;              // same as the whole-plane candidate
            PlaneArgs a{};
            size_t bytes = 0;
            fill_args(a, N, C, H, W, level, lpp, B2, bytes);
            if (bytes > LDS_CU) continue;
            a.nblk = C / (2 * lpp);
            PlanePlan cand{};
            cand.ok = true; cand.lpp = lpp; cand.lds_bytes = bytes; cand.args = a;
            cand.nt = force_nt ? force_nt : pick_threads(a, lpp, cpl);
            if (cand.nt < 64 || cand.nt > PL_NT || cand.nt % 64 || cand.nt % lpp) cand.nt = 256;
            if (!a.single && band_stage_items(a, lpp, cpl) > cand.nt * PL_IPB) continue;   // FIFO slot too small
            // score: resident waves per CU (latency hiding across workgroups), with a bonus for wide
            // channel blocks (fuller HBM/L2 lines) and a penalty when the grid cannot fill the chip
            int wg_cu = (int)(LDS_CU / bytes);
            const int by_waves = 16 / (cand.nt / 64);      // <= 16 waves/CU at ~128+ VGPRs
            if (wg_cu > by_waves) wg_cu = by_waves;
            if (wg_cu < 1) wg_cu = 1;
            const long long wgs = (long long)N * a.nblk;
            double waves = (double)wg_cu * (cand.nt / 64);
            double fill = (double)wgs / (256.0 * wg_cu);
            if (fill > 1.0) fill = 1.0;
            double score = waves * fill * (lpp >= 16 ? 1.25 : (lpp == 8 ? 1.0 : 0.7)) * (a.single ? 1.2 : 1.0);
            if (score > best_score) { best_score = score; best = cand; }
        }
    }
    return best;
}

template <int LPP, typename TIO>
static hipError_t launch_plane_t(const void* x, void* y, const float* wpack, const float* bpack, const PlanePlan& p, int mode, hipStream_t s)
{
    auto kfn = k_recconv_plane<LPP, TIO>;
    if (p.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    PlaneArgs a = p.args;
    a.has_bias = bpack != nullptr;
    a.mode = mode;
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
