// Channel-per-lane, TILED RecConv2d for the two large blocks of RecNeXt at 224x224 (model/recnext.py:24-34):
//   56x56 / level 4 (stage 0) and 28x28 / level 3 (stage 1).
//
// Layout.  A LANE owns one channel of one 14x14 tile of the full-resolution plane; a workgroup owns one image x one block of
// channels and all T x T tiles of its plane (T = 4: 56x56, T = 2: 28x28):
//   T = 4: 8 waves, a wave = 32 channels x 2 tiles (lanes 0-31 / 32-63 own the tiles (tr, tc) and (tr, tc + 2): same row,
//          same column parity, so every row quantity and every parity is wave-uniform and only the column origin and the two
//          image-edge flags differ per lane); 64 contiguous bytes per pixel and half-wave in NHWC bf16;
//   T = 2: 4 waves, a wave = 64 channels x 1 tile (128 contiguous bytes per pixel).
// Nothing is ever exchanged between lanes: no DPP (rcx_lanes.h: a DPP move puts the SIMD into its slow issue mode), and the
// level-0 work -- 80 % of the FMAs -- never touches LDS: both passes stream x one row at a time straight from global memory
// into registers (hand-issued buffer loads that run AHEAD rows in front; out-of-image halo columns are out-of-range buffer
// offsets and read 0) and are input-row stationary on v_pk_fma_f32 pairs exactly as rcx_cpl14.hip (column pairs for the
// stride-1 convs, tap pairs for the stride-2 conv).  What tiles must share -- the planes of level >= 1, at most 28x28 -- lives in
// LDS as float32 [pixel][channel of the block], lane-contiguous (conflict-free 32-bit accesses):
//   pass 1   F1 tile (7x7) = down(x tile + halo)                    -> LDS                                        (:27-29)
//   chain    F2..FL = down ladder, C_L .. C_2 = conv(F_l + resize(C_{l+1})) on the small planes, row segments dealt over
//            the T*T tile-lanes ("pieces", gathered from LDS; rows outside a plane are redirected to a zero row)    (:27-33)
//   level 1  T1 = F1 + resize(C2) and C1 = conv(T1) per tile (halo read from the neighbours' LDS pixels)          (:31-33)
//   pass 2   y tile = conv(x + resize(C1)) with an 18x18 input window, five accumulator rows in flight             (:34)
// Same arithmetic as the other schedules: float32 throughout, one rounding at the final store.  The exact-2x bilinear steps use
// the closed form of ATen's index arithmetic (weights 0.25 / 0.75, clamped borders: at a clamped border both taps read the same
// pixel, 0.25 v + 0.75 v instead of ATen's v -- one ulp); the 4 -> 7 step uses ATen's float formulas (rcx_common.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "rcx_common.h"
#include "rcx_lanes.h"
#include "rcx_launch.h"

namespace rcx {
namespace cpt {

using lanes::f32x2;
using lanes::IC;
using lanes::sfor;
using lanes::vtab;
using lanes::VT;

#define CPT_FENCE __builtin_amdgcn_sched_barrier(0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* gcptr;
typedef __attribute__((address_space(1))) char* gptr;

__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 shift1(f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 1, 2); }
__device__ __forceinline__ gcptr opaque(gcptr p) { asm volatile("" : "+s"(p)); return p; }
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
template <int A> __device__ __forceinline__ void pin(f32x2 (&v)[A]) {
#pragma unroll
    for (int i = 0; i < A; ++i) pin(v[i]);
}
template <int A> __device__ __forceinline__ void pin(float (&v)[A]) {
#pragma unroll
    for (int i = 0; i < A; ++i) pin(v[i]);
}

// ---- x rows: hand-issued buffer loads (the compiler would sink them to their first use and the prefetch distance collapses;
// rcx_cpl14.hip).  address = image base (descriptor) + soff (uniform: row and column) + voff (this lane: tile column origin and
// channel, or 0x80000000 = out of range -> the load returns 0: the zero padding left and right of the image)
template <typename TIO> struct BufLd;
// The scalar offset the load reads is produced by an SALU add INSIDE the statement: an SGPR operand handed in from outside may
// have just been reloaded from a spill lane by v_readlane_b32 (a VALU write), and a vector-memory instruction that reads an SGPR
// within 5 wait states of a VALU write to it sees the old value -- the hazard recognizer does not look inside inline asm.
// (s_add_i32 writes SCC: declared, or a scalar select scheduled behind the statement reads the wrong condition.)
template <> struct BufLd<float> {
    static __device__ __forceinline__ void ld(uint32_t& dst, unsigned voff, i32x4 rsrc, int rb, int koff)
    {
        int tmp;
        asm volatile("s_add_i32 %0, %3, %4\n\tbuffer_load_dword %1, %2, %5, %0 offen" : "=&s"(tmp), "=v"(dst) : "v"(voff), "s"(rb), "s"(koff), "s"(rsrc) : "scc");
    }
};
template <> struct BufLd<bf16_t> {
    // bf16 -> float32 without an instruction: the D16 "hi" load fills the upper half and zeroes the lower (tools/ubench/d16_probe.hip)
    static __device__ __forceinline__ void ld(uint32_t& dst, unsigned voff, i32x4 rsrc, int rb, int koff)
    {
        int tmp;
        asm volatile("s_add_i32 %0, %3, %4\n\tbuffer_load_short_d16_hi %1, %2, %5, %0 offen" : "=&s"(tmp), "=v"(dst) : "v"(voff), "s"(rb), "s"(koff), "s"(rsrc) : "scc");
    }
};

// first touch of a row of 18 hand-issued loads: wait until at most PENDING younger memory operations are outstanding (stores
// issued in between only make the true count larger: the wait can come out longer than necessary, never shorter)
template <int PENDING>
__device__ __forceinline__ void pin_row(uint32_t (&v)[18])
{
    asm volatile("s_waitcnt vmcnt(%18)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]),
                 "+v"(v[17]) : "n"(PENDING));
}

template <typename TIO> struct PixSt;
template <> struct PixSt<float> {
    typedef f32x2 packed;
    static __device__ __forceinline__ packed prep(f32x2 v) { return v; }
    static __device__ __forceinline__ void st(gcptr p, packed v, int half) { *(__attribute__((address_space(1))) float*)(p) = half ? v.y : v.x; }
};
template <> struct PixSt<bf16_t> {
    typedef uint32_t packed;
    static __device__ __forceinline__ packed prep(f32x2 v)
    {
        uint32_t pk;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v.x), "v"(v.y));
        return pk;
    }
    static __device__ __forceinline__ void st(gcptr p, packed v, int half)
    {
        *(__attribute__((address_space(1))) bf16_t*)(p) = half ? (bf16_t)(v >> 16) : (bf16_t)v;
    }
};

// the 25 taps of one conv for this lane's channel as three register pairs per tap row: (w0,w1) (w2,w3) (w4,0)
struct Taps {
    f32x2 p[5][3];
    float bias;
    __device__ __forceinline__ float at(int u, int v) const { return (v & 1) ? p[u][v >> 1].y : p[u][v >> 1].x; }
};

__device__ __forceinline__ void load_taps(Taps& t, const float* __restrict__ wpack, const float* __restrict__ bpack, int conv, int C, int c, int has_bias)
{
    const gcptr wb = (gcptr)(wpack + (size_t)conv * 25 * C);
    const unsigned vow = (unsigned)c * 4u;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const gcptr tb = opaque(wb + (size_t)(u * 5 + v) * C * 4);     // uniform base (SGPR pair) + this lane's 32-bit offset
            const float w = *reinterpret_cast<const __attribute__((address_space(1))) float*>(tb + vow);
            if (v & 1) t.p[u][v >> 1].y = w;
            else t.p[u][v >> 1].x = w;
        }
        t.p[u][2].y = 0.f;
    }
    t.bias = has_bias ? bpack[(size_t)conv * C + c] : 0.f;
}

constexpr int plane_size(int T, int l) { return l == 0 ? 14 * T : (plane_size(T, l - 1) + 1) / 2; }

// ---- pieces: one output row segment of a small plane, gathered from LDS.  L* point at this lane's channel column; a pixel is
// PIXF floats.  Rows outside the plane read the zero row; columns outside are compile-time zeros.
// stride-2 conv: outputs COL0 .. COL0+NOUT-1 of row `orow` of down(PIN x PIN)
template <int PIN, int COL0, int NOUT, int PIXF>
__device__ __forceinline__ void down_piece(const float* Lin, const float* Lzero, int orow, const Taps& t, float (&out)[NOUT])
{
    f32x2 acc[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) acc[i] = f32x2{t.bias, 0.f};
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int r = 2 * orow + u - 2;
        const float* rp = ((unsigned)r < (unsigned)PIN) ? Lin + r * (PIN * PIXF) : Lzero;
        f32x2 in[NOUT + 2];
#pragma unroll
        for (int k = 0; k < NOUT + 2; ++k) {
            const int c0 = 2 * COL0 - 2 + 2 * k, c1 = c0 + 1;
            in[k].x = (c0 >= 0 && c0 < PIN) ? rp[c0 * PIXF] : 0.f;
            in[k].y = (c1 >= 0 && c1 < PIN && k < NOUT + 1) ? rp[c1 * PIXF] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NOUT; ++i) acc[i] = pfma(in[i], t.p[u][0], acc[i]);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) acc[i] = pfma(in[i + 1], t.p[u][1], acc[i]);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) acc[i].x = fmaf(in[i + 2].x, t.p[u][2].x, acc[i].x);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) out[i] = acc[i].x + acc[i].y;
}

// stride-1 conv: outputs COL0 .. COL0+NOUT-1 of row `orow` of a P x P plane, as pairs (the odd tail element is not an output)
template <int P, int COL0, int NOUT, int PIXF>
__device__ __forceinline__ void conv_piece(const float* Lin, const float* Lzero, int orow, const Taps& t, f32x2 (&acc)[(NOUT + 1) / 2])
{
    constexpr int NP = (NOUT + 1) / 2;
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = splat(t.bias);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int r = orow + u - 2;
        const float* rp = ((unsigned)r < (unsigned)P) ? Lin + r * (P * PIXF) : Lzero;
        f32x2 in[NP + 2], odd[NP + 1];
#pragma unroll
        for (int k = 0; k < NP + 2; ++k) {
            const int c0 = COL0 - 2 + 2 * k, c1 = c0 + 1;
            in[k].x = (c0 >= 0 && c0 < P && c0 <= COL0 + NOUT + 1) ? rp[c0 * PIXF] : 0.f;
            in[k].y = (c1 >= 0 && c1 < P && c1 <= COL0 + NOUT + 1) ? rp[c1 * PIXF] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NP + 1; ++j) odd[j] = shift1(in[j], in[j + 1]);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] = pfma(in[j], splat(t.at(u, 0)), acc[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] = pfma(odd[j], splat(t.at(u, 1)), acc[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] = pfma(in[j + 1], splat(t.at(u, 2)), acc[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] = pfma(odd[j + 1], splat(t.at(u, 3)), acc[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] = pfma(in[j + 2], splat(t.at(u, 4)), acc[j]);
    }
}

// T = F + resize(C): row `orow`, columns COL0 .. COL0+NOUT-1 of the P x P plane Lf, in place; C is PC x PC.  The row index is
// this lane's (ATen's float formulas at run time), the columns are compile-time table entries.
template <int MODE, int PC, int P, int COL0, int NOUT, int PIXF>
__device__ __forceinline__ void tform_piece(float* Lf, const float* Lc, int orow, bool active)
{
    constexpr float scale = (float)PC / (float)P;
    int i0, i1;
    float lam;
    if (MODE == 1) { i0 = i1 = nearest_src(orow, PC, scale); lam = 0.f; }
    else { const Lerp lr = bilinear_src(orow, PC, scale); i0 = lr.i0; i1 = lr.i1; lam = lr.lam; }
    const float* r0 = Lc + i0 * (PC * PIXF);
    const float* r1 = Lc + i1 * (PC * PIXF);
    constexpr int cmin = vtab(MODE, PC, P, COL0).i0, cmax = vtab(MODE, PC, P, COL0 + NOUT - 1).i1;
    float V[cmax - cmin + 1];
    const float l0 = 1.f - lam;
#pragma unroll
    for (int c = cmin; c <= cmax; ++c) V[c - cmin] = MODE == 1 ? r0[c * PIXF] : fmaf(lam, r1[c * PIXF], l0 * r0[c * PIXF]);
    float* fp = Lf + (orow * P + COL0) * PIXF;
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        const VT h = vtab(MODE, PC, P, COL0 + j);
        const float up = (MODE == 1 || h.i0 == h.i1) ? V[h.i0 - cmin] : fmaf(h.l, V[h.i1 - cmin], (1.f - h.l) * V[h.i0 - cmin]);
        const float f = fp[j * PIXF];
        if (active) fp[j * PIXF] = f + up;
    }
}

// exact-2x step, source index relative to the base column b and weight of the second tap, for destination column c of a run
// that starts at an even (par = 0) or odd (par = 1) absolute position.  bilinear: b = (d0 - 1) >> 1, nearest: b = d0 >> 1
struct Rel { int idx; float l; };
constexpr Rel rel2(int mode, int par, int c)
{
    if (mode == 1) return Rel{par ? (c + 1) / 2 : c / 2, 0.f};
    if (par == 0) return Rel{(c & 1) ? (c + 1) / 2 : c / 2, (c & 1) ? 0.25f : 0.75f};
    return Rel{(c & 1) ? (c - 1) / 2 : c / 2, (c & 1) ? 0.75f : 0.25f};
}

template <int T_, int HALVES, int MODE, typename TIO>
struct Geo {
    static constexpr int T = T_;
    static constexpr int NL = T == 4 ? 4 : 3;
    static constexpr int NW = T * T / HALVES;
    static constexpr int NT = NW * 64;
    static constexpr int PIXF = 64 / HALVES;
    static constexpr int NWORK = T * T;
    static constexpr int P0 = 14 * T, P1 = 7 * T, P2 = plane_size(T, 2), P3 = plane_size(T, 3), P4 = plane_size(T, 4);
    // LDS, in pixels: zero row | guard | L1 | guard | L2 | L3 | L4
    static constexpr int ZR = P1;
    static constexpr int O1 = ZR + 2;
    static constexpr int O2 = O1 + P1 * P1 + 2;
    static constexpr int O3 = O2 + P2 * P2;
    static constexpr int O4 = O3 + P3 * P3;
    static constexpr int NPIX = O4 + (NL >= 4 ? P4 * P4 : 0);
    static constexpr int LDS_BYTES = NPIX * PIXF * 4;
};

template <int T, int HALVES, int MODE, typename TIO>
__global__ __launch_bounds__(T * T / HALVES * 64, T == 4 ? 1 : 2)
void k_recconv_cpt(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                   int N, int C, int has_bias)
{
    using G = Geo<T, HALVES, MODE, TIO>;
    constexpr int NL = G::NL, PIXF = G::PIXF, NWORK = G::NWORK, P0 = G::P0, P1 = G::P1, P2 = G::P2, P3 = G::P3, P4 = G::P4;
    constexpr int ESZ = (int)sizeof(TIO);
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int nb = (C + PIXF - 1) / PIXF;
    unsigned b = blockIdx.x;
    const unsigned GD = gridDim.x;
    if ((GD & 7u) == 0) b = (b & 7u) * (GD >> 3) + (b >> 3);            // XCD-aware order: each XCD gets a contiguous run of (image, channel block) units
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= N) return;

    const int tid = (int)threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = HALVES == 2 ? (lane >> 5) : 0;
    const int ch = lane & (PIXF - 1);
    const int tr = T == 4 ? (w >> 1) : (w >> 1);
    const int tcb = w & 1;
    const int tc = HALVES == 2 ? tcb + 2 * h : tcb;                      // per lane (HALVES == 2) / uniform
    const int q = tr * T + tc;                                          // this tile-lane's worker id
    const bool ledge = tc == 0, redge = tc == T - 1;
    const int c = cb * PIXF + ch;
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;
    const int pix = C * ESZ;                                            // bytes between horizontally adjacent pixels

    float* const L = lds + ch;
    const float* const Lzero = L;
    float* const L1 = L + G::O1 * PIXF;
    float* const L2 = L + G::O2 * PIXF;
    float* const L3 = L + G::O3 * PIXF;
    float* const L4 = L + G::O4 * PIXF;

    // ---- zero the whole LDS image (zero row, guards; and every later read is of finite data)
    for (int i = tid; i < G::LDS_BYTES / 16; i += G::NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // x image as a raw buffer: base, num_records = bytes of the image (offsets past it read 0)
    const char* ximg = reinterpret_cast<const char*>(x) + (size_t)n * P0 * P0 * pix;
    i32x4 rsrc;
    {
        const unsigned long long a = (unsigned long long)ximg;
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
        rsrc.z = P0 * P0 * pix;
        rsrc.w = 0x00020000;
    }
    const unsigned OOB = 0x80000000u;
    const unsigned voffM = (unsigned)((14 * tc) * pix + cc * ESZ);
    const unsigned voffL = ledge ? OOB : voffM - 2u * (unsigned)pix;      // columns -2, -1 of the tile
    const unsigned voffR = redge ? OOB : voffM + 14u * (unsigned)pix;     // columns 14, 15
    // row r (tile-local, -2 .. 15), all 18 columns; rows outside the image are redirected to a valid row (loaded, not used)
    auto load_row = [&](uint32_t (&raw)[18], int r) {
        int ar = 14 * tr + r;
        ar = ar < 0 ? 0 : (ar > P0 - 1 ? P0 - 1 : ar);
        const int rb = __builtin_amdgcn_readfirstlane(ar * (P0 * pix));     // uniform by construction; the asm below needs it in an SGPR
        BufLd<TIO>::ld(raw[0], voffL, rsrc, rb, 0);
        BufLd<TIO>::ld(raw[1], voffL, rsrc, rb, pix);
#pragma unroll
        for (int k = 0; k < 14; ++k) BufLd<TIO>::ld(raw[2 + k], voffM, rsrc, rb, k * pix);
        BufLd<TIO>::ld(raw[16], voffR, rsrc, rb, 0);
        BufLd<TIO>::ld(raw[17], voffR, rsrc, rb, pix);
    };
    auto row_valid = [&](int r) -> bool { const int ar = 14 * tr + r; return ar >= 0 && ar < P0; };   // uniform

    Taps td;
    load_taps(td, wpack, bpack, 0, C, cc, has_bias);
    __syncthreads();

    // ================= pass 1: F1 tile = down(x), rows -2 .. 14 of the tile, input-row stationary (tap pairs) =================
    float f1[7][7];                                          // this lane's F1 tile stays in registers until T1 is formed
    {
        constexpr int AHEAD = 2, R0 = -2, NR = 17;
        uint32_t raw[NR][18];
        f32x2 facc[3][7];
        sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], R0 + decltype(rc)::value); });
        sfor<NR>([&](auto rc) {
            constexpr int ri = decltype(rc)::value, r = R0 + ri;
            if constexpr (ri + AHEAD < NR) load_row(raw[ri + AHEAD], r + AHEAD);
            pin_row<18 * (NR - 1 - ri < AHEAD ? NR - 1 - ri : AHEAD)>(raw[ri]);
            f32x2 xr[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) xr[k] = f32x2{__uint_as_float(raw[ri][2 * k]), __uint_as_float(raw[ri][2 * k + 1])};
            const bool rv = row_valid(r);
#pragma unroll
            for (int o = 0; o < 7; ++o) {
                const int u = r - 2 * o + 2;
                if (u < 0 || u > 4) continue;
                f32x2(&a)[7] = facc[o % 3];
                if (u == 0) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) a[i] = f32x2{td.bias, 0.f};
                }
                if (rv) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) a[i] = pfma(xr[i], td.p[u][0], a[i]);
#pragma unroll
                    for (int i = 0; i < 7; ++i) a[i] = pfma(xr[i + 1], td.p[u][1], a[i]);
#pragma unroll
                    for (int i = 0; i < 7; ++i) a[i].x = fmaf(xr[i + 2].x, td.p[u][2].x, a[i].x);
                }
                if (u == 4) {
                    float* dst = L1 + ((7 * tr + o) * P1 + 7 * tc) * PIXF;
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        f1[o][i] = a[i].x + a[i].y;
                        dst[i * PIXF] = f1[o][i];
                    }
                    pin(f1[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < 7; ++o) if (r - 2 * o + 2 >= 0 && r - 2 * o + 2 < 4) pin(facc[o % 3]);
            CPT_FENCE;
        });
    }
    __syncthreads();

    // ================= chain: the small planes, pieces dealt over the T*T tile-lanes =================
    // conv j of the pack: 0 = down, 1 + (NL - l) = the conv of level l, 1 + NL = the final conv
    constexpr int PL[5] = {P0, P1, P2, P3, P4};
    float* const LP[5] = {nullptr, L1, L2, L3, L4};
    // piece rounds of a P-wide plane: P == 14 (16 workers): two rounds = the two 7-wide column segments, row = q; else full rows,
    // row = q + NWORK * round
    auto for_pieces = [&](auto pc, auto&& f) {
        constexpr int P = decltype(pc)::value;
        if constexpr (P == 14) {
            static_assert(NWORK == 16, "14-wide piece planes are dealt over 16 workers");
            const bool act = q < 14;
            const int row = act ? q : 0;
            f(IC<0>{}, IC<0>{}, IC<7>{}, row, act);
            f(IC<1>{}, IC<7>{}, IC<7>{}, row, act);
        } else {
            constexpr int RNDS = (P + NWORK - 1) / NWORK;
            sfor<RNDS>([&](auto rc) {
                constexpr int rnd = decltype(rc)::value;
                const int rr = q + NWORK * rnd;
                const bool act = rr < P;
                f(rc, IC<0>{}, IC<P>{}, act ? rr : 0, act);
            });
        }
    };
    // down ladder: F_l = down(F_{l-1}), l = 2 .. NL
    sfor<NL - 1>([&](auto lc) {
        constexpr int l = 2 + decltype(lc)::value;
        constexpr int PIN = PL[l - 1], PO = PL[l];
        for_pieces(IC<PO>{}, [&](auto, auto col0c, auto noutc, int row, bool act) {
            constexpr int COL0 = decltype(col0c)::value, NOUT = decltype(noutc)::value;
            float out[NOUT];
            down_piece<PIN, COL0, NOUT, PIXF>(LP[l - 1], Lzero, row, td, out);
            if (act) {
                float* dst = LP[l] + (row * PO + COL0) * PIXF;
#pragma unroll
                for (int i = 0; i < NOUT; ++i) dst[i * PIXF] = out[i];
            }
        });
        __syncthreads();
    });
    // up recursion on the piece planes: l = NL .. 2: T_l = F_l + resize(C_{l+1}) in place (l < NL), C_l = conv(T_l) in place
    sfor<NL - 1>([&](auto lc) {
        constexpr int l = NL - decltype(lc)::value;
        constexpr int P = PL[l];
        Taps tc_;
        load_taps(tc_, wpack, bpack, 1 + (NL - l), C, cc, has_bias);
        if constexpr (l < NL) {
            constexpr int PC = PL[l + 1];
            for_pieces(IC<P>{}, [&](auto, auto col0c, auto noutc, int row, bool act) {
                tform_piece<MODE, PC, P, decltype(col0c)::value, decltype(noutc)::value, PIXF>(LP[l], LP[l + 1], row, act);
            });
            __syncthreads();
        }
        constexpr int RN = P == 14 ? 2 : (P + NWORK - 1) / NWORK;
        f32x2 res[RN][4];
        for_pieces(IC<P>{}, [&](auto rc, auto col0c, auto noutc, int row, bool) {
            constexpr int COL0 = decltype(col0c)::value, NOUT = decltype(noutc)::value;
            f32x2 acc[(NOUT + 1) / 2];
            conv_piece<P, COL0, NOUT, PIXF>(LP[l], Lzero, row, tc_, acc);
#pragma unroll
            for (int j = 0; j < (NOUT + 1) / 2; ++j) res[decltype(rc)::value][j] = acc[j];
        });
        __syncthreads();                                     // every read of T_l is done: C_l may replace it
        for_pieces(IC<P>{}, [&](auto rc, auto col0c, auto noutc, int row, bool act) {
            constexpr int COL0 = decltype(col0c)::value, NOUT = decltype(noutc)::value;
            if (act) {
                float* dst = LP[l] + (row * P + COL0) * PIXF;
#pragma unroll
                for (int i = 0; i < NOUT; ++i) dst[i * PIXF] = (i & 1) ? res[decltype(rc)::value][i >> 1].y : res[decltype(rc)::value][i >> 1].x;
            }
        });
        __syncthreads();
    });

    // ================= level 1, per tile: T1 = F1 + resize(C2) (exact 2x), C1 = conv(T1) =================
    Taps t1;
    load_taps(t1, wpack, bpack, NL, C, cc, has_bias);        // conv of level 1 = pack 1 + (NL - 1)
    {
        // columns: run of 7 starting at absolute column 7*tc (parity uniform), source columns b .. b+4 of C2, clamped
        const int d0 = 7 * tc;
        const int bcol = MODE == 1 ? (d0 >> 1) : ((d0 - 1) >> 1);
        int cofs[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int cx = bcol + k;
            cx = cx < 0 ? 0 : (cx > P2 - 1 ? P2 - 1 : cx);
            cofs[k] = cx * PIXF;
        }
        const int cpar = __builtin_amdgcn_readfirstlane(d0 & 1);
        auto form = [&](auto parc) {
            constexpr int PAR = decltype(parc)::value;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int dr = 7 * tr + r;                      // uniform
                int i0, i1;
                float lam;
                if (MODE == 1) { i0 = i1 = dr >> 1; lam = 0.f; }
                else if (dr & 1) { i0 = (dr - 1) >> 1; i1 = i0 + 1; lam = 0.25f; }
                else { i0 = (dr >> 1) - 1; i1 = i0 + 1; lam = 0.75f; }
                i0 = i0 < 0 ? 0 : (i0 > P2 - 1 ? P2 - 1 : i0);
                i1 = i1 < 0 ? 0 : (i1 > P2 - 1 ? P2 - 1 : i1);
                const float* r0 = L2 + i0 * (P2 * PIXF);
                const float* r1 = L2 + i1 * (P2 * PIXF);
                float V[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) V[k] = MODE == 1 ? r0[cofs[k]] : fmaf(lam, r1[cofs[k]], (1.f - lam) * r0[cofs[k]]);
                sfor<7>([&](auto cic) {
                    constexpr int cI = decltype(cic)::value;
                    constexpr Rel rl = rel2(MODE, PAR, cI);
                    const float up = MODE == 1 ? V[rl.idx] : fmaf(rl.l, V[rl.idx + 1], (1.f - rl.l) * V[rl.idx]);
                    f1[r][cI] += up;
                });
            }
        };
        if (cpar) form(IC<1>{}); else form(IC<0>{});
        float* dst = L1 + ((7 * tr) * P1 + 7 * tc) * PIXF;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int cI = 0; cI < 7; ++cI) dst[(r * P1 + cI) * PIXF] = f1[r][cI];
    }
    __syncthreads();
    // halo masks of the tile (per lane): columns outside the plane contribute nothing
    const float lmask = ledge ? 0.f : 1.f, rmask = redge ? 0.f : 1.f;
    {
        // C1 tile, input-row stationary over T1 rows -2 .. 8, columns -2 .. 8 (the guards before and after the plane make every
        // address valid; what a masked column reads is finite)
        f32x2 c1[7][4];
#pragma unroll
        for (int o = 0; o < 7; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) c1[o][j] = splat(t1.bias);
        const float* base = L1 + ((7 * tr) * P1 + 7 * tc) * PIXF;
#pragma unroll
        for (int t = -2; t <= 8; ++t) {
            const int ar = 7 * tr + t;
            if (ar >= 0 && ar < P1) {                        // uniform
                const float* rp = base + t * (P1 * PIXF);
                f32x2 in[6], odd[5];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    in[k].x = rp[(2 * k - 2) * PIXF];
                    in[k].y = k < 5 ? rp[(2 * k - 1) * PIXF] : 0.f;
                }
                in[0] = in[0] * splat(lmask);
                in[4].y *= rmask;                            // column 7
                in[5].x *= rmask;                            // column 8
#pragma unroll
                for (int j = 0; j < 5; ++j) odd[j] = shift1(in[j], in[j + 1]);
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int o = t - u + 2;
                    if (o < 0 || o > 6) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(in[j], splat(t1.at(u, 0)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(odd[j], splat(t1.at(u, 1)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(in[j + 1], splat(t1.at(u, 2)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(odd[j + 1], splat(t1.at(u, 3)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(in[j + 2], splat(t1.at(u, 4)), c1[o][j]);
                }
            }
            CPT_FENCE;
        }
        __syncthreads();                                     // every read of T1 is done
        float* dst = L1 + ((7 * tr) * P1 + 7 * tc) * PIXF;
#pragma unroll
        for (int o = 0; o < 7; ++o)
#pragma unroll
            for (int cI = 0; cI < 7; ++cI) dst[(o * P1 + cI) * PIXF] = (cI & 1) ? c1[o][cI >> 1].y : c1[o][cI >> 1].x;
    }
    Taps tf;
    load_taps(tf, wpack, bpack, 1 + NL, C, cc, has_bias);
    __syncthreads();

    // ================= pass 2: y tile = conv(x + resize(C1)), input rows -2 .. 15, five accumulator rows in flight =================
    {
        constexpr int AHEAD = 2, R0 = -2, NR = 18;
        // C1 columns -2 .. 8 of the tile: the two on each side may lie outside the plane (clamped: ATen's border rule)
        const int cb0 = 7 * tc;
        const int cL0 = (ledge ? 0 : cb0 - 2) * PIXF, cL1 = (ledge ? 0 : cb0 - 1) * PIXF;
        const int cR0 = (redge ? P1 - 1 : cb0 + 7) * PIXF, cR1 = (redge ? P1 - 1 : cb0 + 8) * PIXF;
        // horizontal weights; the pairs that lie outside the image (columns -2, -1 at the left edge, 14, 15 at the right) are zeroed here
        const f32x2 wq = MODE == 1 ? splat(0.f) : splat(0.25f), wt = MODE == 1 ? splat(1.f) : splat(0.75f);
        uint32_t raw[NR][18];
        f32x2 H[2][9];
        f32x2 acc[5][7];
        const gcptr yimg = (gcptr)(reinterpret_cast<char*>(y) + (size_t)n * P0 * P0 * pix);
        const unsigned yoff = (unsigned)((14 * tc) * pix + c * ESZ);
        // H[i]: C1 row i (tile-local, -2 .. 8; clamped into the plane) resized horizontally to the 18 columns -2 .. 15
        auto build_H = [&](f32x2 (&Hs)[9], int i) {
            int ar = 7 * tr + i;
            ar = ar < 0 ? 0 : (ar > P1 - 1 ? P1 - 1 : ar);
            const float* rp = L1 + ar * (P1 * PIXF);
            float cv[11];
            cv[0] = rp[cL0];
            cv[1] = rp[cL1];
#pragma unroll
            for (int k = 0; k < 7; ++k) cv[2 + k] = rp[(cb0 + k) * PIXF];
            cv[9] = rp[cR0];
            cv[10] = rp[cR1];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                // columns 2j-2 (even) and 2j-1 (odd): 0.25 c[j] + 0.75 c[j+1] and 0.75 c[j+1] + 0.25 c[j+2]; nearest: c[j+1] twice
                const f32x2 e = f32x2{cv[j], cv[j + 2]};
                Hs[j] = pfma(splat(cv[j + 1]), wt, e * wq);
            }
            Hs[0] = Hs[0] * splat(lmask);
            Hs[8] = Hs[8] * splat(rmask);
        };
        sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], R0 + decltype(rc)::value); });
        build_H(H[0], -2);
        build_H(H[1], -1);
        sfor<NR>([&](auto rc) {
            constexpr int ri = decltype(rc)::value, t = R0 + ri;
            if constexpr (ri + AHEAD < NR) load_row(raw[ri + AHEAD], t + AHEAD);
            // accumulator row entering the window: output row t + 2
            if constexpr (t + 2 >= 0 && t + 2 <= 13) {
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[(t + 2) % 5][j] = splat(tf.bias);
            }
            // vertical source rows (tile origin is even): t even -> (t/2 - 1, t/2) weight 0.75; t odd -> ((t-1)/2, (t+1)/2) weight 0.25
            constexpr int te = (t + 2) & 1;                  // parity of t (t + 2 >= 0)
            constexpr int i0 = MODE == 1 ? ((t + 2) >> 1) - 1 : (te ? (t - 1) / 2 : t / 2 - 1);
            constexpr int i1 = MODE == 1 ? i0 : i0 + 1;
            constexpr float lam = MODE == 1 ? 0.f : (te ? 0.25f : 0.75f);
            // H[i1] is first needed here when t is odd (H[-2], H[-1] were built up front)
            if constexpr (MODE == 0 && te && t >= -1) build_H(H[(i1 + 2) & 1], i1);
            if constexpr (MODE == 1 && !te && t >= 0) build_H(H[(i0 + 2) & 1], i0);
            pin_row<18 * (NR - 1 - ri < AHEAD ? NR - 1 - ri : AHEAD)>(raw[ri]);
            if (row_valid(t)) {
                f32x2 row[9], odd[8];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const f32x2 xv = f32x2{__uint_as_float(raw[ri][2 * k]), __uint_as_float(raw[ri][2 * k + 1])};
                    if (MODE == 1) row[k] = xv + H[(i0 + 2) & 1][k];
                    else row[k] = pfma(splat(lam), H[(i1 + 2) & 1][k], pfma(splat(1.f - lam), H[(i0 + 2) & 1][k], xv));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) odd[j] = shift1(row[j], row[j + 1]);
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int o = t - u + 2;
                    if (o < 0 || o > 13) continue;
                    f32x2(&a)[7] = acc[o % 5];
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), a[j]);
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
                }
            }
            // output row t - 2 has seen its last input row
            if constexpr (t - 2 >= 0 && t - 2 <= 13) {
                constexpr int o = t - 2;
                typename PixSt<TIO>::packed pk[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) pk[j] = PixSt<TIO>::prep(acc[o % 5][j]);
                const gcptr rowb = opaque(yimg + (size_t)((14 * tr + o) * P0) * pix);
                if (cvalid) {
#pragma unroll
                    for (int k = 0; k < 14; ++k) PixSt<TIO>::st(opaque(rowb + (size_t)(k * pix)) + yoff, pk[k >> 1], k & 1);
                }
            }
#pragma unroll
            for (int o = 0; o < 14; ++o) if (o > t - 2 && o <= t + 2) pin(acc[o % 5]);
            pin(H[0]);
            pin(H[1]);
            CPT_FENCE;
        });
    }
}

static inline bool enabled()
{
    const char* v = getenv("RCX_CPT");
    const char* l = getenv("RCX_LANES");
    return !(v && *v == '0') && !(l && *l == '0');
}

template <int T, int HALVES, int MODE, typename TIO>
static hipError_t launch(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s)
{
    using G = Geo<T, HALVES, MODE, TIO>;
    auto kfn = k_recconv_cpt<T, HALVES, MODE, TIO>;
    static bool attr_set = false;                              // once per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const unsigned grid = (unsigned)(N * ((C + G::PIXF - 1) / G::PIXF));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(G::NT), G::LDS_BYTES, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr);
    return hipGetLastError();
}

template <int T, int HALVES>
static hipError_t launch_md(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s)
{
    if (dtype == 1) return mode == 1 ? launch<T, HALVES, 1, bf16_t>(x, y, wpack, bpack, N, C, s) : launch<T, HALVES, 0, bf16_t>(x, y, wpack, bpack, N, C, s);
    return mode == 1 ? launch<T, HALVES, 1, float>(x, y, wpack, bpack, N, C, s) : launch<T, HALVES, 0, float>(x, y, wpack, bpack, N, C, s);
}

}  // namespace cpt

bool cpt_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    (void)N;
    if (!cpt::enabled() || k != 5 || C < 1 || !(dtype == 0 || dtype == 1)) return false;
    return (H == 56 && W == 56 && level == 4) || (H == 28 && W == 28 && level == 3);
}

int cpt_describe(int N, int C, int H, int mode, char* buf, int len)
{
    const int T = H / 14, halves = T == 4 ? 2 : 1, pixf = 64 / halves;
    return snprintf(buf, len, "cpt(k_recconv_cpt<%d, %d, %d>,cb=%d,nt=%d,blocks=%d,lds=%d)", T, halves, mode, pixf, T * T / halves * 64,
                    N * ((C + pixf - 1) / pixf), T == 4 ? cpt::Geo<4, 2, 0, float>::LDS_BYTES : cpt::Geo<2, 1, 0, float>::LDS_BYTES);
}

hipError_t cpt_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int H, int mode, int dtype, hipStream_t s)
{
    if (H == 56) return cpt::launch_md<4, 2>(x, y, wpack, bpack, N, C, mode, dtype, s);
    return cpt::launch_md<2, 1>(x, y, wpack, bpack, N, C, mode, dtype, s);
}

}  // namespace rcx
