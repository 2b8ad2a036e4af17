// EXPERIMENT (round 4; not part of the product library -- measured slower than rcx_cpt_kernel.h, profiles/r04_cph_half_tile.txt).
// Channel-per-lane, HALF-TILE RecConv2d for the two large blocks of RecNeXt at 224x224 (model/recnext.py:24-34): 56x56 / level 4
// (stage 0) and 28x28 / level 3 (stage 1): the four-waves-per-SIMD geometry VERDICT r3 item 1 asked for.  Parity-green against the C oracle
// (tools/debug_cpt.py, all seven stage probes, both modes, max |err| 3.8e-6), <= 127 registers and no scratch in all 24 instantiations --
// and 125.7 us against 111.7 at 256 x 64 x 56 x 56 bf16 (28x28 x 128: 53.6 against 50.6): both kernels are bound by the number of
// vector-memory wave-instructions (7.3 CU cycles each whatever the width up to 4 bytes per lane: tools/ubench/vmem_rate.hip), which a
// narrower tile raises (more halo) and more waves do not hide.  Kept for the measurement harness (cph_bench.hip: phase stamps, ablations).
//
// What changed and why.  k_recconv_cpt runs 8 waves per CU at 256 registers: its phases (x -> F1, the
// small planes, T1 / C1, the final conv) are a chain of LDS round trips, barriers and load waits with two waves per SIMD to hide them, and
// the vector pipe is 30 % busy.  Both limits bind at two: LDS (float32 level-1 plane: 32 channel-images per CU) and registers (a lane
// that owns a 14-wide tile needs five accumulator rows of 14).  Here a LANE owns one channel of HALF a 14x14 tile -- 14 rows x 7 columns
// in the final conv (35 accumulators, 11-column input rows), 4 or 3 of the tile's 7 rows in every phase that works on the 7x7 level-1
// tile -- so a channel-image has 2 T^2 lanes instead of T^2 and the same LDS planes carry SIXTEEN waves per CU at <= 128 registers:
//   T = 4 (56x56): one workgroup of 16 waves = one image x 32 channels; a wave = 32 channels x the half-tiles (tr, tc, h), (tr, tc + 2, h)
//          (same tile row, same half, same column parity: every row quantity, the half and both resize parities are wave-uniform);
//   T = 2 (28x28): a workgroup of 8 waves = one image x 64 channels (a wave = 64 channels x one half-tile, 128 contiguous bytes per
//          pixel), two workgroups per CU, in different phases.
// At four waves per SIMD plain v_fma_f32 issues at the packed rate (a SIMD issues one vector instruction per ~2.2 cycles when two or
// more waves are ready, a v_pk_fma_f32 costs 4.3-4.8: DESIGN 5.1), so the level-0 passes and the level-1 conv are written on scalar
// FMAs: no register pairs, no v_pk_mov_b32 for the odd-aligned pairs, an odd number of columns per lane costs nothing.
//
// Phases (LDS planes float32 [pixel][channel of the block], as rcx_cpt_kernel.h):
//   pass 1   F1 rows of the tile split 4 + 3 over the two halves: F1 = down(x), input-row stationary, 18-column rows        (:27-29)
//   chain    the small planes by pieces (rcx_cpt_kernel.h's piece functions), a row per tile, its two column segments per half (:27-33)
//   level 1  T1 = F1 + resize(C2), C1 = conv(T1): the tile's rows split 4 + 3                                               (:31-33)
//   pass 2   y = conv(x + resize(C1)): 14 rows x 7 columns per lane, 18 input rows of 11 columns                            (:34)
// Same arithmetic as the other schedules: float32 throughout, one rounding at the store.  x + resize(C1) is formed per input row from
// two C1 rows of seven pixels held in registers: vertical blend first (7 values), then the horizontal 2x step into the 11 columns; the
// two halves of a tile start at columns of different parity, which is a pair of wave-uniform weights on the even columns and a third
// term with weight 0 on the odd ones (0 * v is exact; a non-finite C1 pixel would reach one column further than in ATen).
#pragma once
#include "rcx_cpt_kernel.h"

namespace rcx {
namespace cph {

using cpt::gcptr;
using cpt::i32x4;
using cpt::plane_size;
using cpt::raw_f32;
using cpt::SavedPyr;
using cpt::Taps;
using lanes::f32x2;
using lanes::IC;
using lanes::sfor;

#ifndef RCX_CPH_TRAIN
#define RCX_CPH_TRAIN 0                    /* the training-forward instantiations (they save the float32 pyramid) */
#endif
#ifndef RCX_CPH_ABL
#define RCX_CPH_ABL 0                      /* tools/cph_bench.hip: timing-only ablations (results are wrong): 1 / 8 = no x loads in pass 2 / pass 1, 2 / 16 = no FMAs there, 4 = no y stores */
#endif
#ifndef RCX_CPH_AHEAD1
#define RCX_CPH_AHEAD1 2                   /* rows of x in flight in front of the row being used, pass 1 / pass 2 */
#endif
#ifndef RCX_CPH_AHEAD2
#define RCX_CPH_AHEAD2 1                   /* 2 spills twelve registers at 128 */
#endif

// diagnostic build only (-DRCX_STAMPS, tools/cph_bench.hip): lane 0 of every wave of the first workgroups writes the clock at phase boundaries
#ifdef RCX_STAMPS
static __device__ unsigned long long* g_cph_stamps = nullptr;
#define CPH_STAMP(id)                                                                                                    \
    do {                                                                                                                 \
        if ((threadIdx.x & 63) == 0 && g_cph_stamps && blockIdx.x < 512)                                                 \
            g_cph_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (id)] = __builtin_readcyclecounter();            \
    } while (0)
#define CPH_STAMP_RT(id)                                                                                                 \
    do {                                                                                                                 \
        if ((threadIdx.x & 63) == 0 && g_cph_stamps && blockIdx.x < 512)                                                 \
            g_cph_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (id)] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
#else
#define CPH_STAMP(id) do { } while (0)
#define CPH_STAMP_RT(id) do { } while (0)
#endif

// the 25 taps of one conv for this lane's channel, one register each
struct TapsS {
    float w[25];
    float bias;
};
__device__ __forceinline__ void load_taps_s(TapsS& t, __amdgpu_buffer_rsrc_t wsrc, __amdgpu_buffer_rsrc_t bsrc, int conv, int C, int c)
{
    asm volatile("" : "+s"(C));                               // the 25 scalar offsets are recomputed here, not hoisted out of the unit loop (SGPR spills)
    const int vow = c * 4, base = conv * 25 * C * 4;
#pragma unroll
    for (int i = 0; i < 25; ++i) t.w[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrc, vow, base + i * C * 4, 0));
    t.bias = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bsrc, vow, conv * C * 4, 0));    // a buffer of zero records when there is no bias
}

// ---- pass 2's x rows: 11 columns (2 left of the half-tile, its 7, 2 right), one asm statement (see rcx_cpt_kernel.h, row_load).
// vl: columns 0, 1 (or out of range: left of the image), vm: columns 2 .. 8, vr: columns 9, 10 (or out of range: right of the image)
#define CPH_OUT11(v) "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10])
#define CPH_ROW11_IMM(OP)                                                                                                            \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                  \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1)                                                                          \
    CPT_LI(OP, 2, "vm", "t", 0) CPT_LI(OP, 3, "vm", "t", 1) CPT_LI(OP, 4, "vm", "t", 2) CPT_LI(OP, 5, "vm", "t", 3)                  \
    CPT_LI(OP, 6, "vm", "t", 4) CPT_LI(OP, 7, "vm", "t", 5) CPT_LI(OP, 8, "vm", "t", 6)                                              \
    CPT_LI(OP, 9, "vr", "t", 0) CPT_LI(OP, 10, "vr", "t", 1)
#define CPH_ROW11_GEN(OP)                                                                                                            \
    "s_add_i32 %[t], %[rb], 0\n\ts_add_i32 %[t2], %[rb], %[pix]\n\t"                                                                \
    CPT_LG(OP, 0, "vl", "t") CPT_LG(OP, 1, "vl", "t2") CPT_LG(OP, 9, "vr", "t") CPT_LG(OP, 10, "vr", "t2")                           \
    CPT_LG(OP, 2, "vm", "t") CPT_LG(OP, 3, "vm", "t2")                                                                               \
    CPT_LGN(OP, 4) CPT_LGN(OP, 5) CPT_LGN(OP, 6) CPT_LGN(OP, 7) CPT_LGN(OP, 8)

template <typename TIO, int PIXB>
__device__ __forceinline__ void row_load11(uint32_t (&v)[11], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 6 <= 4095) {
        (void)pix; (void)t2;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPH_ROW11_IMM(CPT_LDH) : CPH_OUT11(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPH_ROW11_IMM(CPT_LD16) : CPH_OUT11(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(CPH_ROW11_IMM(CPT_LD32) : CPH_OUT11(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else {
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPH_ROW11_GEN(CPT_LDH) : CPH_OUT11(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPH_ROW11_GEN(CPT_LD16) : CPH_OUT11(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else
            asm volatile(CPH_ROW11_GEN(CPT_LD32) : CPH_OUT11(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    }
}

template <int PENDING>
__device__ __forceinline__ void pin_row11(uint32_t (&v)[11])
{
    asm volatile("s_waitcnt vmcnt(%11)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]) : "n"(PENDING));
}

// ---- one output row of the half-tile: 7 stores in one statement.  vo = this lane's offset, or out of range (dropped: the lanes of a
// ragged last channel block).  16-bit: four registers of two converted pixels each (the last one holds one), low half = even column.
template <typename TIO, int PIXB> struct RowSt7;
template <typename T16, int PIXB> struct RowSt7_16 {
    static __device__ __forceinline__ void st(const float (&a)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        uint32_t p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                         // one conversion for two pixels (RNE, NaN stays NaN)
            const float lo = a[2 * j], hi = j < 3 ? a[2 * j + 1] : a[6];
            if constexpr (std::is_same<T16, f16_t>::value) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(lo), "v"(hi));
            else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(lo), "v"(hi));
        }
        int t, t2;
        if constexpr (PIXB > 0 && PIXB * 6 <= 4095) {
            (void)pix; (void)t2;
            asm volatile("s_add_i32 %[t], %[rb], 0\n\t"
                         CPT_SI("buffer_store_short", 0, "t", 0) CPT_SI("buffer_store_short_d16_hi", 0, "t", 1)
                         CPT_SI("buffer_store_short", 1, "t", 2) CPT_SI("buffer_store_short_d16_hi", 1, "t", 3)
                         CPT_SI("buffer_store_short", 2, "t", 4) CPT_SI("buffer_store_short_d16_hi", 2, "t", 5)
                         CPT_SI("buffer_store_short", 3, "t", 6)
                         : [t] "=&s"(t)
                         : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc", "memory");
        } else {
            asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                         CPT_SG("buffer_store_short", 0, "t2") CPT_SGN("buffer_store_short_d16_hi", 0)
                         CPT_SGN("buffer_store_short", 1) CPT_SGN("buffer_store_short_d16_hi", 1)
                         CPT_SGN("buffer_store_short", 2) CPT_SGN("buffer_store_short_d16_hi", 2)
                         CPT_SGN("buffer_store_short", 3)
                         : [t2] "=&s"(t2)
                         : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
            (void)t;
        }
    }
};
template <int PIXB> struct RowSt7<bf16_t, PIXB> : RowSt7_16<bf16_t, PIXB> {};
template <int PIXB> struct RowSt7<f16_t, PIXB> : RowSt7_16<f16_t, PIXB> {};
template <int PIXB> struct RowSt7<float, PIXB> {
    static __device__ __forceinline__ void st(const float (&a)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        int t2;
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_dword", 0, "t2") CPT_SGN("buffer_store_dword", 1) CPT_SGN("buffer_store_dword", 2) CPT_SGN("buffer_store_dword", 3)
                     CPT_SGN("buffer_store_dword", 4) CPT_SGN("buffer_store_dword", 5) CPT_SGN("buffer_store_dword", 6)
                     : [t2] "=&s"(t2)
                     : [p0] "v"(a[0]), [p1] "v"(a[1]), [p2] "v"(a[2]), [p3] "v"(a[3]), [p4] "v"(a[4]), [p5] "v"(a[5]), [p6] "v"(a[6]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    }
};

__device__ __forceinline__ void pinf(float& v) { asm volatile("" : "+v"(v)); }
template <int A> __device__ __forceinline__ void pinf(float (&v)[A]) {
#pragma unroll
    for (int i = 0; i < A; ++i) pinf(v[i]);
}

template <int T_>
struct Geo {
    static constexpr int T = T_;
    static_assert(T == 4 || T == 2, "56x56 or 28x28");
    static constexpr int NL = T == 4 ? 4 : 3;              // levels of the block (the full ladder down to 4 x 4)
    static constexpr int HW = T == 4 ? 2 : 1;              // half-tiles per wave
    static constexpr int CB = 64 / HW;                     // channels of a workgroup's block
    static constexpr int PIXF = CB;                        // floats between two pixels of an LDS plane
    static constexpr int NTILE = T * T;
    static constexpr int NW = 2 * NTILE / HW;              // 16 / 8 waves
    static constexpr int NT = NW * 64;
    static constexpr int P0 = 14 * T, P1 = 7 * T, P2 = plane_size(T, 2), P3 = plane_size(T, 3), P4 = plane_size(T, 4);
    // LDS, in pixels: zero row | guard | L1 | guard | L2 | L3 | L4
    static constexpr int ZR = P1;
    static constexpr int O1 = ZR + 2;
    static constexpr int O2 = O1 + P1 * P1 + 2;
    static constexpr int O3 = O2 + P2 * P2;
    static constexpr int O4 = O3 + P3 * P3;
    static constexpr int NPIX = O4 + (NL >= 4 ? P4 * P4 : 0);
    static constexpr int LDS_BYTES = NPIX * PIXF * 4;
    static constexpr int PER_CU = T == 4 ? 1 : 2;          // workgroups resident per CU: 16 waves either way
    static_assert(PER_CU * LDS_BYTES <= 160 * 1024, "LDS");
};

template <int T, int MODE, int PIXB, typename TIO, bool TRAIN = false>
__global__ __launch_bounds__(Geo<T>::NT) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_recconv_cph(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                   int N, int C, int has_bias, SavedPyr sv)
{
    using G = Geo<T>;
    constexpr int NL = G::NL, PIXF = G::PIXF, NTILE = G::NTILE, P0 = G::P0, P1 = G::P1, P2 = G::P2, P3 = G::P3, P4 = G::P4;
    constexpr int ESZ = (int)sizeof(TIO);
    constexpr int CHB = G::CB;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // persistent workgroups, XCD-aware unit order (rcx_cpt_kernel.h)
    const int nb = (C + CHB - 1) / CHB;
    const unsigned total = (unsigned)N * (unsigned)nb, GD = gridDim.x;
    const bool xcd = (total & 7u) == 0 && (GD & 7u) == 0;
    const int tid = (int)threadIdx.x;
    const int pix = C * ESZ;                                      // bytes between horizontally adjacent pixels
    const unsigned OOB = 0x80000000u;

    // ---- zero the whole LDS image once (zero row, guards; and every later read is of finite data)
    for (int i = tid; i < G::LDS_BYTES / 16; i += G::NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // wave -> (tile row, half, tile column(s)).  The bits that pick the SIMD (w & 3) carry neither the half nor the high bit of the tile
    // row, so every SIMD holds two waves of either half (the halves' shares of the 7-row phases are 4 and 3 rows).
    // What derives from the wave and lane indices is recomputed in every phase from opaque copies of them, not hoisted out of the unit loop
    // and kept live (or spilled) across the phases: the kernel runs at 128 registers.  tc: tile column (per lane at T = 4), ch: channel of the block
    const int w_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto wave_now = [&]() { int v = w_; asm volatile("" : "+s"(v)); return v; };
    auto tr_of = [&](int w) { return T == 4 ? (w >> 3) * 2 + ((w >> 1) & 1) : ((w >> 1) & 1); };
    auto h_of = [&](int w) { return T == 4 ? ((w >> 2) & 1) : (w >> 2); };
    auto lane_now = [&]() { int l = tid & 63; asm volatile("" : "+v"(l)); return l; };
    auto tc_of = [&](int w, int lane) { return T == 4 ? (w & 1) + 2 * (lane >> 5) : (w & 1); };
    auto ch_of = [&](int lane) { return T == 4 ? (lane & 31) : lane; };
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, (NL + 2) * 25 * C * 4, 0x00020000);
    // no bias: a buffer of zero records, every load returns 0 (no per-lane flag, no branch)
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bpack, 0, has_bias ? (NL + 2) * C * 4 : 0, 0x00020000);

  for (unsigned it = 0;; ++it) {
    unsigned unit;
    if (xcd) {
        const unsigned k = (blockIdx.x >> 3) + it * (GD >> 3);
        if (k >= (total >> 3)) break;
        unit = (blockIdx.x & 7u) * (total >> 3) + k;
    } else {
        unit = blockIdx.x + it * GD;
        if (unit >= total) break;
    }
    const int n = (int)(unit / (unsigned)nb), cb = (int)(unit - (unsigned)n * (unsigned)nb);
    auto sv_ptr = [&](unsigned long long off, int P, int row, int col, int c) -> float* {
        return reinterpret_cast<float*>(reinterpret_cast<char*>(sv.base) + off) + (((size_t)n * P + row) * P + col) * C + c;
    };

    // x image as a raw buffer: base, num_records = bytes of the image (offsets past it read 0)
    i32x4 rsrc;
    {
        const unsigned long long a = (unsigned long long)(reinterpret_cast<const char*>(x) + (size_t)n * P0 * P0 * pix);
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
        rsrc.z = P0 * P0 * pix;
        rsrc.w = 0x00020000;
    }
    auto row_base = [&](int tr, int r) -> int {                  // rows outside the image are redirected to a valid row (loaded, not used)
        int ar = 14 * tr + r;
        ar = ar < 0 ? 0 : (ar > P0 - 1 ? P0 - 1 : ar);
        return __builtin_amdgcn_readfirstlane(ar * (P0 * pix));
    };
    CPH_STAMP(0);
    CPH_STAMP_RT(9);
    __syncthreads();                                              // LDS zeroed (first unit); the previous unit's pass 2 has read its C1
    CPH_STAMP(1);

    // ================= pass 1: F1 = down(x), input-row stationary over the tile's rows -2 .. 14; the halves share the tile's 7 F1 columns
    // 4 + 3: half 0 reads x columns -2 .. 8 of the tile, half 1 columns 5 .. 15 (the windows pass 2 reads: same lane offsets) =================
    {
        const int w = wave_now(), tr = tr_of(w), h = h_of(w), lane = lane_now(), tc = tc_of(w, lane), ch = ch_of(lane);
        const int c = cb * CHB + ch, cc = c < C ? c : C - 1;
        const bool svon = TRAIN && sv.base != nullptr && c < C;
        TapsS td;
        load_taps_s(td, wsrc, bsrc, 0, C, cc);
        const unsigned voffM = (unsigned)((14 * tc + 7 * h) * pix + cc * ESZ);                 // column 0 of the half-tile
        const unsigned voffL = (tc == 0 && h == 0) ? OOB : voffM - 2u * (unsigned)pix;          // its columns -2, -1: left of the image?
        const unsigned voffR = (tc == T - 1 && h == 1) ? OOB : voffM + 7u * (unsigned)pix;      // its columns 7, 8: right of the image?
        float* const f1dst = lds + ch + (G::O1 + (7 * tr) * P1 + 7 * tc) * PIXF;
        auto pass1 = [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            constexpr int NC = HH ? 3 : 4, K0 = HH ? 1 : 0, CO0 = HH ? 4 : 0;       // F1 columns CO0 .. CO0+NC-1; column i, tap v reads window column K0 + 2 i + v
            constexpr int AHEAD = RCX_CPH_AHEAD1, R0 = -2, NR = 17;
            uint32_t raw[NR][11];
            float acc[3][NC];
            if constexpr (!(RCX_CPH_ABL & 8)) sfor<AHEAD>([&](auto rc) { row_load11<TIO, PIXB>(raw[decltype(rc)::value], voffL, voffM, voffR, rsrc, row_base(tr, R0 + decltype(rc)::value), pix); });
            sfor<NR>([&](auto rc) {
                constexpr int ri = decltype(rc)::value, r = R0 + ri;
                if constexpr (RCX_CPH_ABL & 8) {
#pragma unroll
                    for (int k = 0; k < 11; ++k) raw[ri][k] = (uint32_t)(lane + k + ri) << 16;
                } else {
                if constexpr (ri + AHEAD < NR) row_load11<TIO, PIXB>(raw[ri + AHEAD], voffL, voffM, voffR, rsrc, row_base(tr, r + AHEAD), pix);
                constexpr int NY = (NR - 1 - ri < AHEAD ? NR - 1 - ri : AHEAD) * 11;
                pin_row11<(NY > 63 ? 63 : NY)>(raw[ri]);
                }
                float xr[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) xr[k] = raw_f32<TIO>(raw[ri][k]);
                const bool rv = 14 * tr + r >= 0 && 14 * tr + r < P0;        // uniform
#pragma unroll
                for (int o = 0; o < 7; ++o) {
                    const int u = r - 2 * o + 2;
                    if (u < 0 || u > 4) continue;
                    float(&a)[NC] = acc[o % 3];
                    if (rv) {
#pragma unroll
                        for (int v = 0; v < ((RCX_CPH_ABL & 16) ? 1 : 5); ++v)
#pragma unroll
                            for (int i = 0; i < NC; ++i) a[i] = fmaf(xr[K0 + 2 * i + v], td.w[u * 5 + v], (u == 0 && v == 0) ? td.bias : a[i]);
                    } else if (u == 0) {
#pragma unroll
                        for (int i = 0; i < NC; ++i) a[i] = td.bias;
                    }
                    if (u == 4) {
#pragma unroll
                        for (int i = 0; i < NC; ++i) f1dst[(o * P1 + CO0 + i) * PIXF] = a[i];
                        if constexpr (TRAIN) if (svon) {
#pragma unroll
                            for (int i = 0; i < NC; ++i) *sv_ptr(sv.f_off[1], P1, 7 * tr + o, 7 * tc + CO0 + i, c) = a[i];
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < 7; ++o) if (r - 2 * o + 2 >= 0 && r - 2 * o + 2 < 4) pinf(acc[o % 3]);
                CPT_FENCE;
            });
        };
        if (h == 0) pass1(IC<0>{});
        else pass1(IC<1>{});
    }
    CPH_STAMP(2);
    __syncthreads();
    CPH_STAMP(3);

    // ================= chain: the small planes by pieces.  Worker = (tile tl, half h): row tl (+ NTILE per round), column segment h =================
    // conv j of the pack: 0 = down, 1 + (NL - l) = the conv of level l, 1 + NL = the final conv
    {
        const int w = wave_now(), tr = tr_of(w), h = h_of(w), lane = lane_now(), tc = tc_of(w, lane), ch = ch_of(lane);
        const int c = cb * CHB + ch, cc = c < C ? c : C - 1;
        const bool svon = TRAIN && sv.base != nullptr && c < C;
        const int tl = tr * T + tc;
        float* const L = lds + ch;
        const float* const Lzero = L;
        constexpr int PL[5] = {P0, P1, P2, P3, P4};
        float* const LP[5] = {nullptr, L + G::O1 * PIXF, L + G::O2 * PIXF, L + G::O3 * PIXF, L + G::O4 * PIXF};
        auto for_pieces = [&](auto pc, auto&& f) {
            constexpr int P = decltype(pc)::value;
            constexpr int NA = (P + 1) / 2, NB = P - NA;
            constexpr int RNDS = (P + NTILE - 1) / NTILE;
            sfor<RNDS>([&](auto rc) {
                constexpr int rnd = decltype(rc)::value;
                const int rr = tl + NTILE * rnd;
                const bool act = rr < P;
                if (h == 0) f(rc, IC<0>{}, IC<NA>{}, act ? rr : 0, act);
                else f(rc, IC<NA>{}, IC<NB>{}, act ? rr : 0, act);
            });
        };
        {
            Taps td;
            cpt::load_taps(td, wsrc, bsrc, 0, C, cc);
            // down ladder: F_l = down(F_{l-1}), l = 2 .. NL
            sfor<NL - 1>([&](auto lc) {
                constexpr int l = 2 + decltype(lc)::value;
                constexpr int PIN = PL[l - 1], PO = PL[l];
                for_pieces(IC<PO>{}, [&](auto, auto col0c, auto noutc, int row, bool act) {
                    constexpr int COL0 = decltype(col0c)::value, NOUT = decltype(noutc)::value;
                    float out[NOUT];
                    cpt::down_piece<PIN, COL0, NOUT, PIXF>(LP[l - 1], Lzero, row, td, out);
                    if (act) {
                        float* dst = LP[l] + (row * PO + COL0) * PIXF;
#pragma unroll
                        for (int i = 0; i < NOUT; ++i) dst[i * PIXF] = out[i];
                        if constexpr (TRAIN) if (svon) {
#pragma unroll
                            for (int i = 0; i < NOUT; ++i) *sv_ptr(sv.f_off[l], PO, row, COL0 + i, c) = out[i];
                        }
                    }
                });
                __syncthreads();
            });
        }
        CPH_STAMP(4);
        // up recursion on the piece planes: l = NL .. 2: T_l = F_l + resize(C_{l+1}) in place (l < NL), C_l = conv(T_l) in place
        sfor<NL - 1>([&](auto lc) {
            constexpr int l = NL - decltype(lc)::value;
            constexpr int P = PL[l];
            Taps tc_;
            cpt::load_taps(tc_, wsrc, bsrc, 1 + (NL - l), C, cc);
            if constexpr (l < NL) {
                constexpr int PC = PL[l + 1];
                for_pieces(IC<P>{}, [&](auto, auto col0c, auto noutc, int row, bool act) {
                    cpt::tform_piece<MODE, PC, P, decltype(col0c)::value, decltype(noutc)::value, PIXF>(LP[l], LP[l + 1], row, act);
                });
                __syncthreads();
            }
            constexpr int RN = (P + NTILE - 1) / NTILE;
            f32x2 res[RN][4];
            for_pieces(IC<P>{}, [&](auto rc, auto col0c, auto noutc, int row, bool) {
                constexpr int COL0 = decltype(col0c)::value, NOUT = decltype(noutc)::value;
                f32x2 acc[(NOUT + 1) / 2];
                cpt::conv_piece<P, COL0, NOUT, PIXF>(LP[l], Lzero, row, tc_, acc);
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
                    if constexpr (TRAIN) if (svon) {
#pragma unroll
                        for (int i = 0; i < NOUT; ++i)
                            *sv_ptr(sv.c_off[l], P, row, COL0 + i, c) = (i & 1) ? res[decltype(rc)::value][i >> 1].y : res[decltype(rc)::value][i >> 1].x;
                    }
                }
            });
            __syncthreads();
        });
    }

    CPH_STAMP(5);
    // ================= level 1, per tile, rows O0 .. O0+NO-1 by half: T1 = F1 + resize(C2) (exact 2x), C1 = conv(T1) =================
    {
        const int w = wave_now(), tr = tr_of(w), h = h_of(w), lane = lane_now(), tc = tc_of(w, lane), ch = ch_of(lane);
        const int c = cb * CHB + ch, cc = c < C ? c : C - 1;
        const bool svon = TRAIN && sv.base != nullptr && c < C;
        const bool ledge = tc == 0, redge = tc == T - 1;              // per lane at T = 4
        float* const L2 = lds + ch + G::O2 * PIXF;
        float* const tile = lds + ch + (G::O1 + (7 * tr) * P1 + 7 * tc) * PIXF;
        // columns: run of 7 starting at absolute column 7*tc (its parity is wave-uniform), source columns b .. b+4 of C2, clamped
        const int d0 = 7 * tc;
        {
            const int bcol = MODE == 1 ? (d0 >> 1) : ((d0 - 1) >> 1);
            int cofs[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                int cx = bcol + k;
                cx = cx < 0 ? 0 : (cx > P2 - 1 ? P2 - 1 : cx);
                cofs[k] = cx * PIXF;
            }
            auto form = [&](auto parc, auto o0c, auto noc) {
                constexpr int PAR = decltype(parc)::value, O0 = decltype(o0c)::value, NO = decltype(noc)::value;
#pragma unroll
                for (int r = O0; r < O0 + NO; ++r) {
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
                    float* rowp = tile + r * (P1 * PIXF);
                    sfor<7>([&](auto cic) {
                        constexpr int cI = decltype(cic)::value;
                        constexpr cpt::Rel rl = cpt::rel2(MODE, PAR, cI);
                        const float up = MODE == 1 ? V[rl.idx] : fmaf(rl.l, V[rl.idx + 1], (1.f - rl.l) * V[rl.idx]);
                        rowp[cI * PIXF] += up;
                    });
                }
            };
            const int cpar = __builtin_amdgcn_readfirstlane(d0 & 1);
            if (h == 0) { if (cpar) form(IC<1>{}, IC<0>{}, IC<4>{}); else form(IC<0>{}, IC<0>{}, IC<4>{}); }
            else { if (cpar) form(IC<1>{}, IC<4>{}, IC<3>{}); else form(IC<0>{}, IC<4>{}, IC<3>{}); }
        }
        __syncthreads();
        CPH_STAMP(6);

        TapsS t1;
        load_taps_s(t1, wsrc, bsrc, NL, C, cc);                   // conv of level 1 = pack 1 + (NL - 1)
        // C1 rows O0 .. O0+NO-1 of the tile, input-row stationary over T1 rows O0-2 .. O0+NO+1, columns -2 .. 8 (the guards before and
        // after the plane make every address valid; a column outside the plane is selected away)
        float c1[4][7];
        auto c1conv = [&](auto o0c, auto noc) {
            constexpr int O0 = decltype(o0c)::value, NO = decltype(noc)::value;
#pragma unroll
            for (int t = O0 - 2; t <= O0 + NO + 1; ++t) {
                const int ar = 7 * tr + t;
                if (ar >= 0 && ar < P1) {                        // uniform
                    const float* rp = tile + t * (P1 * PIXF);
                    float in[11];
#pragma unroll
                    for (int k = 0; k < 11; ++k) in[k] = rp[(k - 2) * PIXF];
                    in[0] = ledge ? 0.f : in[0];
                    in[1] = ledge ? 0.f : in[1];
                    in[9] = redge ? 0.f : in[9];
                    in[10] = redge ? 0.f : in[10];
#pragma unroll
                    for (int o = O0; o < O0 + NO; ++o) {
                        const int u = t - o + 2;
                        if (u < 0 || u > 4) continue;
#pragma unroll
                        for (int v = 0; v < 5; ++v)
#pragma unroll
                            for (int j = 0; j < 7; ++j) c1[o - O0][j] = fmaf(in[j + v], t1.w[u * 5 + v], (u == 0 && v == 0) ? t1.bias : c1[o - O0][j]);
                    }
                } else if (t + 2 >= O0 && t + 2 < O0 + NO) {   // a row above the plane: the output row it would have opened starts from the bias
#pragma unroll
                    for (int j = 0; j < 7; ++j) c1[t + 2 - O0][j] = t1.bias;
                }
                CPT_FENCE;
            }
        };
        auto c1store = [&](auto o0c, auto noc) {
            constexpr int O0 = decltype(o0c)::value, NO = decltype(noc)::value;
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
                for (int j = 0; j < 7; ++j) tile[((O0 + o) * P1 + j) * PIXF] = c1[o][j];
            if constexpr (TRAIN) if (svon) {
#pragma unroll
                for (int o = 0; o < NO; ++o)
#pragma unroll
                    for (int j = 0; j < 7; ++j) *sv_ptr(sv.c_off[1], P1, 7 * tr + O0 + o, 7 * tc + j, c) = c1[o][j];
            }
        };
        if (h == 0) c1conv(IC<0>{}, IC<4>{});
        else c1conv(IC<4>{}, IC<3>{});
        __syncthreads();                                         // every read of T1 is done
        if (h == 0) c1store(IC<0>{}, IC<4>{});
        else c1store(IC<4>{}, IC<3>{});
    }
    __syncthreads();
    CPH_STAMP(7);

    // ================= pass 2: y half-tile = conv(x + resize(C1)), input rows -2 .. 15, 11 columns, five accumulator rows in flight =================
    {
        constexpr int AHEAD = RCX_CPH_AHEAD2, R0 = -2, NR = 18;
        const int w = wave_now(), tr = tr_of(w), h = h_of(w), lane = lane_now(), tc = tc_of(w, lane), ch = ch_of(lane);
        const int c = cb * CHB + ch, cc = c < C ? c : C - 1;
        TapsS tf;
        load_taps_s(tf, wsrc, bsrc, 1 + NL, C, cc);
        const bool lout = tc == 0 && h == 0, rout = tc == T - 1 && h == 1;                    // the two halo columns on that side are outside the image
        const unsigned voffM = (unsigned)((14 * tc + 7 * h) * pix + cc * ESZ);                 // column 0 of the half-tile
        const unsigned voffL = lout ? OOB : voffM - 2u * (unsigned)pix;
        const unsigned voffR = rout ? OOB : voffM + 7u * (unsigned)pix;
        i32x4 ysrc;                                           // y image as a raw buffer; lanes past the last channel store out of range (dropped)
        {
            const unsigned long long a = (unsigned long long)(reinterpret_cast<char*>(y) + (size_t)n * P0 * P0 * pix);
            ysrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            ysrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
            ysrc.z = P0 * P0 * pix;
            ysrc.w = 0x00020000;
        }
        const unsigned yoff = c < C ? voffM : OOB;
        // The 11 columns start at absolute column c0 = 14 tc + 7 h - 2: even for h = 0, odd for h = 1.  C1 columns vb .. vb + 6, vb = 7 tc - 2 (h = 0)
        // or 7 tc + 2 (h = 1), clamped into the plane (ATen's border rule); only the outer two on either side can leave it.
        const int vb = 7 * tc + (h ? 2 : -2);
        auto clampc = [&](int cx) { return (cx < 0 ? 0 : (cx > P1 - 1 ? P1 - 1 : cx)) * PIXF; };
        const int q0 = clampc(vb), q1 = clampc(vb + 1), q5 = clampc(vb + 5), q6 = clampc(vb + 6);
        const float* const L1c = lds + ch + G::O1 * PIXF;
        const float* const Lm = L1c + (vb + 2) * PIXF;            // columns vb + 2 .. vb + 4: always inside
        // column k (absolute c0 + k): bilinear, k even: ae V[k/2] + be V[k/2 + 1]; k odd, m = (k - 1) / 2: oa V[m] + ob V[m + 1] + oc V[m + 2];
        // nearest: k even: V[k/2 + 1] (h = 0) or V[k/2] (h = 1) as the same two-term form with weights 0 / 1; k odd: V[(k + 1) / 2]
        const float ae = MODE == 1 ? (h ? 1.f : 0.f) : (h ? 0.75f : 0.25f), be = MODE == 1 ? (h ? 0.f : 1.f) : (h ? 0.25f : 0.75f);
        const float oa = h ? 0.25f : 0.f, ob = 0.75f, oc = h ? 0.f : 0.25f;
        float Rr[2][7];                                           // C1 rows i (tile-local, -2 .. 8; clamped into the plane), slot (i + 2) & 1
        auto load_R = [&](float (&R)[7], int i) {
            int ar = 7 * tr + i;
            ar = ar < 0 ? 0 : (ar > P1 - 1 ? P1 - 1 : ar);
            const float* rp = L1c + ar * (P1 * PIXF);
            R[0] = rp[q0];
            R[1] = rp[q1];
            const float* rm = Lm + ar * (P1 * PIXF);
#pragma unroll
            for (int k = 0; k < 3; ++k) R[2 + k] = rm[k * PIXF];
            R[5] = rp[q5];
            R[6] = rp[q6];
        };
        uint32_t raw[NR][11];
        float acc[5][7];
        if constexpr (!(RCX_CPH_ABL & 1)) sfor<AHEAD>([&](auto rc) { row_load11<TIO, PIXB>(raw[decltype(rc)::value], voffL, voffM, voffR, rsrc, row_base(tr, R0 + decltype(rc)::value), pix); });
        if constexpr (MODE == 1) load_R(Rr[1], -1);
        else { load_R(Rr[0], -2); load_R(Rr[1], -1); }
        sfor<NR>([&](auto rc) {
            constexpr int ri = decltype(rc)::value, t = R0 + ri;
            if constexpr (RCX_CPH_ABL & 1) {
#pragma unroll
                for (int k = 0; k < 11; ++k) raw[ri][k] = (uint32_t)(lane + k + ri) << 16;
            } else if constexpr (ri + AHEAD < NR) row_load11<TIO, PIXB>(raw[ri + AHEAD], voffL, voffM, voffR, rsrc, row_base(tr, t + AHEAD), pix);
            // vertical source rows (tile origin is even): t even -> (t/2 - 1, t/2) weight 0.75; t odd -> ((t-1)/2, (t+1)/2) weight 0.25
            constexpr int te = (t + 2) & 1;                  // parity of t (t + 2 >= 0)
            constexpr int i0 = MODE == 1 ? ((t + 2) >> 1) - 1 : (te ? (t - 1) / 2 : t / 2 - 1);
            constexpr int i1 = MODE == 1 ? i0 : i0 + 1;
            constexpr float lam = MODE == 1 ? 0.f : (te ? 0.25f : 0.75f);
            if constexpr (MODE == 0 && te && t >= -1) load_R(Rr[(i1 + 2) & 1], i1);
            if constexpr (MODE == 1 && !te && t >= 0) load_R(Rr[(i0 + 2) & 1], i0);
            // younger memory operations: the rows requested since (11 loads each) and the output rows stored at the end of the iterations in
            // between (7 stores each; iteration i stores a row for 4 <= i <= 17)
            constexpr int NLD = NR - 1 - ri < AHEAD ? NR - 1 - ri : AHEAD;
            constexpr int NST = [] { int k = 0; for (int j = 1; j <= AHEAD; ++j) k += (ri - j >= 4 && ri - j <= 17) ? 1 : 0; return k; }();
            if constexpr (!(RCX_CPH_ABL & 1)) pin_row11<(11 * NLD + 7 * NST > 63 ? 63 : 11 * NLD + 7 * NST)>(raw[ri]);
            if (14 * tr + t >= 0 && 14 * tr + t < P0) {               // uniform
                float V[7], row[11];
#pragma unroll
                for (int j = 0; j < 7; ++j) V[j] = MODE == 1 ? Rr[(i0 + 2) & 1][j] : fmaf(lam, Rr[(i1 + 2) & 1][j], (1.f - lam) * Rr[(i0 + 2) & 1][j]);
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float xv = raw_f32<TIO>(raw[ri][k]);
                    if (k & 1) {
                        const int m = (k - 1) / 2;
                        if (MODE == 1) row[k] = xv + V[m + 1];
                        else row[k] = fmaf(oc, V[m + 2], fmaf(ob, V[m + 1], fmaf(oa, V[m], xv)));
                    } else {
                        const int m = k / 2;
                        row[k] = fmaf(be, V[m + 1], fmaf(ae, V[m], xv));
                    }
                }
                row[0] = lout ? 0.f : row[0];
                row[1] = lout ? 0.f : row[1];
                row[9] = rout ? 0.f : row[9];
                row[10] = rout ? 0.f : row[10];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int o = t - u + 2;
                    if (o < 0 || o > 13) continue;
                    float(&a)[7] = acc[o % 5];
#pragma unroll
                    for (int v = 0; v < ((RCX_CPH_ABL & 2) ? 1 : 5); ++v)
#pragma unroll
                        for (int j = 0; j < 7; ++j) a[j] = fmaf(row[j + v], tf.w[u * 5 + v], (u == 0 && v == 0) ? tf.bias : a[j]);     // u == 0: output row t + 2 enters the window
                }
            } else if constexpr (t + 2 >= 0 && t + 2 <= 13) {  // a row outside the image: the output row it would have opened starts from the bias
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[(t + 2) % 5][j] = tf.bias;
            }
            // output row t - 2 has seen its last input row
            if constexpr (t - 2 >= 0 && t - 2 <= 13) {
                constexpr int o = t - 2;
                const int yrb = __builtin_amdgcn_readfirstlane((14 * tr + o) * (P0 * pix));
                if constexpr (RCX_CPH_ABL & 4) { if (acc[o % 5][0] == 1.2345f) RowSt7<TIO, PIXB>::st(acc[o % 5], yoff, ysrc, yrb, pix); }
                else RowSt7<TIO, PIXB>::st(acc[o % 5], yoff, ysrc, yrb, pix);
            }
#pragma unroll
            for (int o = 0; o < 14; ++o) if (o > t - 2 && o <= t + 2) pinf(acc[o % 5]);
            pinf(Rr[0]);
            pinf(Rr[1]);
            CPT_FENCE;
        });
    }
    CPH_STAMP(8);
    CPH_STAMP_RT(10);
    // the next unit's first barrier (top of the loop) orders this unit's C1 reads before the next pass 1's F1 writes
  }
}

template <int T, int MODE, int PIXB, typename TIO, bool TRAIN = false>
static hipError_t launch(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    using G = Geo<T>;
#if RCX_CPH_TRAIN
    if constexpr (!TRAIN && MODE == 0) {                           // training forward: bilinear only (what RecConv2d trains with)
        if (sv.base) return launch<T, MODE, PIXB, TIO, true>(x, y, wpack, bpack, N, C, s, sv);
    }
#endif
    if (!TRAIN && sv.base) return hipErrorInvalidConfiguration;
    auto kfn = k_recconv_cph<T, MODE, PIXB, TIO, TRAIN>;
    RCX_SET_LDS_ONCE(kfn, G::LDS_BYTES);
    static std::atomic<int> cus_cache{0};
    int cus = cus_cache.load(std::memory_order_relaxed);
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus = v;
        cus_cache.store(v, std::memory_order_relaxed);
    }
    const unsigned total = (unsigned)(N * ((C + G::CB - 1) / G::CB));
    unsigned cap = (unsigned)cus * (unsigned)G::PER_CU;
    if (const char* e = rcx::opt::value(rcx::opt::CPT_GRID)) { const int g = atoi(e); if (g > 0) cap = (unsigned)g; }    // A/B knob
    cap &= ~7u;
    const unsigned grid = total <= cap || cap == 0 ? total : cap;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(G::NT), G::LDS_BYTES, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
    return hipGetLastError();
}

// the channel counts of RecNeXt-M3 / M4 get the compile-time pixel pitch (immediate column offsets), the rest the run-time one
template <int T, int MODE, typename TIO>
static hipError_t launch_c(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    constexpr int CM3 = T == 4 ? 64 : 128;
    if (C == CM3) return launch<T, MODE, CM3 * (int)sizeof(TIO), TIO>(x, y, wpack, bpack, N, C, s, sv);
    return launch<T, MODE, 0, TIO>(x, y, wpack, bpack, N, C, s, sv);
}

template <int T>
static hipError_t launch_md(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s, const SavedPyr& sv)
{
    if (dtype == 1) return mode == 1 ? launch_c<T, 1, bf16_t>(x, y, wpack, bpack, N, C, s, sv) : launch_c<T, 0, bf16_t>(x, y, wpack, bpack, N, C, s, sv);
    if (dtype == 2) return mode == 1 ? launch_c<T, 1, f16_t>(x, y, wpack, bpack, N, C, s, sv) : launch_c<T, 0, f16_t>(x, y, wpack, bpack, N, C, s, sv);
    return mode == 1 ? launch_c<T, 1, float>(x, y, wpack, bpack, N, C, s, sv) : launch_c<T, 0, float>(x, y, wpack, bpack, N, C, s, sv);
}

}  // namespace cph
}  // namespace rcx
