// Backward of the FINE levels of the RecConv2d block on the 56x56 and 28x28 planes (the adjoint of model/recnext.py:31-34 and of the
// first steps of the down ladder, :27-29), channel per lane, tiled like the forward's single steps (rcx_upcpt.hip):
// a wave = 64 channels of one 14 x 14 tile of the plane, a lane = one channel, nothing shared between lanes, no LDS, no barrier.
//
// With g the gradient of a level's conv output (gy at level 0, gC_l below), K^ the conv with its taps flipped, D the shared stride-2
// conv, R the exact-2x resize (rcx_bwd.hip has the whole recursion):
//
//   k_bwd_gc   gC = R^T (K^ g)            the gradient handed DOWN to the next level, at half resolution: the full-resolution gT = K^ g
//                                         never exists in memory -- a tile forms it on its 14 x 14 pixels plus the one-pixel ring R^T
//                                         reads (16 x 16 outputs from a 20 x 20 window of g) and leaves its 7 x 7 coarse pixels;
//   k_bwd_gx   gF = K^ g + D^T G          the gradient handed UP: G = the total gradient of the next level's plane (known once the
//                                         coarser levels are done); at level 0 gF is gx, rounded once at the store.  K^ g is formed
//                                         a second time here instead of being kept: 25 packed FMAs per pixel pair against a float32
//                                         plane written and read back (at 56 x 56 x 64 channels x 128 images: 206 MB).
//
// The per-step schedule these replace ran K^ g (float32 in and out), R^T and D^T as three gathers of one thread per four channels
// of a pixel: 46 + 38 + 71 us at 128 x 64 x 56 x 56 where the two kernels here move 0.13 GB (profiles/r06_train_*).
//
// R^T of the exact 2x bilinear step: coarse pixel i collects fine pixels 2i-1 .. 2i+2 with weights 1/4, 3/4, 3/4, 1/4; at the border the
// forward clamps its source index, so the fine pixel that would lie outside gives its weight to the edge pixel (1/4 + 3/4 = 1):
// resolved per tile with wave-uniform weights, no branch.  Nearest: weights 0, 1, 1, 0.
// Arithmetic: float32 throughout, packed pairs of horizontally adjacent pixels (v_pk_fma_f32), input-row stationary with five
// accumulator rows in flight, exactly as the forward's pass 2.  Rows are hand-issued buffer loads through per-row descriptors (rcx_upcpt.hip): a 16-bit
// element lands in float32 position (D16-hi), everything outside the plane is out of range and reads 0 -- no conversion, mask or address instructions;
// every wave issues the same sequence, so the s_waitcnt counts are compile-time numbers (SchedGX / SchedGC / SchedWD).
#pragma once
#include "rcx_cpt_kernel.h"
#include "rcx_opts.h"
// Compiled as five translation units (rcx_cptbwd_gx.hip, _gc.hip, _wk.hip, _wd.hip, _dn.hip: RCX_CPTBWD_PART = 1 .. 5), each instantiating one kernel family.

#ifndef RCX_GX_AHEAD
#define RCX_GX_AHEAD 1              /* g rows in flight in front of the row being used (k_bwd_gx: two measured 31.1 vs 30.0 us at 128 x 64 x 56 x 56 -- these kernels move 3 - 4.3 TB/s, not latency-bound) */
#endif
#ifndef RCX_GC_AHEAD
#define RCX_GC_AHEAD 1
#endif

namespace rcx {
namespace cptbwd {

using namespace cpt;

template <int H> struct Geo {
    static constexpr int W = H, NT = W / 14, NB = H / 14, Hc = H / 2, Wc = W / 2;
    static_assert(H % 14 == 0, "whole 14 x 14 tiles");
};

struct Unit {
    int n, cb, tr, tc, c;
    bool live;
    int cl;
};
template <int H>
__device__ __forceinline__ bool decode_unit(Unit& u, int N, int C)
{
    using G = Geo<H>;
    const int lane = (int)(threadIdx.x & 63);
    const int nb = (C + 63) / 64;
    const unsigned total = (unsigned)N * nb * G::NB * G::NT;
    constexpr unsigned upp = (unsigned)(G::NB * G::NT);          // tile-waves per (image, channel block) plane: 16 or 4
    const unsigned wg = (upp & 3u) == 0 ? xcd_workgroup(blockIdx.x, (upp >> 2) ? (upp >> 2) : 1u, total / upp) : blockIdx.x;      // (a 14 x 14 plane is one tile: natural order)
    const unsigned unit = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (unit >= total) return false;
    u.tc = (int)(unit % (unsigned)G::NT);
    unsigned q = unit / (unsigned)G::NT;
    u.tr = (int)(q % (unsigned)G::NB);
    q /= (unsigned)G::NB;
    u.cb = (int)(q % (unsigned)nb);
    u.n = (int)(q / (unsigned)nb);
    u.c = u.cb * 64 + lane;
    u.live = u.c < C;
    u.cl = u.live ? u.c : C - 1;                      // ragged last block: the spare lanes shadow the last channel and store out of range
    return true;
}

// A row of a plane as a buffer of its own (base = the row, num_records = its bytes): everything left and right of it -- and, with zero records, a row
// outside the plane -- is out of range: loads return 0 (the zero padding of the convs, the absent terms of the adjoint sums), stores are dropped.
// Scalar arithmetic on uniform values only (see rcx_upcpt.hip::row_desc for the hazard a v_readfirstlane here would open).
__device__ __forceinline__ i32x4 row_desc(unsigned long long base, int row, int rows, int rowbytes)
{
    const bool ok = row >= 0 && row < rows;
    const unsigned long long a = base + (unsigned long long)(ok ? row : 0) * (unsigned long long)rowbytes;
    i32x4 d;
    d.x = (int)(unsigned)a;
    d.y = (int)(unsigned)(a >> 32) & 0xffff;
    d.z = ok ? rowbytes : 0;
    d.w = 0x00020000;
    return d;
}

// ---- row statements of this file (the 18-column row of 2 + 14 + 2 is rcx_cpt_kernel.h's row_load) ----
// nine float32 columns of a coarse row: -1 (vl), 0 .. 6 (vm), 7 (vr); run-time pitch
#define BW_LD9                                                                                                                        \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                   \
    "buffer_load_dword %0, %[vl], %[rs], %[t] offen\n\tbuffer_load_dword %8, %[vr], %[rs], %[t] offen\n\t"                           \
    "buffer_load_dword %1, %[vm], %[rs], %[t] offen\n\t"                                                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %2, %[vm], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %3, %[vm], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %4, %[vm], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %5, %[vm], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %6, %[vm], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %7, %[vm], %[rs], %[t] offen\n\t"
__device__ __forceinline__ void row_load9(uint32_t (&v)[9], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t;
    asm volatile(BW_LD9 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), [t] "=&s"(t)
                 : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
}
template <int PENDING> __device__ __forceinline__ void pin_row9(uint32_t (&v)[9])
{
    asm volatile("s_waitcnt vmcnt(%9)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]) : "n"(PENDING));
}
// twenty columns as 3 + 14 + 3: -3, -2, -1 (vl), 0 .. 13 (vm), 14, 15, 16 (vr) -- the three columns either side share the fate of their voffset (all out of
// range at a plane edge).  Immediate pitch where 13 pitches fit the 12-bit offset field, else the scalar offset walks along the row.
#define BW_L20_IMM(OP)                                                                                                               \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                   \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1) CPT_LI(OP, 2, "vl", "t", 2)                                               \
    CPT_LI(OP, 3, "vm", "t", 0) CPT_LI(OP, 4, "vm", "t", 1) CPT_LI(OP, 5, "vm", "t", 2) CPT_LI(OP, 6, "vm", "t", 3)                   \
    CPT_LI(OP, 7, "vm", "t", 4) CPT_LI(OP, 8, "vm", "t", 5) CPT_LI(OP, 9, "vm", "t", 6) CPT_LI(OP, 10, "vm", "t", 7)                  \
    CPT_LI(OP, 11, "vm", "t", 8) CPT_LI(OP, 12, "vm", "t", 9) CPT_LI(OP, 13, "vm", "t", 10) CPT_LI(OP, 14, "vm", "t", 11)             \
    CPT_LI(OP, 15, "vm", "t", 12) CPT_LI(OP, 16, "vm", "t", 13)                                                                       \
    CPT_LI(OP, 17, "vr", "t", 0) CPT_LI(OP, 18, "vr", "t", 1) CPT_LI(OP, 19, "vr", "t", 2)
#define BW_N(OP, d, V) "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_LG(OP, d, V, "t2")
#define BW_L20_GEN(OP)                                                                                                               \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 0, "vl", "t2") BW_N(OP, 1, "vl") BW_N(OP, 2, "vl")                                      \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 17, "vr", "t2") BW_N(OP, 18, "vr") BW_N(OP, 19, "vr")                                   \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 3, "vm", "t2") BW_N(OP, 4, "vm") BW_N(OP, 5, "vm") BW_N(OP, 6, "vm") BW_N(OP, 7, "vm")   \
    BW_N(OP, 8, "vm") BW_N(OP, 9, "vm") BW_N(OP, 10, "vm") BW_N(OP, 11, "vm") BW_N(OP, 12, "vm") BW_N(OP, 13, "vm") BW_N(OP, 14, "vm")  \
    BW_N(OP, 15, "vm") BW_N(OP, 16, "vm")
template <typename TIO, int PIXB>
__device__ __forceinline__ void row_load20w(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 13 <= 4095) {
        (void)pix; (void)t2;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(BW_L20_IMM(CPT_LDH) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(BW_L20_IMM(CPT_LD16) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(BW_L20_IMM(CPT_LD32) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else {
        (void)t;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(BW_L20_GEN(CPT_LDH) : CPT_OUT20(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(BW_L20_GEN(CPT_LD16) : CPT_OUT20(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else
            asm volatile(BW_L20_GEN(CPT_LD32) : CPT_OUT20(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    }
}
// seven float32 values, run-time pitch
__device__ __forceinline__ void row_store7(const float (&p)[7], unsigned vo, i32x4 rs, int rb, int pix)
{
    int t2;
    asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                 CPT_SG("buffer_store_dword", 0, "t2") CPT_SGN("buffer_store_dword", 1) CPT_SGN("buffer_store_dword", 2) CPT_SGN("buffer_store_dword", 3)
                 CPT_SGN("buffer_store_dword", 4) CPT_SGN("buffer_store_dword", 5) CPT_SGN("buffer_store_dword", 6)
                 : [t2] "=&s"(t2)
                 : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]),
                   [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
}

// The order in which a wave issues its vector-memory instructions -- identical for every wave: rows outside the plane still load (and return 0), the stores
// of the spare lanes are issued and dropped -- replayed at compile time: the s_waitcnt counts are exact numbers (rcx_upcpt.hip::Sched).
template <int AHEAD, bool HAS_D, bool HAS_K = true> struct SchedGX {
    static constexpr int NS = 18;
    // kind 0: g row `target`, 1: G row `target`.  Result: memory instructions issued after the awaited ones and before the wait.
    static constexpr int pending(int kind, int target)
    {
        int seq = 0, gend[NS + 8] = {}, Gend[12] = {};
        if (HAS_D) { seq += 9; Gend[0] = seq; }
        for (int r = 0; r < AHEAD && HAS_K; ++r) { seq += 18; gend[r] = seq; }
        for (int s = 0; s < NS; ++s) {
            if (HAS_K && s + AHEAD < NS) { seq += 18; gend[s + AHEAD] = seq; }
            if (HAS_D && (s & 1) == 0 && s / 2 + 1 <= 8) { seq += 9; Gend[s / 2 + 1] = seq; }
            if (kind == 0 && s == target) return seq - gend[s];
            if (HAS_D && (s & 1) == 0 && kind == 1 && s / 2 == target) return seq - Gend[s / 2];
            if (s >= 4) seq += 14;
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};
template <int AHEAD> struct SchedGC {
    static constexpr int NS = 20;
    static constexpr int pending(int target)
    {
        int seq = 0, gend[NS + 8] = {};
        for (int r = 0; r < AHEAD; ++r) { seq += 20; gend[r] = seq; }
        for (int s = 0; s < NS; ++s) {
            if (s + AHEAD < NS) { seq += 20; gend[s + AHEAD] = seq; }
            if (s == target) return seq - gend[s];
            if (s >= 7 && ((s - 4) & 1) == 1) seq += 7;          // gT row o = s - 4 (odd, >= 3) closes coarse row (o - 3) / 2: seven stores
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// k_bwd_gx: out = K^ g + D^T G on a 14 x 14 tile.  g: N x H x H x C (TG), G: N x H/2 x H/2 x C float32 (HAS_D), out: TO.
// wf: this conv's 25 x C flipped taps; wd: the shared down conv's 25 x C taps (as the forward applies them).
// PG / PO: bytes per pixel of g / out when known at compile time (0 = run time).  AH = g rows requested ahead of use, OCC = workgroups per CU the
// register budget is set for.
// HAS_K = false: out = D^T G alone (the input gradient of a plain stride-2 conv5: RecAttn2d's `down` conv in a training step); g and wf are not read.
template <typename TG, typename TO, int H, bool HAS_D, int PG, int PO, int AH = RCX_GX_AHEAD, int OCC = 2, bool HAS_K = true>
__global__ __launch_bounds__(256, OCC)
void k_bwd_gx(const TG* __restrict__ g, const float* __restrict__ Gc, TO* __restrict__ out, const float* __restrict__ wf,
              const float* __restrict__ wd, int N, int C)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, Hc = GE::Hc, Wc = GE::Wc, AHEAD = AH, NS = 18, GSZ = (int)sizeof(TG), OSZ = (int)sizeof(TO);
    using S = SchedGX<AHEAD, HAS_D, HAS_K>;
    static_assert(HAS_K || HAS_D, "nothing to compute");
    Unit U;
    if (!decode_unit<H>(U, N, C)) return;
    const int pixg = PG ? PG : C * GSZ, pixo = PO ? PO : C * OSZ, pixf = C * 4;
    const unsigned long long gbase = (unsigned long long)(reinterpret_cast<const char*>(g) + (size_t)U.n * H * W * pixg);
    const unsigned long long Gbase = (unsigned long long)(reinterpret_cast<const char*>(Gc) + (size_t)U.n * Hc * Wc * pixf);
    const unsigned long long obase = (unsigned long long)(reinterpret_cast<char*>(out) + (size_t)U.n * H * W * pixo);
    // g columns: tile column 0 of the left-most tile has its two halo columns at a negative offset = out of range = 0; the right-most likewise
    const unsigned voffM = (unsigned)((14 * U.tc) * pixg + U.cl * GSZ);
    const unsigned voffL = voffM - 2u * (unsigned)pixg, voffR = voffM + 14u * (unsigned)pixg;
    const unsigned GvM = (unsigned)((7 * U.tc) * pixf + U.cl * 4), GvL = GvM - (unsigned)pixf, GvR = GvM + 7u * (unsigned)pixf;
    auto load_row = [&](uint32_t (&raw)[18], int s) { row_load<TG, PG>(raw, voffL, voffM, voffR, row_desc(gbase, 14 * U.tr - 2 + s, H, W * pixg), 0, pixg); };
    auto load_G = [&](uint32_t (&raw)[9], int m) { row_load9(raw, GvL, GvM, GvR, row_desc(Gbase, 7 * U.tr - 1 + m, Hc, Wc * pixf), 0, pixf); };

    uint32_t rg[NS][18];
    uint32_t rG[2][9];
    if constexpr (HAS_D) load_G(rG[0], 0);
    if constexpr (HAS_K) sfor<AHEAD>([&](auto sc) { load_row(rg[decltype(sc)::value], decltype(sc)::value); });
    const __amdgpu_buffer_rsrc_t zsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_K ? wf : wd), 0, 0, 0x00020000);            // no bias: zero records
    Taps tf, td;
    if constexpr (HAS_K) load_taps(tf, __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, 25 * C * 4, 0x00020000), zsrc, 0, C, U.cl);
    if constexpr (HAS_D) load_taps(td, __builtin_amdgcn_make_buffer_rsrc((void*)wd, 0, 25 * C * 4, 0x00020000), zsrc, 0, C, U.cl);
    // the taps land HERE, on every path (rcx_upcpt.hip: left to the compiler their waits sink into the loop and drain the row prefetch)
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        if constexpr (HAS_K) { pin(tf.p[u][0]); pin(tf.p[u][1]); pin(tf.p[u][2]); }
        if constexpr (HAS_D) { pin(td.p[u][0]); pin(td.p[u][1]); pin(td.p[u][2]); }
    }
    const unsigned yoff = U.live ? (unsigned)((14 * U.tc) * pixo + U.c * OSZ) : 0x80000000u;
    f32x2 acc[5][7];

    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (HAS_K && s + AHEAD < NS) load_row(rg[s + AHEAD], s + AHEAD);
        // D^T is input-row stationary too: coarse row i (local m = i + 1) feeds the fine rows o = 2i - 2 .. 2i + 2 (tap row u = o + 2 - 2i), exactly
        // the accumulator rows in flight in iteration s = 2i + 2 = 2m: G row m is taken there, and requested one even iteration earlier
        if constexpr (HAS_D && (s & 1) == 0 && s / 2 + 1 <= 8) load_G(rG[(s / 2 + 1) & 1], s / 2 + 1);
        if constexpr (HAS_K) {
    pin_row<S::cap(S::pending(0, s))>(rg[s]);
            f32x2 row[9], odd[8];
#pragma unroll
            for (int k = 0; k < 9; ++k) row[k] = f32x2{raw_f32<TG>(rg[s][2 * k]), raw_f32<TG>(rg[s][2 * k + 1])};
#pragma unroll
            for (int j = 0; j < 8; ++j) odd[j] = shift1(row[j], row[j + 1]);
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int o = s - u;
                if (o < 0 || o > 13) continue;
                f32x2(&a)[7] = acc[o % 5];
                if (u == 0) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = row[j] * splat(tf.at(0, 0));
                } else {
#pragma unroll
                    for (int j = 0; j < 7; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), a[j]);
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
            }
        } else if constexpr (s <= 13) {                           // no conv: row s of the tile opens empty
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[s % 5][j] = f32x2{0.f, 0.f};
        }
        if constexpr (HAS_D && (s & 1) == 0 && s / 2 <= 8) {
            constexpr int m = s / 2;
            pin_row9<S::cap(S::pending(1, m))>(rG[m & 1]);
            float Gm[9];                                    // zero outside the plane (the descriptors): the adjoint sums over existing coarse pixels only
#pragma unroll
            for (int q = 0; q < 9; ++q) Gm[q] = __uint_as_float(rG[m & 1][q]);
            // out(2j, 2j+1) += G[i][j+1] (w0, w1) + G[i][j] (w2, w3) + (G[i][j-1] w4, 0); coarse column j of the tile = local q = j + 1
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int o = 2 * (m - 1) - 2 + u;
                if (o < 0 || o > 13) continue;
                f32x2(&a)[7] = acc[o % 5];
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(splat(Gm[j + 2]), td.p[u][0], a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(splat(Gm[j + 1]), td.p[u][1], a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j].x = fmaf(Gm[j], td.p[u][2].x, a[j].x);
            }
        }
        if constexpr (s >= 4) {
            constexpr int o = s - 4;
            RowSt<TO, PO>::st(acc[o % 5], yoff, row_desc(obase, 14 * U.tr + o, H, W * pixo), 0, pixo);
        }
#pragma unroll
        for (int o = 0; o < 14; ++o) if (o > s - 4 && o <= s) pin(acc[o % 5]);
        CPT_FENCE;
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_bwd_gc: gC = R^T (K^ g) on the 7 x 7 coarse pixels of a 14 x 14 tile.  g: N x H x H x C (TG); gC: N x H/2 x H/2 x C float32.
// Local frames: g row s = image row r0 - 3 + s, column q = image column c0 - 3 + q (20 x 20); gT row o = image row r0 - 1 + o,
// column p = image column c0 - 1 + p (16 x 16: pairs start at an ODD image column); gT(o, p) = sum_{u,v} K^[u][v] g(o + u, p + v).
template <int MODE, typename TG, int H, int PG, int AH = RCX_GC_AHEAD, int OCC = 2>
__global__ __launch_bounds__(256, OCC)
void k_bwd_gc(const TG* __restrict__ g, float* __restrict__ gC, const float* __restrict__ wf, int N, int C)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, Hc = GE::Hc, Wc = GE::Wc, AHEAD = AH, NS = 20, GSZ = (int)sizeof(TG);
    constexpr float WQ = MODE == 1 ? 0.f : 0.25f, WT = MODE == 1 ? 1.f : 0.75f;
    using S = SchedGC<AHEAD>;
    Unit U;
    if (!decode_unit<H>(U, N, C)) return;
    const int r0 = 14 * U.tr, c0 = 14 * U.tc;
    const int pixg = PG ? PG : C * GSZ, pixf = C * 4;
    const unsigned long long gbase = (unsigned long long)(reinterpret_cast<const char*>(g) + (size_t)U.n * H * W * pixg);
    const unsigned long long cbase = (unsigned long long)(reinterpret_cast<char*>(gC) + (size_t)U.n * Hc * Wc * pixf);
    const unsigned voffM = (unsigned)(c0 * pixg + U.cl * GSZ);
    const unsigned voffL = voffM - 3u * (unsigned)pixg, voffR = voffM + 14u * (unsigned)pixg;
    auto load_row = [&](uint32_t (&raw)[20], int s) { row_load20w<TG, PG>(raw, voffL, voffM, voffR, row_desc(gbase, r0 - 3 + s, H, W * pixg), 0, pixg); };

    uint32_t rg[NS][20];
    sfor<AHEAD>([&](auto sc) { load_row(rg[decltype(sc)::value], decltype(sc)::value); });
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, 25 * C * 4, 0x00020000);
    Taps tf;
    load_taps(tf, wsrc, __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, 0, 0x00020000), 0, C, U.cl);
#pragma unroll
    for (int u = 0; u < 5; ++u) { pin(tf.p[u][0]); pin(tf.p[u][1]); pin(tf.p[u][2]); }

    const bool left = c0 == 0, right = c0 + 14 == W, top = r0 == 0, bottom = r0 + 14 == H;
    // R^T, vertical: gT row o feeds coarse rows (o even) o/2 with WQ' and o/2 - 1 with WT', (o odd) (o-1)/2 with WT' and (o-3)/2 with WQ';
    // the clamped borders move the outside row's weight to the edge row: rows 0 / 15 get 0, rows 1 / 14 get WT + WQ there
    const float wv_o0 = top ? 0.f : WQ, wv_o1 = top ? WT + WQ : WT, wv_o14 = bottom ? WT + WQ : WT, wv_o15 = bottom ? 0.f : WQ;
    // horizontal: coarse column j = dot(A[j], (WQ, WT)) + dot(A[j+1], (WT, WQ)) over the pairs A[k] = (p = 2k, 2k + 1); same border rule
    const f32x2 hA0 = f32x2{left ? 0.f : WQ, left ? WT + WQ : WT}, hB6 = f32x2{right ? WT + WQ : WT, right ? 0.f : WQ};
    const unsigned yoff = U.live ? (unsigned)((7 * U.tc) * pixf + U.c * 4) : 0x80000000u;

    f32x2 acc[5][8];
    float V[2][7];
    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < NS) load_row(rg[s + AHEAD], s + AHEAD);
        pin_row20<S::cap(S::pending(s))>(rg[s]);
        f32x2 row[10], odd[9];
#pragma unroll
        for (int k = 0; k < 10; ++k) row[k] = f32x2{raw_f32<TG>(rg[s][2 * k]), raw_f32<TG>(rg[s][2 * k + 1])};
#pragma unroll
        for (int j = 0; j < 9; ++j) odd[j] = shift1(row[j], row[j + 1]);
        // g row s meets gT rows o = s - u
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int o = s - u;
            if (o < 0 || o > 15) continue;
            f32x2(&a)[8] = acc[o % 5];
            if (u == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = row[j] * splat(tf.at(0, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), a[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
        }
        if constexpr (s >= 4) {
            constexpr int o = s - 4;                        // gT row o is complete: its horizontal adjoint first (7 values), then the vertical one on those
            const f32x2(&a)[8] = acc[o % 5];
            float h[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const f32x2 wa = j == 0 ? hA0 : f32x2{WQ, WT}, wb = j == 6 ? hB6 : f32x2{WT, WQ};
                const f32x2 t = pfma(a[j + 1], wb, a[j] * wa);
                h[j] = t.x + t.y;
            }
            // coarse row i collects o = 2i (first), 2i + 1, 2i + 2, 2i + 3 (last)
            if constexpr ((o & 1) == 0) {
                if constexpr (o / 2 <= 6) {                 // opens coarse row o / 2
                    const float w = o == 0 ? wv_o0 : WQ;
#pragma unroll
                    for (int j = 0; j < 7; ++j) V[(o / 2) & 1][j] = h[j] * w;
                }
                if constexpr (o / 2 - 1 >= 0) {
                    const float w = o == 14 ? wv_o14 : WT;
#pragma unroll
                    for (int j = 0; j < 7; ++j) V[(o / 2 - 1) & 1][j] = fmaf(h[j], w, V[(o / 2 - 1) & 1][j]);
                }
            } else {
                if constexpr ((o - 1) / 2 <= 6) {
                    const float w = o == 1 ? wv_o1 : WT;
#pragma unroll
                    for (int j = 0; j < 7; ++j) V[((o - 1) / 2) & 1][j] = fmaf(h[j], w, V[((o - 1) / 2) & 1][j]);
                }
                if constexpr (o >= 3) {
                    constexpr int i = (o - 3) / 2;          // closes coarse row i
                    const float w = o == 15 ? wv_o15 : WQ;
                    float outv[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) outv[j] = fmaf(h[j], w, V[i & 1][j]);
                    row_store7(outv, yoff, row_desc(cbase, 7 * U.tr + i, Hc, Wc * pixf), 0, pixf);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) if (o > s - 4 && o <= s) pin(acc[o % 5]);
#pragma unroll
        for (int j = 0; j < 7; ++j) { pin(V[0][j]); pin(V[1][j]); }
        CPT_FENCE;
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_wgrad_d: the shared stride-2 conv's weight gradient from one level, gW_d[u][v] += sum_{o,i} G[o][i] a[2o + u - 2][2i + v - 2]
// (model/recnext.py:21, :28), on the same tiles: a wave = 64 channels of a 14 x 14 tile of a = the 7 x 7 pixels of G it produced.
// rcx_cplwgrad.hip's k_wgrad2_cpl gives a wave a 14 x 14 tile of G (28 x 28 of a): 512 waves at 128 x 64 x 56 x 56, 330 registers, one
// wave on every second SIMD -- 50 us for 0.1 GB.  Here: four times the waves, G's 49 values resident, a-row stationary with the taps paired
// against a's aligned pairs (rcx_cplbwd_pieces.h), rows requested AHEAD before use.  The NT column tiles of a band share a workgroup and add
// their 26 sums per channel through LDS in a fixed order: one partial row per (image, 14-row band), reduced over the batch by k_wgrad_reduce_jobs.
// seven float32 columns, run-time pitch
#define BW_LD7                                                                                                                        \
    "s_add_i32 %[t], %[rb], 0\n\tbuffer_load_dword %0, %[vo], %[rs], %[t] offen\n\t"                                                 \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %1, %[vo], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %2, %[vo], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %3, %[vo], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %4, %[vo], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %5, %[vo], %[rs], %[t] offen\n\t"                                             \
    "s_add_i32 %[t], %[t], %[pix]\n\tbuffer_load_dword %6, %[vo], %[rs], %[t] offen\n\t"
__device__ __forceinline__ void row_load7(uint32_t (&v)[7], unsigned vo, i32x4 rs, int rb, int pix)
{
    int t;
    asm volatile(BW_LD7 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), [t] "=&s"(t)
                 : [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
}
template <int PENDING> __device__ __forceinline__ void pin_row7(uint32_t (&v)[7])
{
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) : "n"(PENDING));
}
// the K x K tap sums of a stride-2 conv: tap pairs (0,1) (2,3) .. ((K-1), -)
template <int K> struct DAccK {
    static constexpr int KP = (K + 1) / 2;
    f32x2 a[K][KP];
    f32x2 bs;
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
            for (int k = 0; k < KP; ++k) a[u][k] = f32x2{0.f, 0.f};
        bs = f32x2{0.f, 0.f};
    }
    __device__ __forceinline__ float tap(int u, int v) const { return (v & 1) ? a[u][v >> 1].y : a[u][v >> 1].x; }
    __device__ __forceinline__ float bias() const { return bs.x + bs.y; }
};
template <int AHEAD, int K> struct SchedWD {
    static constexpr int NS = 14 + K - 2, NC = K == 5 ? 18 : 20;
    static constexpr int pending(int target)
    {
        int seq = 0, aend[NS + 8] = {};
        for (int r = 0; r < AHEAD; ++r) { seq += NC; aend[r] = seq; }
        for (int s = 0; s < NS; ++s) {
            if (s + AHEAD < NS) { seq += NC; aend[s + AHEAD] = seq; }
            if (s == target) return seq - aend[s];
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};
template <typename TA, int PA> __device__ __forceinline__ void wd_row_load(uint32_t (&v)[18], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int pix) { row_load<TA, PA>(v, vl, vm, vr, rs, 0, pix); }
template <typename TA, int PA> __device__ __forceinline__ void wd_row_load(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int pix) { row_load20w<TA, PA>(v, vl, vm, vr, rs, 0, pix); }
template <int PENDING> __device__ __forceinline__ void wd_row_pin(uint32_t (&v)[18]) { pin_row<PENDING>(v); }
template <int PENDING> __device__ __forceinline__ void wd_row_pin(uint32_t (&v)[20]) { pin_row20<PENDING>(v); }

// K = 5, MULT = 1: the block's shared down conv.  K = 7, MULT = 2: Downsample's conv (nn.Conv2d(C/2, C, 7, stride 2, groups = C/2): a lane is an OUTPUT channel and
// reads input channel c / 2; C counts g's channels), on the 56 / 28 / 14 planes -- rcx_cplwgrad.hip's k_wgrad2_cpl<7, 2> and the generic k_wgrad_rows took
// 70 + 44 + 83 us for RecNeXt-M3's three at batch 128.
template <typename TA, int H, int PA, int K = 5, int MULT = 1>
__global__ __launch_bounds__(64 * (H / 14), 2)
void k_wgrad_d(const TA* __restrict__ a, const float* __restrict__ Gc, float* __restrict__ partial, int N, int C)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, NT = GE::NT, NB = GE::NB, Hc = GE::Hc, Wc = GE::Wc, AHEAD = 2, P = K / 2, NS = 14 + K - 2, NC = K == 5 ? 18 : 20, KP = (K + 1) / 2, KK1 = K * K + 1;
    constexpr int ASZ = (int)sizeof(TA);
    static_assert(K == 5 || K == 7, "tap rows of 18 (2 + 14 + 2) or 20 (3 + 14 + 3) columns");
    using S = SchedWD<AHEAD, K>;
    __shared__ float red[NT][KK1][64];
    const int tile = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    const int nb = (C + 63) / 64;
    const unsigned unit = blockIdx.x;
    const int cb = __builtin_amdgcn_readfirstlane((int)(unit % (unsigned)nb)), band = __builtin_amdgcn_readfirstlane((int)((unit / (unsigned)nb) % (unsigned)NB)),
              n = __builtin_amdgcn_readfirstlane((int)(unit / (unsigned)(nb * NB)));
    const int c = cb * 64 + lane;
    const bool live = c < C;
    const int cl = live ? c : C - 1;
    const int Ca = C / MULT;
    const int pixa = PA ? PA : Ca * ASZ, pixf = C * 4;
    const unsigned long long abase = (unsigned long long)(reinterpret_cast<const char*>(a) + (size_t)n * H * W * pixa);
    const unsigned long long Gbase = (unsigned long long)(reinterpret_cast<const char*>(Gc) + (size_t)n * Hc * Wc * pixf);
    const unsigned voffM = (unsigned)((14 * tile) * pixa + (cl / MULT) * ASZ);
    const unsigned voffL = voffM - (unsigned)P * (unsigned)pixa, voffR = voffM + 14u * (unsigned)pixa;
    const unsigned Gvo = (unsigned)((7 * tile) * pixf + cl * 4);
    auto load_row = [&](uint32_t (&raw)[NC], int s) { wd_row_load<TA, PA>(raw, voffL, voffM, voffR, row_desc(abase, 14 * band - P + s, H, W * pixa), pixa); };

    uint32_t ra[NS][NC], rG[7][7];
    sfor<7>([&](auto oc) { row_load7(rG[decltype(oc)::value], Gvo, row_desc(Gbase, 7 * band + decltype(oc)::value, Hc, Wc * pixf), 0, pixf); });
    sfor<AHEAD>([&](auto sc) { load_row(ra[decltype(sc)::value], decltype(sc)::value); });
    DAccK<K> acc;
    acc.zero();
    float G[7][7];
    sfor<7>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
        pin_row7<AHEAD * NC>(rG[o]);                             // every G row is older than the first a rows
#pragma unroll
        for (int i = 0; i < 7; ++i) { G[o][i] = __uint_as_float(rG[o][i]); if (i & 1) acc.bs.y += G[o][i]; else acc.bs.x += G[o][i]; }
    });
    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < NS) load_row(ra[s + AHEAD], s + AHEAD);
        wd_row_pin<S::cap(S::pending(s))>(ra[s]);
        f32x2 ar[NC / 2];
#pragma unroll
        for (int m = 0; m < NC / 2; ++m) ar[m] = f32x2{raw_f32<TA>(ra[s][2 * m]), raw_f32<TA>(ra[s][2 * m + 1])};
        // a row s = 2o + u: the G rows o whose window covers it, tap row u; G column i against the pairs ar[i + k] = a columns 2i + 2k, 2i + 2k + 1 (local column q = image column c0 - P + q)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int o = 0; o < 7; ++o) {
                const int u = s - 2 * o;
                if (u < 0 || u > K - 1) continue;
                const f32x2 gv = splat(G[o][i]);
#pragma unroll
                for (int k = 0; k < KP; ++k) acc.a[u][k] = pfma(gv, ar[i + k], acc.a[u][k]);
            }
        }
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
            for (int k = 0; k < KP; ++k) pin(acc.a[u][k]);
        CPT_FENCE;
    });
#pragma unroll
    for (int u = 0; u < K; ++u)
#pragma unroll
        for (int v = 0; v < K; ++v) red[tile][u * K + v][lane] = acc.tap(u, v);
    red[tile][K * K][lane] = acc.bias();
    __syncthreads();
    if (tile == 0 && live) {
        float* q = partial + ((size_t)(n * NB + band) * KK1) * C + c;
#pragma unroll
        for (int t = 0; t < KK1; ++t) {
            float sum = red[0][t][lane];
#pragma unroll
            for (int w = 1; w < NT; ++w) sum += red[w][t][lane];
            q[(size_t)t * C] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_wgrad_k: weight gradient of a stride-1 5x5 conv whose input is T = a + R(coarse) (the final conv and the level convs, model/recnext.py:33-34; MODE 2:
// T = a, a plain depthwise conv), gW[u][v] = sum_{t,c} g[t][c] T[t+u-2][c+v-2], on the same tiles.  rcx_cplwgrad.hip's k_wgrad_cpl keeps a five-row ring
// of T and two accumulator sets (358 registers: one wave per SIMD, 60 us at 128 x 64 x 56 x 56); here T rows are built once and used at once -- T-row
// stationary against a five-row ring of g (7 pairs a row instead of 9) -- and the odd gradient columns read T's pairs shifted by one pixel (eight
// v_pk_mov per row); and the packed FMAs pair PIXELS, not taps: acc[u][v] += (g[t][c], g[t][c+1]) * (T[t+u][c+v], T[t+u][c+v+1]), both halves of an
// accumulator belong to the same tap -- no broadcast operand (a broadcast of an odd column costs a move and an aligned register pair each), 175 packed FMAs
// per (T row, g row) instead of 210.  T local column q = image column c0 - 2 + q; hand-issued rows; two waves per SIMD.
// eleven float32 columns of the coarse plane at per-column scalar offsets (clamped into the plane by the caller: ATen's border rule)
#define BW_C(d) "s_add_i32 %[t], %[rb], %[c" #d "]\n\tbuffer_load_dword %" #d ", %[vo], %[rs], %[t] offen\n\t"
__device__ __forceinline__ void coarse_load11(uint32_t (&v)[11], unsigned vo, i32x4 rs, int rb, const int (&ck)[11])
{
    int t;
    asm volatile(BW_C(0) BW_C(1) BW_C(2) BW_C(3) BW_C(4) BW_C(5) BW_C(6) BW_C(7) BW_C(8) BW_C(9) BW_C(10)
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), [t] "=&s"(t)
                 : [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [c0] "s"(ck[0]), [c1] "s"(ck[1]), [c2] "s"(ck[2]), [c3] "s"(ck[3]), [c4] "s"(ck[4]), [c5] "s"(ck[5]),
                   [c6] "s"(ck[6]), [c7] "s"(ck[7]), [c8] "s"(ck[8]), [c9] "s"(ck[9]), [c10] "s"(ck[10]) : "scc");
}
template <int PENDING> __device__ __forceinline__ void pin_row11(uint32_t (&v)[11])
{
    asm volatile("s_waitcnt vmcnt(%11)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]) : "n"(PENDING));
}
// fourteen columns of a row (the tile's own: always inside the plane), run-time pitch
#define BW_L14(OP)                                                                                                                   \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 0, "vm", "t2") BW_N(OP, 1, "vm") BW_N(OP, 2, "vm") BW_N(OP, 3, "vm") BW_N(OP, 4, "vm")   \
    BW_N(OP, 5, "vm") BW_N(OP, 6, "vm") BW_N(OP, 7, "vm") BW_N(OP, 8, "vm") BW_N(OP, 9, "vm") BW_N(OP, 10, "vm") BW_N(OP, 11, "vm")     \
    BW_N(OP, 12, "vm") BW_N(OP, 13, "vm")
#define BW_OUT14(v) "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), \
                    "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13])
template <typename TIO>
__device__ __forceinline__ void row_load14(uint32_t (&v)[14], unsigned vm, i32x4 rs, int rb, int pix)
{
    int t2;
    if constexpr (std::is_same<TIO, f16_t>::value) asm volatile(BW_L14(CPT_LDH) : BW_OUT14(v), [t2] "=&s"(t2) : [vm] "v"(vm), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    else if constexpr (sizeof(TIO) == 2) asm volatile(BW_L14(CPT_LD16) : BW_OUT14(v), [t2] "=&s"(t2) : [vm] "v"(vm), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    else asm volatile(BW_L14(CPT_LD32) : BW_OUT14(v), [t2] "=&s"(t2) : [vm] "v"(vm), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
}
template <int PENDING> __device__ __forceinline__ void pin_row14(uint32_t (&v)[14])
{
    asm volatile("s_waitcnt vmcnt(%14)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]) : "n"(PENDING));
}

// which coarse rows T row s reads (local il = coarse row 7 band - 2 + il, clamped): the exact 2x step -- even image row -> (s/2, s/2 + 1) with (1/4, 3/4),
// odd -> ((s+1)/2, (s+1)/2 + 1) with (3/4, 1/4); nearest: (s + 2) >> 1
template <int MODE> struct CoarseRows {
    static constexpr int hi(int s) { return MODE == 1 ? (s + 2) >> 1 : ((s & 1) ? (s + 1) / 2 : s / 2) + 1; }       // the highest local coarse row T row s reads
    static constexpr int first_new(int s) { return s == 0 ? (MODE == 1 ? hi(0) : 0) : (hi(s) > hi(s - 1) ? hi(s) : -1); }   // s = 0 (bilinear): rows 0 and 1
    static constexpr int count_new(int s) { return MODE == 2 ? 0 : (s == 0 ? (MODE == 1 ? 1 : 2) : (hi(s) > hi(s - 1) ? 1 : 0)); }
};
// issue order of one tile: prologue = coarse rows of T row 0, a rows 0 .. AHEAD-1, g row 0; iteration s = [a row s + AHEAD] [coarse rows new to T row s + 1]
// [g row s + 1]; waits in the order a row s, coarse rows of T row s, g row s
template <int MODE, int AHEAD> struct SchedWK {
    static constexpr int NS = 18;
    // kind 0: a row, 1: the coarse rows T row `target` is the first to read, 2: g row
    static constexpr int pending(int kind, int target)
    {
        using CR = CoarseRows<MODE>;
        int seq = 0, aend[NS + 8] = {}, cend[NS + 2] = {}, gend[16] = {};
        seq += 11 * CR::count_new(0); cend[0] = seq;
        for (int r = 0; r < AHEAD; ++r) { seq += 18; aend[r] = seq; }
        seq += 14; gend[0] = seq;
        for (int s = 0; s < NS; ++s) {
            if (s + AHEAD < NS) { seq += 18; aend[s + AHEAD] = seq; }
            if (s + 1 < NS) { seq += 11 * CR::count_new(s + 1); cend[s + 1] = seq; }
            if (s + 1 < 14) { seq += 14; gend[s + 1] = seq; }
            if (kind == 0 && s == target) return seq - aend[s];
            if (kind == 1 && s == target) return seq - cend[s];
            if (kind == 2 && s == target) return seq - gend[s];
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};

template <int MODE, typename TA, typename TG, int H, int PA>
__global__ __launch_bounds__(64 * (H / 14), 2)
void k_wgrad_k(const TA* __restrict__ a, const float* __restrict__ coarse, const TG* __restrict__ g, float* __restrict__ partial, int N, int C)
{
    using GE = Geo<H>;
    using CR = CoarseRows<MODE>;
    constexpr int W = GE::W, NT = GE::NT, NB = GE::NB, Hc = GE::Hc, Wc = GE::Wc, AHEAD = 1, NS = 18, ASZ = (int)sizeof(TA), GSZ = (int)sizeof(TG);
    using S = SchedWK<MODE, AHEAD>;
    __shared__ float red[NT][26][64];
    const int tile = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    const int nb = (C + 63) / 64;
    const unsigned unit = blockIdx.x;
    const int cb = __builtin_amdgcn_readfirstlane((int)(unit % (unsigned)nb)), band = __builtin_amdgcn_readfirstlane((int)((unit / (unsigned)nb) % (unsigned)NB)),
              n = __builtin_amdgcn_readfirstlane((int)(unit / (unsigned)(nb * NB)));
    const int c = cb * 64 + lane;
    const bool live = c < C;
    const int cl = live ? c : C - 1;
    const int r0 = 14 * band, c0 = 14 * tile;
    const int pixa = PA ? PA : C * ASZ, pixg = C * GSZ, pixf = C * 4;
    const unsigned long long abase = (unsigned long long)(reinterpret_cast<const char*>(a) + (size_t)n * H * W * pixa);
    const unsigned long long gbase = (unsigned long long)(reinterpret_cast<const char*>(g) + (size_t)n * H * W * pixg);
    const unsigned voffM = (unsigned)(c0 * pixa + cl * ASZ);
    const unsigned voffL = voffM - 2u * (unsigned)pixa, voffR = voffM + 14u * (unsigned)pixa;
    const unsigned gvo = (unsigned)(c0 * pixg + cl * GSZ);
    auto load_a = [&](uint32_t (&raw)[18], int s) { row_load<TA, PA>(raw, voffL, voffM, voffR, row_desc(abase, r0 - 2 + s, H, W * pixa), 0, pixa); };
    auto load_g = [&](uint32_t (&raw)[14], int t) { row_load14<TG>(raw, gvo, row_desc(gbase, r0 + t, H, W * pixg), 0, pixg); };
    // the coarse plane of this image as one buffer: its row and column indices are clamped, never out of range
    i32x4 csrc;
    int ck[11];
    if constexpr (MODE != 2) {
        const unsigned long long ca = (unsigned long long)(reinterpret_cast<const char*>(coarse) + (size_t)n * Hc * Wc * pixf);
        csrc.x = (int)(unsigned)ca;
        csrc.y = (int)(unsigned)(ca >> 32) & 0xffff;
        csrc.z = Hc * Wc * pixf;
        csrc.w = 0x00020000;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            int col = 7 * tile - 2 + k;
            col = col < 0 ? 0 : (col > Wc - 1 ? Wc - 1 : col);
            ck[k] = __builtin_amdgcn_readfirstlane(col * pixf);
        }
    }
    const unsigned cvo = (unsigned)(cl * 4);
    auto load_c = [&](uint32_t (&raw)[11], int il) {
        int i = 7 * band - 2 + il;
        i = i < 0 ? 0 : (i > Hc - 1 ? Hc - 1 : i);
        coarse_load11(raw, cvo, csrc, __builtin_amdgcn_readfirstlane(i * Wc * pixf), ck);
    };

    uint32_t ra[NS][18], rg[14][14], rc[11][11];
    f32x2 Hr[2][9];
    // the resized coarse row is zero outside the plane (the conv pads T with zeros): column masks of the two halo pairs, row mask per T row
    const f32x2 keep_lo = splat(c0 == 0 ? 0.f : 1.f), keep_hi = splat(c0 + 14 == W ? 0.f : 1.f);
    auto build_h = [&](auto ic) {
        constexpr int il = decltype(ic)::value;
        float cv[11];
#pragma unroll
        for (int jl = 0; jl < 11; ++jl) cv[jl] = __uint_as_float(rc[il][jl]);
#pragma unroll
        for (int m = 0; m < 9; ++m) {                            // T columns q = 2m (even image column), 2m + 1 (odd)
            if (MODE == 1) Hr[il & 1][m] = f32x2{cv[m + 1], cv[m + 1]};
            else Hr[il & 1][m] = f32x2{fmaf(0.25f, cv[m], 0.75f * cv[m + 1]), fmaf(0.75f, cv[m + 1], 0.25f * cv[m + 2])};
        }
        Hr[il & 1][0] = Hr[il & 1][0] * keep_lo;
        Hr[il & 1][8] = Hr[il & 1][8] * keep_hi;
    };

    // prologue, in the order SchedWK replays
    if constexpr (MODE == 0) { load_c(rc[0], 0); load_c(rc[1], 1); }
    if constexpr (MODE == 1) load_c(rc[1], 1);
    sfor<AHEAD>([&](auto sc) { load_a(ra[decltype(sc)::value], decltype(sc)::value); });
    load_g(rg[0], 0);

    f32x2 acc[5][5], bs = f32x2{0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 5; ++v) acc[u][v] = f32x2{0.f, 0.f};
    f32x2 Gr[5][7];

    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < NS) load_a(ra[s + AHEAD], s + AHEAD);
        if constexpr (s + 1 < NS && CR::count_new(s + 1) > 0) load_c(rc[CR::hi(s + 1)], CR::hi(s + 1));
        if constexpr (s + 1 < 14) load_g(rg[s + 1], s + 1);
        pin_row<S::cap(S::pending(0, s))>(ra[s]);
        f32x2 T[9], To[8];
#pragma unroll
        for (int m = 0; m < 9; ++m) T[m] = f32x2{raw_f32<TA>(ra[s][2 * m]), raw_f32<TA>(ra[s][2 * m + 1])};
        if constexpr (MODE != 2) {
            if constexpr (CR::count_new(s) > 0) {
                pin_row11<S::cap(S::pending(1, s))>(rc[CR::hi(s)]);     // the youngest of the rows T row s is the first to read
                if constexpr (s == 0 && MODE == 0) {                   // ... and the older one: landed before it, but its registers must be tied to a wait as well
                    pin_row11<S::cap(S::pending(1, s))>(rc[0]);
                    build_h(IC<0>{});
                }
                build_h(IC<CR::hi(s)>{});
            }
            // rows outside the plane are zero padding of T: a read 0 there (descriptor), the resized coarse row is masked
            const int r = r0 - 2 + s;
            const f32x2 keep = splat((s >= 2 && s < 16) || (r >= 0 && r < H) ? 1.f : 0.f);
            if constexpr (MODE == 1) {
#pragma unroll
                for (int m = 0; m < 9; ++m) T[m] = (s < 2 || s >= 16) ? pfma(Hr[CR::hi(s) & 1][m], keep, T[m]) : T[m] + Hr[CR::hi(s) & 1][m];
            } else {
                constexpr int i0 = CR::hi(s) - 1;
                constexpr float w0 = (s & 1) ? 0.75f : 0.25f, w1 = 1.f - w0;
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    if (s < 2 || s >= 16) T[m] = pfma(pfma(splat(w1), Hr[(i0 + 1) & 1][m], splat(w0) * Hr[i0 & 1][m]), keep, T[m]);
                    else T[m] = pfma(splat(w1), Hr[(i0 + 1) & 1][m], pfma(splat(w0), Hr[i0 & 1][m], T[m]));
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) To[p] = shift1(T[p], T[p + 1]);
        if constexpr (s < 14) {                                  // g row s enters the ring (slot s % 5: row s - 5 met its last T row in iteration s - 1)
            pin_row14<S::cap(S::pending(2, s))>(rg[s]);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                Gr[s % 5][j] = f32x2{raw_f32<TG>(rg[s][2 * j]), raw_f32<TG>(rg[s][2 * j + 1])};
                bs = bs + Gr[s % 5][j];
            }
        }
        // pixel pairs, no broadcast: acc[u][v] (both halves belong to tap (u, v); added at the end) += (g[t][2j], g[t][2j+1]) * (T[2j + v], T[2j + v + 1]):
        // T's aligned pairs for even v, the pairs shifted by one pixel for odd v
#pragma unroll
        for (int j = 0; j < 7; ++j) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int t = s - u;
                if (t < 0 || t > 13) continue;
#pragma unroll
                for (int v = 0; v < 5; ++v) acc[u][v] = pfma(Gr[t % 5][j], (v & 1) ? To[j + (v - 1) / 2] : T[j + v / 2], acc[u][v]);
            }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) { pin(acc[u][0]); pin(acc[u][1]); pin(acc[u][2]); pin(acc[u][3]); pin(acc[u][4]); }
        pin(bs);                                                 // (unpinned, the bias sums sink to the end of the kernel and every g row stays live until then)
        pin(Hr[0]);
        pin(Hr[1]);
        CPT_FENCE;
    });
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 5; ++v) red[tile][u * 5 + v][lane] = acc[u][v].x + acc[u][v].y;
    red[tile][25][lane] = bs.x + bs.y;
    __syncthreads();
    if (tile == 0 && live) {
        float* q = partial + ((size_t)(n * NB + band) * 26) * C + c;
#pragma unroll
        for (int t = 0; t < 26; ++t) {
            float sum = red[0][t][lane];
#pragma unroll
            for (int w = 1; w < NT; ++w) sum += red[w][t][lane];
            q[(size_t)t * C] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_bwd_down7m2: the input gradient of Downsample's depthwise conv (nn.Conv2d(C, 2C, 7, stride 2, padding 3, groups = C), model/recnext.py:165) on the same
// tiles: gx[c][y][x] = sum over the two output channels o = 2c, 2c + 1 and the taps (u, v) with 2 oy + u - 3 = y, 2 ox + v - 3 = x of W[o][u][v] g[o][oy][ox].
// rcx_bwd.hip's k_down_bwd_input_k<7, 2> is a gather of one thread per two channels of a pixel (105 us average over the three convs of RecNeXt-M3 at batch 128:
// the largest kernel of this library in a training step).  Here a lane is an OUTPUT channel o, as in the forward's k_down7m2_cpt (a wave reads 256 contiguous
// bytes of g per pixel): it forms its own channel's share of the 14 x 14 input tile, coarse-row stationary -- coarse row m feeds the fine rows 2m - 5 .. 2m + 1
// of the tile (tap row u = y + 5 - 2m), seven accumulator rows of seven pixel pairs in flight -- with the taps paired against the pixel pair's parities:
//   (out(2j), out(2j+1)) += g[j+3] (0, w0) + g[j+2] (w1, w2) + g[j+1] (w3, w4) + g[j] (w5, w6)          (coarse column j of the tile = local q = j + 1 ...)
// and a finished row's two shares are added across the lane pair by one DPP add (quad_perm 1,0,3,2); the even lane then stores the even pixels of the row, the
// odd lane the odd ones (7 stores each, every lane busy).  g rows are hand-issued through per-row descriptors like everything else in this file.
template <int AHEAD> struct SchedDN {
    static constexpr int NM = 10;
    static constexpr int rows_done(int m) { return m < 2 ? 0 : (m == 2 ? 1 : (m == 9 ? 1 : 2)); }        // fine rows completed by coarse row m: 2m - 5, 2m - 4 inside [0, 13]
    static constexpr int pending(int target)
    {
        int seq = 0, gend[NM + 8] = {};
        for (int r = 0; r < AHEAD; ++r) { seq += 10; gend[r] = seq; }
        for (int m = 0; m < NM; ++m) {
            if (m + AHEAD < NM) { seq += 10; gend[m + AHEAD] = seq; }
            if (m == target) return seq - gend[m];
            seq += 7 * rows_done(m);
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};
// ten columns of a coarse row as 1 + 7 + 2: -1 (vl), 0 .. 6 (vm), 7, 8 (vr); run-time pitch
#define BW_L10(OP)                                                                                                                   \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 0, "vl", "t2") CPT_LG(OP, 8, "vr", "t2") CPT_LG(OP, 1, "vm", "t2")                       \
    "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_LG(OP, 9, "vr", "t2") CPT_LG(OP, 2, "vm", "t2")                                          \
    BW_N(OP, 3, "vm") BW_N(OP, 4, "vm") BW_N(OP, 5, "vm") BW_N(OP, 6, "vm") BW_N(OP, 7, "vm")
#define BW_OUT10(v) "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9])
template <typename TIO>
__device__ __forceinline__ void row_load10(uint32_t (&v)[10], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t2;
    if constexpr (std::is_same<TIO, f16_t>::value) asm volatile(BW_L10(CPT_LDH) : BW_OUT10(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    else if constexpr (sizeof(TIO) == 2) asm volatile(BW_L10(CPT_LD16) : BW_OUT10(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    else asm volatile(BW_L10(CPT_LD32) : BW_OUT10(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
}
template <int PENDING> __device__ __forceinline__ void pin_row10(uint32_t (&v)[10])
{
    asm volatile("s_waitcnt vmcnt(%10)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]) : "n"(PENDING));
}
// seven elements of one row at every second pixel (pitch2 = two pixels), low 16 bits of each register for the 16-bit types
template <typename TO>
__device__ __forceinline__ void row_store7_stride2(const uint32_t (&p)[7], unsigned vo, i32x4 rs, int rb, int pitch2)
{
    int t2;
    if constexpr (sizeof(TO) == 2)
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_short", 0, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG("buffer_store_short", 1, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t"
                     CPT_SG("buffer_store_short", 2, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG("buffer_store_short", 3, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t"
                     CPT_SG("buffer_store_short", 4, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG("buffer_store_short", 5, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t"
                     CPT_SG("buffer_store_short", 6, "t2")
                     : [t2] "=&s"(t2)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pitch2) : "scc", "memory");
    else
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_dword", 0, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG("buffer_store_dword", 1, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t"
                     CPT_SG("buffer_store_dword", 2, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG("buffer_store_dword", 3, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t"
                     CPT_SG("buffer_store_dword", 4, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG("buffer_store_dword", 5, "t2") "s_add_i32 %[t2], %[t2], %[pix]\n\t"
                     CPT_SG("buffer_store_dword", 6, "t2")
                     : [t2] "=&s"(t2)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pitch2) : "scc", "memory");
}

// g: N x H/2 x H/2 x 2 Cin float32 (the gradient of the conv's output), w: the (7, 7, 2 Cin) pack the forward applies, gx: N x H x H x Cin of TO
template <typename TO, int H>
__global__ __launch_bounds__(256, 2)
void k_bwd_down7m2(const float* __restrict__ g, TO* __restrict__ gx, const float* __restrict__ w, int N, int Cin)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, Hc = GE::Hc, Wc = GE::Wc, AHEAD = 2, NM = 10, OSZ = (int)sizeof(TO);
    using S = SchedDN<AHEAD>;
    const int Co = 2 * Cin;
    Unit U;
    if (!decode_unit<H>(U, N, Co)) return;                       // a wave = 64 OUTPUT channels of one tile
    const int pixg = Co * 4, pixo = Cin * OSZ;
    const unsigned long long gbase = (unsigned long long)(reinterpret_cast<const char*>(g) + (size_t)U.n * Hc * Wc * pixg);
    const unsigned long long obase = (unsigned long long)(reinterpret_cast<char*>(gx) + (size_t)U.n * H * W * pixo);
    const unsigned gvM = (unsigned)((7 * U.tc) * pixg + U.cl * 4), gvL = gvM - (unsigned)pixg, gvR = gvM + 7u * (unsigned)pixg;
    auto load_g = [&](uint32_t (&raw)[10], int m) { row_load10<float>(raw, gvL, gvM, gvR, row_desc(gbase, 7 * U.tr - 1 + m, Hc, Wc * pixg), 0, pixg); };

    uint32_t rg[NM][10];
    sfor<AHEAD>([&](auto mc) { load_g(rg[decltype(mc)::value], decltype(mc)::value); });
    // the 49 taps of this lane's output channel as pairs against a pixel pair's parities: (0, w0) (w1, w2) (w3, w4) (w5, w6) per tap row
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 49 * Co * 4, 0x00020000);
    f32x2 P[7][4];
    {
        int Cc = Co;
        asm volatile("" : "+s"(Cc));
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            float t[7];
#pragma unroll
            for (int v = 0; v < 7; ++v) t[v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrc, U.cl * 4, (u * 7 + v) * Cc * 4, 0));
            P[u][0] = f32x2{0.f, t[0]}; P[u][1] = f32x2{t[1], t[2]}; P[u][2] = f32x2{t[3], t[4]}; P[u][3] = f32x2{t[5], t[6]};
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) { pin(P[u][0]); pin(P[u][1]); pin(P[u][2]); pin(P[u][3]); }
    }
    const int odd = U.c & 1;                                     // this lane stores the odd pixels of a row (its partner the even ones)
    const unsigned yoff = U.live ? (unsigned)((14 * U.tc + odd) * pixo + (U.c >> 1) * OSZ) : 0x80000000u;
    f32x2 acc[7][7];

    sfor<NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m + AHEAD < NM) load_g(rg[m + AHEAD], m + AHEAD);
        pin_row10<S::cap(S::pending(m))>(rg[m]);
        f32x2 gs[10];                                            // each value in the low half of an aligned pair: the packed FMAs broadcast it for free
#pragma unroll
        for (int q = 0; q < 10; ++q) gs[q] = f32x2{__uint_as_float(rg[m][q]), 0.f};
        // coarse row m (coarse row 7 tr - 1 + m) feeds fine row y = 2m + u - 5 with tap row u; rows opened here (first touched): u = 6 -> y = 2m + 1, u = 5 -> y = 2m
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int y = 2 * m + u - 5;
            if (y < 0 || y > 13) continue;
            f32x2(&a)[7] = acc[y % 7];
            const bool opens = (m == 0) || (u >= 5);             // the first coarse row opens every row it touches; later ones the two newest
            if (opens) {
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = splat(gs[j + 3].x) * P[u][0];
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(splat(gs[j + 3].x), P[u][0], a[j]);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(splat(gs[j + 2].x), P[u][1], a[j]);
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(splat(gs[j + 1].x), P[u][2], a[j]);
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(splat(gs[j].x), P[u][3], a[j]);
        }
        // fine rows 2m - 5 (tap row 0 was its last) and 2m - 4 (tap row 1) are complete: add the lane pair's shares, store this lane's parity of the pixels
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int y = 2 * m - 5 + d;
            if (y < 0 || y > 13) continue;
            uint32_t pk[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const f32x2 v = acc[y % 7][j];
                const float mine = odd ? v.y : v.x, other = odd ? v.x : v.y;         // what this lane stores / what its partner stores
                // partner's share of MY pixel: the partner holds it in its `other` slot
                const float theirs = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, other), 0xB1, 0xf, 0xf, true));
                const float sum = mine + theirs;
                if constexpr (std::is_same<TO, float>::value) pk[j] = __float_as_uint(sum);
                else pk[j] = pk16<TO>(sum, sum);
            }
            row_store7_stride2<TO>(pk, yoff, row_desc(obase, 14 * U.tr + y, H, W * pixo), 0, 2 * pixo);
        }
#pragma unroll
        for (int y = 0; y < 14; ++y) if (y >= 2 * m - 3 && y <= 2 * m + 1) pin(acc[y % 7]);
        CPT_FENCE;
    });
}

#if RCX_CPTBWD_PART == 1
// bytes per pixel as a template argument where the row statements have an immediate form for it (64 / 128 channels of a 16-bit type, 64 of float32)
template <typename TG, typename TO, int H, int PG, int PO>
static hipError_t launch_gx2(const void* g, const float* G, void* out, const float* wf, const float* wd, int N, int C, hipStream_t s)
{
    const long long units = (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    if (!G) hipLaunchKernelGGL((k_bwd_gx<TG, TO, H, false, PG, PO>), grid, block, 0, s, (const TG*)g, G, (TO*)out, wf, wd, N, C);
    else hipLaunchKernelGGL((k_bwd_gx<TG, TO, H, true, PG, PO>), grid, block, 0, s, (const TG*)g, G, (TO*)out, wf, wd, N, C);
    return hipGetLastError();
}
template <typename TO, int H>
static hipError_t launch_dT(const float* G, void* out, const float* wd, int N, int C, hipStream_t s)
{
    const long long units = (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14);
    const int po = C * (int)sizeof(TO);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    if (po == 128) hipLaunchKernelGGL((k_bwd_gx<float, TO, H, true, 0, 128, RCX_GX_AHEAD, 2, false>), grid, block, 0, s, (const float*)nullptr, G, (TO*)out, (const float*)nullptr, wd, N, C);
    else if (po == 256) hipLaunchKernelGGL((k_bwd_gx<float, TO, H, true, 0, 256, RCX_GX_AHEAD, 2, false>), grid, block, 0, s, (const float*)nullptr, G, (TO*)out, (const float*)nullptr, wd, N, C);
    else hipLaunchKernelGGL((k_bwd_gx<float, TO, H, true, 0, 0, RCX_GX_AHEAD, 2, false>), grid, block, 0, s, (const float*)nullptr, G, (TO*)out, (const float*)nullptr, wd, N, C);
    return hipGetLastError();
}
template <typename TG, typename TO, int H>
static hipError_t launch_gx(const void* g, const float* G, void* out, const float* wf, const float* wd, int N, int C, hipStream_t s)
{
    const int pg = C * (int)sizeof(TG), po = C * (int)sizeof(TO);
    if (pg == 128 && po == 128) return launch_gx2<TG, TO, H, 128, 128>(g, G, out, wf, wd, N, C, s);
    if (pg == 256 && po == 256) return launch_gx2<TG, TO, H, 256, 256>(g, G, out, wf, wd, N, C, s);
    return launch_gx2<TG, TO, H, 0, 0>(g, G, out, wf, wd, N, C, s);
}

#endif
#if RCX_CPTBWD_PART == 2
template <typename TG, int H, int PG>
static hipError_t launch_gc2(const void* g, float* gC, const float* wf, int N, int C, int mode, hipStream_t s)
{
    const long long units = (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    if (mode == 1) hipLaunchKernelGGL((k_bwd_gc<1, TG, H, PG>), grid, block, 0, s, (const TG*)g, gC, wf, N, C);
    else hipLaunchKernelGGL((k_bwd_gc<0, TG, H, PG>), grid, block, 0, s, (const TG*)g, gC, wf, N, C);
    return hipGetLastError();
}
template <typename TG, int H>
static hipError_t launch_gc(const void* g, float* gC, const float* wf, int N, int C, int mode, hipStream_t s)
{
    const int pg = C * (int)sizeof(TG);
    if (pg == 128) return launch_gc2<TG, H, 128>(g, gC, wf, N, C, mode, s);
    if (pg == 256) return launch_gc2<TG, H, 256>(g, gC, wf, N, C, mode, s);
    return launch_gc2<TG, H, 0>(g, gC, wf, N, C, mode, s);
}

#endif
#if RCX_CPTBWD_PART == 4
template <typename TA, int H, int K, int MULT>
static hipError_t launch_wd(const void* a, const float* G, float* partial, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * (H / 14) * ((C + 63) / 64));
    const int pa = (C / MULT) * (int)sizeof(TA);
    if (pa == 128) hipLaunchKernelGGL((k_wgrad_d<TA, H, 128, K, MULT>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, G, partial, N, C);
    else if (pa == 256) hipLaunchKernelGGL((k_wgrad_d<TA, H, 256, K, MULT>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, G, partial, N, C);
    else hipLaunchKernelGGL((k_wgrad_d<TA, H, 0, K, MULT>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, G, partial, N, C);
    return hipGetLastError();
}
#endif
#if RCX_CPTBWD_PART == 3
template <int MODE, typename TA, typename TG, int H>
static hipError_t launch_wk(const void* a, const float* coarse, const void* g, float* partial, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * (H / 14) * ((C + 63) / 64));
    const int pa = C * (int)sizeof(TA);
    if (pa == 128) hipLaunchKernelGGL((k_wgrad_k<MODE, TA, TG, H, 128>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, coarse, (const TG*)g, partial, N, C);
    else if (pa == 256) hipLaunchKernelGGL((k_wgrad_k<MODE, TA, TG, H, 256>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, coarse, (const TG*)g, partial, N, C);
    else hipLaunchKernelGGL((k_wgrad_k<MODE, TA, TG, H, 0>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, coarse, (const TG*)g, partial, N, C);
    return hipGetLastError();
}
template <int MODE, typename TA, typename TG>
static hipError_t launch_wk_h(const void* a, const float* coarse, const void* g, float* partial, int N, int C, int H, hipStream_t s)
{
    return H == 56 ? launch_wk<MODE, TA, TG, 56>(a, coarse, g, partial, N, C, s) : launch_wk<MODE, TA, TG, 28>(a, coarse, g, partial, N, C, s);
}
#endif
#if RCX_CPTBWD_PART == 5
template <typename TO, int H>
static hipError_t launch_dn(const float* g, void* gx, const float* w, int N, int Cin, hipStream_t s)
{
    const long long units = (long long)N * ((2 * Cin + 63) / 64) * (H / 14) * (H / 14);
    hipLaunchKernelGGL((k_bwd_down7m2<TO, H>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, g, (TO*)gx, w, N, Cin);
    return hipGetLastError();
}
#endif
}  // namespace cptbwd

#if RCX_CPTBWD_PART == 4
// the shared down conv's weight gradient from (a: H x H of a_dt, G: H/2 x H/2 float32): one partial row of 26 C sums per (image, 14-row band of a)
hipError_t bwd_wgrad_d_cpt(const void* a, int a_dt, const float* G, float* partial, int N, int C, int H, hipStream_t s, int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 14);
#define RCX_WD(TA_) (H == 56 ? cptbwd::launch_wd<TA_, 56, 5, 1>(a, G, partial, N, C, s) : cptbwd::launch_wd<TA_, 28, 5, 1>(a, G, partial, N, C, s))
    if (a_dt == 1) return RCX_WD(bf16_t);
    if (a_dt == 2) return RCX_WD(f16_t);
    return RCX_WD(float);
#undef RCX_WD
}

#endif
#if RCX_CPTBWD_PART == 4
// ... and of Downsample's 7 x 7 stride-2 multiplier-2 conv (a: H x H with Cout / 2 channels, G: H/2 x H/2 with Cout): one partial row of 50 Cout sums per (image, band)
bool bwd_wgrad_dm_cpt_applicable(int N, int Cout, int H, int W, int k)
{
    if (rcx::opt::is_zero(rcx::opt::BWD_CPT)) return false;
    return k == 7 && H == W && (H == 56 || H == 28 || H == 14) && Cout >= 2 && Cout % 2 == 0 && N * (H / 14) <= 512;      // 512 partial rows: rcx_dwconv2d_bwd_workspace_bytes
}
hipError_t bwd_wgrad_dm_cpt(const void* a, int a_dt, const float* G, float* partial, int N, int Cout, int H, hipStream_t s, int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 14);
#define RCX_WDM(TA_) (H == 56 ? cptbwd::launch_wd<TA_, 56, 7, 2>(a, G, partial, N, Cout, s) : H == 28 ? cptbwd::launch_wd<TA_, 28, 7, 2>(a, G, partial, N, Cout, s) : cptbwd::launch_wd<TA_, 14, 7, 2>(a, G, partial, N, Cout, s))
    if (a_dt == 1) return RCX_WDM(bf16_t);
    if (a_dt == 2) return RCX_WDM(f16_t);
    return RCX_WDM(float);
#undef RCX_WDM
}
#endif
#if RCX_CPTBWD_PART == 3
// weight gradient of a stride-1 conv over T = a + R(coarse) (coarse == nullptr: T = a) from the gradient g of its output (float32, or a's own 16-bit type):
// one partial row of 26 C sums per (image, 14-row band)
hipError_t bwd_wgrad_k_cpt(const void* a, int a_dt, const float* coarse, const void* g, int g_dt, float* partial, int N, int C, int H, int mode, hipStream_t s,
                           int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 14);
    if (g_dt != 0 && g_dt != a_dt) return hipErrorInvalidValue;
    // float16 a AND g: both rows take a conversion register per element and the kernel no longer fits 256 registers (the compiler then spills row registers
    // whose loads are still in flight: tools/check_asm_hazards.py) -- that combination keeps rcx_cplwgrad.hip's kernel (same partial rows)
    if (a_dt == 2 && g_dt == 2) return wgrad_cpl(a, a_dt, coarse, g, g_dt, partial, N, C, H, mode, s, rows_out);
#define RCX_WK2(MD_, TA_) (g_dt == 0 ? cptbwd::launch_wk_h<MD_, TA_, float>(a, coarse, g, partial, N, C, H, s) : cptbwd::launch_wk_h<MD_, TA_, TA_>(a, coarse, g, partial, N, C, H, s))
#define RCX_WK(MD_) (a_dt == 1 ? RCX_WK2(MD_, bf16_t) : a_dt == 2 ? cptbwd::launch_wk_h<MD_, f16_t, float>(a, coarse, g, partial, N, C, H, s) : cptbwd::launch_wk_h<MD_, float, float>(a, coarse, g, partial, N, C, H, s))
    if (!coarse) return RCX_WK(2);
    return mode == 1 ? RCX_WK(1) : RCX_WK(0);
#undef RCX_WK
#undef RCX_WK2
}

#endif
#if RCX_CPTBWD_PART == 1
bool bwd_cpt_applicable(int N, int C, int H, int W, int k)
{
    if (rcx::opt::is_zero(rcx::opt::BWD_CPT)) return false;
    return k == 5 && H == W && (H == 56 || H == 28) && N >= 1 && C >= 1 && (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14) < (1LL << 31);
}

// out (out_dt) = K^ g + D^T G   (G == nullptr: the conv adjoint alone)
hipError_t bwd_gx_cpt(const void* g, int g_dt, const float* G, void* out, int out_dt, const float* wf, const float* wd, int N, int C, int H, hipStream_t s)
{
#define RCX_GX(TG_, TO_) (H == 56 ? cptbwd::launch_gx<TG_, TO_, 56>(g, G, out, wf, wd, N, C, s) : cptbwd::launch_gx<TG_, TO_, 28>(g, G, out, wf, wd, N, C, s))
    if (g_dt == 0 && out_dt == 0) return RCX_GX(float, float);
    if (g_dt == 1 && out_dt == 1) return RCX_GX(bf16_t, bf16_t);
    if (g_dt == 2 && out_dt == 2) return RCX_GX(f16_t, f16_t);
    if (g_dt == 0 && out_dt == 1) return RCX_GX(float, bf16_t);
    if (g_dt == 0 && out_dt == 2) return RCX_GX(float, f16_t);
#undef RCX_GX
    return hipErrorInvalidValue;
}

#endif
#if RCX_CPTBWD_PART == 1
// out (out_dt) = D^T G: the input gradient of a plain stride-2 conv5 (G: H/2 x H/2 float32)
hipError_t bwd_dT_cpt(const float* G, void* out, int out_dt, const float* wd, int N, int C, int H, hipStream_t s)
{
#define RCX_DT(TO_) (H == 56 ? cptbwd::launch_dT<TO_, 56>(G, out, wd, N, C, s) : cptbwd::launch_dT<TO_, 28>(G, out, wd, N, C, s))
    if (out_dt == 1) return RCX_DT(bf16_t);
    if (out_dt == 2) return RCX_DT(f16_t);
    return RCX_DT(float);
#undef RCX_DT
}
#endif
#if RCX_CPTBWD_PART == 5
// input gradient of the 7 x 7 stride-2 multiplier-2 conv (Downsample) on the 56 x 56 / 28 x 28 / 14 x 14 input planes
bool bwd_down7m2_cpt_applicable(int N, int Cin, int H, int W, int k)
{
    if (rcx::opt::is_zero(rcx::opt::BWD_CPT)) return false;
    return k == 7 && H == W && (H == 56 || H == 28 || H == 14) && N >= 1 && Cin >= 1 && (long long)N * ((2 * Cin + 63) / 64) * (H / 14) * (H / 14) < (1LL << 31);
}
hipError_t bwd_down7m2_cpt(const float* g, void* gx, int x_dt, const float* w, int N, int Cin, int H, hipStream_t s)
{
#define RCX_DN(TO_) (H == 56 ? cptbwd::launch_dn<TO_, 56>(g, gx, w, N, Cin, s) : H == 28 ? cptbwd::launch_dn<TO_, 28>(g, gx, w, N, Cin, s) : cptbwd::launch_dn<TO_, 14>(g, gx, w, N, Cin, s))
    if (x_dt == 1) return RCX_DN(bf16_t);
    if (x_dt == 2) return RCX_DN(f16_t);
    return RCX_DN(float);
#undef RCX_DN
}
#endif
#if RCX_CPTBWD_PART == 2
// gC (float32, H/2 x H/2) = R^T (K^ g)
hipError_t bwd_gc_cpt(const void* g, int g_dt, float* gC, const float* wf, int N, int C, int H, int mode, hipStream_t s)
{
#define RCX_GC(TG_) (H == 56 ? cptbwd::launch_gc<TG_, 56>(g, gC, wf, N, C, mode, s) : cptbwd::launch_gc<TG_, 28>(g, gC, wf, N, C, mode, s))
    if (g_dt == 1) return RCX_GC(bf16_t);
    if (g_dt == 2) return RCX_GC(f16_t);
    return RCX_GC(float);
#undef RCX_GC
}
#endif

}  // namespace rcx
