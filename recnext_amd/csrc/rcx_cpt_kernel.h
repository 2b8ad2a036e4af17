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
#pragma once
#include <hip/hip_runtime.h>
#include "rcx_opts.h"
#include <stdlib.h>
#include <type_traits>

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
#ifndef RCX_CPT_AHEAD1
#define RCX_CPT_AHEAD1 3                   /* rows of x in flight in front of the row being used, pass 1 / pass 2 (tools/cpt_one.hip sweeps them) */
#endif
#ifndef RCX_CPT_AHEAD2
#define RCX_CPT_AHEAD2 2
#endif
#ifndef RCX_CPT_STAGGER
#define RCX_CPT_STAGGER 0                  /* A/B (tools/cpt_one.hip): the second half of a workgroup's waves -- the SIMD partners of the first half -- sleep
                                              64 x this many cycles behind the barrier in front of each streaming pass, so that partners do not burst their
                                              loads and their FMAs in lockstep (MI355X_MICROARCH.md, two waves per SIMD, item 9) */
#endif
#ifndef RCX_CPT_PRIO
#define RCX_CPT_PRIO 0                     /* A/B: s_setprio 1 for the second half of the waves (the arbitration losers by age; ibid. item 4) */
#endif
#ifndef RCX_CPT16_ALIAS
#define RCX_CPT16_ALIAS 0                  /* A/B: 16-pixel tiles with the small planes aliased into the level-1 plane (two workgroups per CU); see Geo::ALIAS */
#endif
#ifndef RCX_CPT_STG_P2
#define RCX_CPT_STG_P2 0                   /* staged x rows (STG, diagnostic build) in pass 2 as well as in pass 1 (round 5 measured pass 1 alone: no gain either) */
#endif
#ifndef RCX_CPT_SKIPW
#define RCX_CPT_SKIPW 1                    /* piece rounds of the 7- and 4-wide planes: waves none of whose tile-lanes has a row skip the round (uniform branch) */
#endif
#ifndef RCX_CPT_ENDBAR
#define RCX_CPT_ENDBAR 1                   /* A/B: the barrier at the end of a unit.  The barrier behind the next unit's tap loads already orders this unit's last reads of C1
                                              (pass 2) before the next unit's first writes of F1 (pass 1); measured equal either way (profiles/r05_cpt_unit_timeline.txt) */
#endif
#ifndef RCX_CPT_PF
#define RCX_CPT_PF 0                       /* wide-load L2 prefetch ahead of pass 1: measured slower, see pass 1 */
#endif

// diagnostic build only (-DRCX_STAMPS, tools/cpt_one.hip): lane 0 of every wave of the first workgroups writes the clock at the phase boundaries of
// its first four units (`it` = the unit loop's counter)
#ifdef RCX_STAMPS
static __device__ unsigned long long* g_cpt_stamps = nullptr;
#define CPT_STAMP(id)                                                                                                    \
    do {                                                                                                                 \
        if ((threadIdx.x & 63) == 0 && g_cpt_stamps && blockIdx.x < 512)                                                 \
            g_cpt_stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + (it < 3 ? it : 3)) * 16 + (id)] = __builtin_readcyclecounter();             \
    } while (0)
#define CPT_STAMP_RT(id)                                                                                                 \
    do {                                                                                                                 \
        if ((threadIdx.x & 63) == 0 && g_cpt_stamps && blockIdx.x < 512)                                                 \
            g_cpt_stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + (it < 3 ? it : 3)) * 16 + (id)] = __builtin_amdgcn_s_memrealtime();         \
    } while (0)
#else
#define CPT_STAMP(id) do { } while (0)
#define CPT_STAMP_RT(id) do { } while (0)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4pf __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char* gcptr;
typedef __attribute__((address_space(1))) char* gptr;

__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
// (a[A], b[B]) in one instruction; the compiler spends two v_mov_b32 on a shuffle whose sources come from LDS reads
template <int A, int B> __device__ __forceinline__ f32x2 pkmov(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[%3,%4]" : "=v"(d) : "v"(a), "v"(b), "n"(A), "n"(B));
    return d;
}
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
// A whole row = ONE asm statement: 18 loads (or 14 stores) back to back, no compiler-inserted padding between them.  The scalar
// offset the instructions read is produced by an SALU instruction INSIDE the statement: an SGPR operand handed in from outside
// may have just been written by a VALU instruction (v_readfirstlane_b32, or v_readlane_b32 reloading a spill), and a
// vector-memory instruction that reads an SGPR within 5 wait states of a VALU write to it sees the old value -- the hazard
// recognizer does not look inside inline asm.  s_add_i32 writes SCC: declared, or a scalar select scheduled behind the
// statement reads the wrong condition.  Outputs are early-clobber: a destination must not share a register with an offset that
// a later load of the same statement still reads.
// PIXB = bytes between horizontally adjacent pixels when that is a compile-time constant (columns become immediate offsets:
// 12-bit, so the columns past 4095 bytes go through a second scalar base), 0 = run-time pitch (one scalar add per column).
#define CPT_OUT18(v) "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), \
                     "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15]), "=&v"(v[16]), "=&v"(v[17])
// immediate columns: L = columns -2, -1 (voffL), M k = column k (voffM), R = columns 14, 15 (voffR)
#define CPT_LI(OP, d, V, S, k) OP " %" #d ", %[" V "], %[rs], %[" S "] offen offset:%[pb]*" #k "\n\t"
#define CPT_ROW_IMM(OP)                                                                                                              \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                  \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1)                                                                          \
    CPT_LI(OP, 2, "vm", "t", 0) CPT_LI(OP, 3, "vm", "t", 1) CPT_LI(OP, 4, "vm", "t", 2) CPT_LI(OP, 5, "vm", "t", 3)                  \
    CPT_LI(OP, 6, "vm", "t", 4) CPT_LI(OP, 7, "vm", "t", 5) CPT_LI(OP, 8, "vm", "t", 6) CPT_LI(OP, 9, "vm", "t", 7)                  \
    CPT_LI(OP, 10, "vm", "t", 8) CPT_LI(OP, 11, "vm", "t", 9) CPT_LI(OP, 12, "vm", "t", 10) CPT_LI(OP, 13, "vm", "t", 11)            \
    CPT_LI(OP, 14, "vm", "t", 12) CPT_LI(OP, 15, "vm", "t", 13)                                                                      \
    CPT_LI(OP, 16, "vr", "t", 0) CPT_LI(OP, 17, "vr", "t", 1)
#define CPT_ROW_BIG(OP)                                                                                                              \
    "s_add_i32 %[t], %[rb], 0\n\ts_add_i32 %[t2], %[rb], %[pb]*7\n\t"                                                               \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1)                                                                          \
    CPT_LI(OP, 2, "vm", "t", 0) CPT_LI(OP, 3, "vm", "t", 1) CPT_LI(OP, 4, "vm", "t", 2) CPT_LI(OP, 5, "vm", "t", 3)                  \
    CPT_LI(OP, 6, "vm", "t", 4) CPT_LI(OP, 7, "vm", "t", 5) CPT_LI(OP, 8, "vm", "t", 6) CPT_LI(OP, 9, "vm", "t2", 0)                 \
    CPT_LI(OP, 10, "vm", "t2", 1) CPT_LI(OP, 11, "vm", "t2", 2) CPT_LI(OP, 12, "vm", "t2", 3) CPT_LI(OP, 13, "vm", "t2", 4)          \
    CPT_LI(OP, 14, "vm", "t2", 5) CPT_LI(OP, 15, "vm", "t2", 6)                                                                      \
    CPT_LI(OP, 16, "vr", "t", 0) CPT_LI(OP, 17, "vr", "t", 1)
// run-time pitch: t2 walks along the row
#define CPT_LG(OP, d, V, S) OP " %" #d ", %[" V "], %[rs], %[" S "] offen\n\t"
#define CPT_LGN(OP, d) "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_LG(OP, d, "vm", "t2")
#define CPT_ROW_GEN(OP)                                                                                                              \
    "s_add_i32 %[t], %[rb], 0\n\ts_add_i32 %[t2], %[rb], %[pix]\n\t"                                                                \
    CPT_LG(OP, 0, "vl", "t") CPT_LG(OP, 1, "vl", "t2") CPT_LG(OP, 16, "vr", "t") CPT_LG(OP, 17, "vr", "t2")                          \
    CPT_LG(OP, 2, "vm", "t") CPT_LG(OP, 3, "vm", "t2")                                                                               \
    CPT_LGN(OP, 4) CPT_LGN(OP, 5) CPT_LGN(OP, 6) CPT_LGN(OP, 7) CPT_LGN(OP, 8) CPT_LGN(OP, 9) CPT_LGN(OP, 10) CPT_LGN(OP, 11)         \
    CPT_LGN(OP, 12) CPT_LGN(OP, 13) CPT_LGN(OP, 14) CPT_LGN(OP, 15)

template <typename TIO> struct IoOp;
// bf16 -> float32 without an instruction: the D16 "hi" load fills the upper half and zeroes the lower (tools/ubench/d16_probe.hip)
#define CPT_LD16 "buffer_load_short_d16_hi"
#define CPT_LDH "buffer_load_ushort"                  /* float16: zero-extended, then one v_cvt_f32_f16 per element */
#define CPT_LD32 "buffer_load_dword"

template <typename TIO, int PIXB>
__device__ __forceinline__ void row_load(uint32_t (&v)[18], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 13 <= 4095) {
        (void)pix; (void)t2;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPT_ROW_IMM(CPT_LDH) : CPT_OUT18(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPT_ROW_IMM(CPT_LD16) : CPT_OUT18(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(CPT_ROW_IMM(CPT_LD32) : CPT_OUT18(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else if constexpr (PIXB > 0) {
        static_assert(PIXB * 6 <= 4095, "pixel pitch too large for two immediate ranges");
        (void)pix;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPT_ROW_BIG(CPT_LDH) : CPT_OUT18(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPT_ROW_BIG(CPT_LD16) : CPT_OUT18(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(CPT_ROW_BIG(CPT_LD32) : CPT_OUT18(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else {
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPT_ROW_GEN(CPT_LDH) : CPT_OUT18(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPT_ROW_GEN(CPT_LD16) : CPT_OUT18(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else
            asm volatile(CPT_ROW_GEN(CPT_LD32) : CPT_OUT18(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    }
}

// a loaded element as float32: float32 and bf16 (D16-hi load) are already there, float16 takes one conversion
template <typename TIO> __device__ __forceinline__ float raw_f32(uint32_t r)
{
    if constexpr (std::is_same<TIO, f16_t>::value) return (float)__builtin_bit_cast(_Float16, (uint16_t)r);
    else return __uint_as_float(r);
}

// ---- one output row of the tile: 14 stores in one statement.  vo = this lane's offset, or out of range (the store is dropped:
// the lanes of a ragged last channel block).  bf16: 7 registers of two converted pixels each, low half = even column.
#define CPT_SI(OP, d, S, k) OP " %[p" #d "], %[vo], %[rs], %[" S "] offen offset:%[pb]*" #k "\n\t"
#define CPT_SG(OP, d, S) OP " %[p" #d "], %[vo], %[rs], %[" S "] offen\n\t"
#define CPT_SGN(OP, d) "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_SG(OP, d, "t2")
#define CPT_ST16_IMM(S0, S1, k0)                                                                                                     \
    CPT_SI("buffer_store_short", 0, S0, 0) CPT_SI("buffer_store_short_d16_hi", 0, S0, 1) CPT_SI("buffer_store_short", 1, S0, 2)      \
    CPT_SI("buffer_store_short_d16_hi", 1, S0, 3) CPT_SI("buffer_store_short", 2, S0, 4) CPT_SI("buffer_store_short_d16_hi", 2, S0, 5) \
    CPT_SI("buffer_store_short", 3, S0, 6)
template <typename TIO, int PIXB> struct RowSt;
template <typename T16, int PIXB> struct RowSt16 {
    static __device__ __forceinline__ void st(const f32x2 (&a)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        uint32_t p[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {                         // one conversion for the two pixels (RNE, NaN stays NaN)
            if constexpr (std::is_same<T16, f16_t>::value) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a[j].x), "v"(a[j].y));
            else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a[j].x), "v"(a[j].y));
        }
        st_packed(p, vo, rs, rb, pix);
    }
    // the row already converted: seven registers of two pixels each
    static __device__ __forceinline__ void st_packed(const uint32_t (&p)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        int t, t2;
        if constexpr (PIXB > 0 && PIXB * 13 <= 4095) {
            (void)pix; (void)t2;
            asm volatile("s_add_i32 %[t], %[rb], 0\n\t"
                         CPT_SI("buffer_store_short", 0, "t", 0) CPT_SI("buffer_store_short_d16_hi", 0, "t", 1)
                         CPT_SI("buffer_store_short", 1, "t", 2) CPT_SI("buffer_store_short_d16_hi", 1, "t", 3)
                         CPT_SI("buffer_store_short", 2, "t", 4) CPT_SI("buffer_store_short_d16_hi", 2, "t", 5)
                         CPT_SI("buffer_store_short", 3, "t", 6) CPT_SI("buffer_store_short_d16_hi", 3, "t", 7)
                         CPT_SI("buffer_store_short", 4, "t", 8) CPT_SI("buffer_store_short_d16_hi", 4, "t", 9)
                         CPT_SI("buffer_store_short", 5, "t", 10) CPT_SI("buffer_store_short_d16_hi", 5, "t", 11)
                         CPT_SI("buffer_store_short", 6, "t", 12) CPT_SI("buffer_store_short_d16_hi", 6, "t", 13)
                         : [t] "=&s"(t)
                         : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]),
                           [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc", "memory");
        } else {
            // run-time pitch (and the large compile-time ones): t2 walks along the row
            asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                         CPT_SG("buffer_store_short", 0, "t2") CPT_SGN("buffer_store_short_d16_hi", 0)
                         CPT_SGN("buffer_store_short", 1) CPT_SGN("buffer_store_short_d16_hi", 1)
                         CPT_SGN("buffer_store_short", 2) CPT_SGN("buffer_store_short_d16_hi", 2)
                         CPT_SGN("buffer_store_short", 3) CPT_SGN("buffer_store_short_d16_hi", 3)
                         CPT_SGN("buffer_store_short", 4) CPT_SGN("buffer_store_short_d16_hi", 4)
                         CPT_SGN("buffer_store_short", 5) CPT_SGN("buffer_store_short_d16_hi", 5)
                         CPT_SGN("buffer_store_short", 6) CPT_SGN("buffer_store_short_d16_hi", 6)
                         : [t2] "=&s"(t2)
                         : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]),
                           [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
            (void)t;
        }
    }
};
template <int PIXB> struct RowSt<bf16_t, PIXB> : RowSt16<bf16_t, PIXB> {};
template <int PIXB> struct RowSt<f16_t, PIXB> : RowSt16<f16_t, PIXB> {};
template <int PIXB> struct RowSt<float, PIXB> {
    static __device__ __forceinline__ void st(const f32x2 (&a)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        int t2;
        float p[14];
#pragma unroll
        for (int j = 0; j < 7; ++j) { p[2 * j] = a[j].x; p[2 * j + 1] = a[j].y; }
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_dword", 0, "t2") CPT_SGN("buffer_store_dword", 1) CPT_SGN("buffer_store_dword", 2) CPT_SGN("buffer_store_dword", 3)
                     CPT_SGN("buffer_store_dword", 4) CPT_SGN("buffer_store_dword", 5) CPT_SGN("buffer_store_dword", 6) CPT_SGN("buffer_store_dword", 7)
                     CPT_SGN("buffer_store_dword", 8) CPT_SGN("buffer_store_dword", 9) CPT_SGN("buffer_store_dword", 10) CPT_SGN("buffer_store_dword", 11)
                     CPT_SGN("buffer_store_dword", 12) CPT_SGN("buffer_store_dword", 13)
                     : [t2] "=&s"(t2)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]),
                       [p8] "v"(p[8]), [p9] "v"(p[9]), [p10] "v"(p[10]), [p11] "v"(p[11]), [p12] "v"(p[12]), [p13] "v"(p[13]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
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

// twenty-register rows (the 16-wide tiles and the 7 x 7 stride-2 conv of rcx_upcpt.hip): the outputs and the first-touch wait
#define CPT_OUT20(v) CPT_OUT18(v), "=&v"(v[18]), "=&v"(v[19])
template <int PENDING>
__device__ __forceinline__ void pin_row20(uint32_t (&v)[20])
{
    asm volatile("s_waitcnt vmcnt(%20)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]),
                 "+v"(v[17]), "+v"(v[18]), "+v"(v[19]) : "n"(PENDING));
}

// ---- 16-pixel tiles (TS = 16: the 16 * 2^k planes of a 512 x 512 input): a row of 20 columns = -2, -1 (vl), 0 .. 15 (vm), 16, 17 (vr), and a row of 16
// outputs; float32 rows as well (the inner block of the split schedule reads the float32 F1 of its outer step)
#define CPT_ROW20_IMM(OP)                                                                                                            \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                  \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1)                                                                          \
    CPT_LI(OP, 2, "vm", "t", 0) CPT_LI(OP, 3, "vm", "t", 1) CPT_LI(OP, 4, "vm", "t", 2) CPT_LI(OP, 5, "vm", "t", 3)                  \
    CPT_LI(OP, 6, "vm", "t", 4) CPT_LI(OP, 7, "vm", "t", 5) CPT_LI(OP, 8, "vm", "t", 6) CPT_LI(OP, 9, "vm", "t", 7)                  \
    CPT_LI(OP, 10, "vm", "t", 8) CPT_LI(OP, 11, "vm", "t", 9) CPT_LI(OP, 12, "vm", "t", 10) CPT_LI(OP, 13, "vm", "t", 11)            \
    CPT_LI(OP, 14, "vm", "t", 12) CPT_LI(OP, 15, "vm", "t", 13) CPT_LI(OP, 16, "vm", "t", 14) CPT_LI(OP, 17, "vm", "t", 15)          \
    CPT_LI(OP, 18, "vr", "t", 0) CPT_LI(OP, 19, "vr", "t", 1)
#define CPT_ROW20_GEN(OP)                                                                                                            \
    "s_add_i32 %[t], %[rb], 0\n\ts_add_i32 %[t2], %[rb], %[pix]\n\t"                                                                \
    CPT_LG(OP, 0, "vl", "t") CPT_LG(OP, 1, "vl", "t2") CPT_LG(OP, 18, "vr", "t") CPT_LG(OP, 19, "vr", "t2")                          \
    CPT_LG(OP, 2, "vm", "t") CPT_LG(OP, 3, "vm", "t2")                                                                               \
    CPT_LGN(OP, 4) CPT_LGN(OP, 5) CPT_LGN(OP, 6) CPT_LGN(OP, 7) CPT_LGN(OP, 8) CPT_LGN(OP, 9) CPT_LGN(OP, 10) CPT_LGN(OP, 11)         \
    CPT_LGN(OP, 12) CPT_LGN(OP, 13) CPT_LGN(OP, 14) CPT_LGN(OP, 15) CPT_LGN(OP, 16) CPT_LGN(OP, 17)
template <typename TIO, int PIXB>
__device__ __forceinline__ void row_load20(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 15 <= 4095) {
        (void)pix; (void)t2;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPT_ROW20_IMM(CPT_LDH) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPT_ROW20_IMM(CPT_LD16) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(CPT_ROW20_IMM(CPT_LD32) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else {
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(CPT_ROW20_GEN(CPT_LDH) : CPT_OUT20(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(CPT_ROW20_GEN(CPT_LD16) : CPT_OUT20(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else
            asm volatile(CPT_ROW20_GEN(CPT_LD32) : CPT_OUT20(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    }
}
// an output row of 16 pixels (run-time pitch: t2 walks along the row)
template <typename TIO>
__device__ __forceinline__ void row_store16(const f32x2 (&a)[8], unsigned vo, i32x4 rs, int rb, int pix)
{
    int t2;
    if constexpr (sizeof(TIO) == 2) {
        uint32_t p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (std::is_same<TIO, f16_t>::value) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a[j].x), "v"(a[j].y));
            else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a[j].x), "v"(a[j].y));
        }
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_short", 0, "t2") CPT_SGN("buffer_store_short_d16_hi", 0) CPT_SGN("buffer_store_short", 1) CPT_SGN("buffer_store_short_d16_hi", 1)
                     CPT_SGN("buffer_store_short", 2) CPT_SGN("buffer_store_short_d16_hi", 2) CPT_SGN("buffer_store_short", 3) CPT_SGN("buffer_store_short_d16_hi", 3)
                     CPT_SGN("buffer_store_short", 4) CPT_SGN("buffer_store_short_d16_hi", 4) CPT_SGN("buffer_store_short", 5) CPT_SGN("buffer_store_short_d16_hi", 5)
                     CPT_SGN("buffer_store_short", 6) CPT_SGN("buffer_store_short_d16_hi", 6) CPT_SGN("buffer_store_short", 7) CPT_SGN("buffer_store_short_d16_hi", 7)
                     : [t2] "=&s"(t2)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    } else {
        float p[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { p[2 * j] = a[j].x; p[2 * j + 1] = a[j].y; }
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_dword", 0, "t2") CPT_SGN("buffer_store_dword", 1) CPT_SGN("buffer_store_dword", 2) CPT_SGN("buffer_store_dword", 3)
                     CPT_SGN("buffer_store_dword", 4) CPT_SGN("buffer_store_dword", 5) CPT_SGN("buffer_store_dword", 6) CPT_SGN("buffer_store_dword", 7)
                     CPT_SGN("buffer_store_dword", 8) CPT_SGN("buffer_store_dword", 9) CPT_SGN("buffer_store_dword", 10) CPT_SGN("buffer_store_dword", 11)
                     CPT_SGN("buffer_store_dword", 12) CPT_SGN("buffer_store_dword", 13) CPT_SGN("buffer_store_dword", 14) CPT_SGN("buffer_store_dword", 15)
                     : [t2] "=&s"(t2)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]),
                       [p8] "v"(p[8]), [p9] "v"(p[9]), [p10] "v"(p[10]), [p11] "v"(p[11]), [p12] "v"(p[12]), [p13] "v"(p[13]), [p14] "v"(p[14]), [p15] "v"(p[15]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    }
}
// the tile-width-generic faces of the row statements
template <typename TIO, int PIXB> __device__ __forceinline__ void trow_load(uint32_t (&v)[18], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix) { row_load<TIO, PIXB>(v, vl, vm, vr, rs, rb, pix); }
template <typename TIO, int PIXB> __device__ __forceinline__ void trow_load(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix) { row_load20<TIO, PIXB>(v, vl, vm, vr, rs, rb, pix); }
template <int PENDING> __device__ __forceinline__ void trow_pin(uint32_t (&v)[18]) { pin_row<PENDING>(v); }
template <int PENDING> __device__ __forceinline__ void trow_pin(uint32_t (&v)[20]) { pin_row20<PENDING>(v); }
template <typename TIO, int PIXB> __device__ __forceinline__ void trow_store(const f32x2 (&a)[7], unsigned vo, i32x4 rs, int rb, int pix) { RowSt<TIO, PIXB>::st(a, vo, rs, rb, pix); }
template <typename TIO, int PIXB> __device__ __forceinline__ void trow_store(const f32x2 (&a)[8], unsigned vo, i32x4 rs, int rb, int pix) { row_store16<TIO>(a, vo, rs, rb, pix); }

// two float32 -> one register of two TIO, low half = the first (RNE, NaN stays NaN): v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 through the compiler
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename TIO> __device__ __forceinline__ uint32_t pk16(float lo, float hi)
{
    if constexpr (std::is_same<TIO, f16_t>::value) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2_t));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t));
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- staged x rows (STG, round 4).  What bounds the passes is the NUMBER of vector-memory wave-instructions: a load costs the CU ~7.3 cycles
// (8.7 at 8 waves per CU) whatever its width up to 4 bytes per lane, 15.8 at 16 bytes (tools/ubench/vmem_rate.hip, profiles/archive/r04_vmem_rate.txt);
// eighteen 2-byte loads per row and lane are 157 cycles of the CU's one texture-address path, three 16-byte LDS-DMA pieces are ~48, and the
// row no longer waits in registers.  A row = the wave's 2 x 18 x 64-byte image (Geo::SLOTB) in LDS order = piece order: piece j, lane i =
// 16-byte chunk 64 j + i; the source address is per lane (an out-of-range offset writes zeros: the padding left and right of the image;
// probed: tools/ubench/ldsdma_probe.hip).  M0 = the LDS byte address of the piece; it is written and restored inside the statement (the
// compiler owns M0), one wait state between an SALU write of M0 and the instruction that reads it, SCC declared.
__device__ __forceinline__ void stage_row(unsigned v0, unsigned v1, unsigned v2, i32x4 rs, int rb, int ldsaddr)
{
    int t, keep;
    unsigned long long ex;
    asm volatile("s_add_i32 %[t], %[rb], 0\n\t"
                 "s_mov_b32 %[keep], m0\n\t"
                 "s_add_i32 m0, %[la], 0\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[v0], %[rs], %[t] offen lds\n\t"
                 "s_add_i32 m0, %[la], 1024\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[v1], %[rs], %[t] offen lds\n\t"
                 "s_add_i32 m0, %[la], 2048\n\t"
                 "s_mov_b64 %[ex], exec\n\ts_mov_b64 exec, 0xffff\n\t"
                 "buffer_load_dwordx4 %[v2], %[rs], %[t] offen lds\n\t"
                 "s_mov_b64 exec, %[ex]\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [t] "=&s"(t), [keep] "=&s"(keep), [ex] "=&s"(ex)
                 : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [rs] "s"(rs), [rb] "s"(rb), [la] "s"(ldsaddr)
                 : "scc", "memory");
}
// The row back as 18 float32-position (bf16) / zero-extended (float16) elements of this lane's channel, exactly what row_load leaves: wait
// until at most PENDING younger memory operations are outstanding, five transposing reads (four pixels of the lane's channel each: lane
// 16 g + 4 q + p supplies the address of pixel 4 m + q, 8-byte chunk p of the group's 16 channels; lane 16 g + i receives channel i), their
// wait, and one shift or mask per element.  Loads and waits in ONE statement: no register is in flight outside it.
template <typename TIO, int PENDING, int OFF>
__device__ __forceinline__ void fetch_row(uint32_t (&raw)[18], unsigned addr)
{
    u32x2 d0, d1, d2, d3, d4;
    asm volatile("s_waitcnt vmcnt(%[n])\n\t"
                 "ds_read_b64_tr_b16 %[d0], %[a] offset:%[o]\n\t"
                 "ds_read_b64_tr_b16 %[d1], %[a] offset:%[o]+256\n\t"
                 "ds_read_b64_tr_b16 %[d2], %[a] offset:%[o]+512\n\t"
                 "ds_read_b64_tr_b16 %[d3], %[a] offset:%[o]+768\n\t"
                 "ds_read_b64_tr_b16 %[d4], %[a] offset:%[o]+1024\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4)
                 : [a] "v"(addr), [n] "n"(PENDING), [o] "n"(OFF) : "memory");
    const u32x2 d[5] = {d0, d1, d2, d3, d4};
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (4 * m + e >= 18) continue;
            const uint32_t v = (e >> 1) ? d[m].y : d[m].x;
            if constexpr (std::is_same<TIO, f16_t>::value) raw[4 * m + e] = (e & 1) ? (v >> 16) : (v & 0xffffu);
            else raw[4 * m + e] = (e & 1) ? (v & 0xffff0000u) : (v << 16);
        }
}

// XCD-aware workgroup order of the tile kernels (rcx_upcpt.hip, rcx_cptbwd_kernels.h).  Workgroups are dealt round-robin over the 8 XCDs, each with its own
// L2: with the natural order the tiles of one (image, channel block) plane -- which read each other's halo rows and columns -- land on eight different L2s
// and every halo row comes from HBM once per tile row (k_upadd_cpt at 32 x 64 x 128 x 128: 1.44 x its bytes).  Remapped, XCD x walks planes x, x + 8, ...
// tile by tile: neighbours in space are neighbours in time on one L2.  b = blockIdx.x, G = workgroups per plane, NP = planes; planes past the last multiple
// of 8 keep the natural order.  (RCX_UPCPT_XCD=0 at build time: natural order, for A/B runs.)
#ifndef RCX_UPCPT_XCD
#define RCX_UPCPT_XCD 1
#endif
__device__ __forceinline__ unsigned xcd_workgroup(unsigned b, unsigned G, unsigned NP)
{
#if RCX_UPCPT_XCD
    const unsigned npf = NP & ~7u;
    if (b >= npf * G) return b;
    const unsigned x = b & 7u, slot = b >> 3;
    return (x + 8u * (slot / G)) * G + slot % G;
#else
    return b;
#endif
}

// the 25 taps of one conv for this lane's channel as three register pairs per tap row: (w0,w1) (w2,w3) (w4,0)
struct Taps {
    f32x2 p[5][3];
    float bias;
    __device__ __forceinline__ float at(int u, int v) const { return (v & 1) ? p[u][v >> 1].y : p[u][v >> 1].x; }
};

// wsrc = the weight pack as a raw buffer: one scalar add and one load per tap, no 64-bit vector address arithmetic
__device__ __forceinline__ void load_taps(Taps& t, __amdgpu_buffer_rsrc_t wsrc, __amdgpu_buffer_rsrc_t bsrc, int conv, int C, int c)
{
    asm volatile("" : "+s"(C));                               // the 25 scalar offsets are recomputed here, not hoisted out of the unit loop and spilled
    const int vow = c * 4, base = conv * 25 * C * 4;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrc, vow, base + (u * 5 + v) * C * 4, 0));
            if (v & 1) t.p[u][v >> 1].y = w;
            else t.p[u][v >> 1].x = w;
        }
        t.p[u][2].y = 0.f;
    }
    t.bias = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bsrc, c * 4, conv * C * 4, 0));    // bsrc has zero records when there is no bias: the load returns 0
}

// training forward (rcx_recconv2d_fwd_train): base != nullptr -> the launch also leaves the float32 pyramid the backward reads, F_l at
// base + f_off[l] and C_l at base + c_off[l] (bytes; each N x P_l x P_l x C), l = 1 .. level
struct SavedPyr {
    float* base;
    unsigned long long f_off[5], c_off[5];
};

constexpr int plane_size(int T, int l, int TS = 14) { return l == 0 ? TS * T : (plane_size(T, l - 1, TS) + 1) / 2; }

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

// NL_ = levels of the block: the full ladder down to 4 x 4 (56 x 56 / level 4, 28 x 28 / level 3: RecNeXt at 224 x 224) or one level less
// (56 x 56 / level 3, 28 x 28 / level 2: the same stages of a 448 x 448 input, and the inner blocks of the nested schedule)
// TS_ = pixels per tile side: 14 (the 7 * 2^k planes of a 224 x 224 input) or 16 (round 5: the 64 x 64 / level 3 block of a 512 x 512 input, 16-channel
// workgroups; every plane of its ladder is even, 64 -> 32 -> 16 -> 8)
template <int T_, int HALVES, int MODE, typename TIO, int NL_ = (T_ == 4 ? 4 : 3), int STG_ = 0, int TS_ = 14>
struct Geo {
    static constexpr int T = T_;
    static constexpr int TS = TS_, TH = TS_ / 2;
    static_assert(TS == 14 || (TS == 16 && T == 4 && HALVES == 4 && NL_ == 3 && STG_ == 0), "16-pixel tiles: 64 x 64 / level 3, 16-channel workgroups");
    static constexpr int NL = NL_;
    static_assert(NL == (T == 4 ? 4 : 3) || NL == (T == 4 ? 3 : 2), "levels");
    static constexpr int NW = T * T / HALVES;
    static constexpr int NT = NW * 64;
    static constexpr int CB = 64 / HALVES;                 // channels of a workgroup's block
    static constexpr int PIXF = CB;                        // floats between two pixels of an LDS plane
    static constexpr int NWORK = T * T;
    static constexpr int P0 = TS * T, P1 = TH * T, P2 = plane_size(T, 2, TS), P3 = plane_size(T, 3, TS), P4 = plane_size(T, 4, TS);
    // LDS, in pixels: zero row | guard | L1 | guard | L2 | L3 | L4
    static constexpr int ZR = P1;
    static constexpr int O1 = ZR + 2;
    // ALIAS (A/B, round 5; off): the planes of levels >= 2 INSIDE the level-1 plane's region.  A lane keeps its F1 tile in registers until T1 is formed, so the
    // LDS copy of F1 is dead once F2 = down(F1) has been read (compute, barrier, write), and T1 overwrites C2 only after every lane has read it (read, barrier,
    // write): 68 KB instead of 88 KB per workgroup = two workgroups per CU with 16-pixel tiles -- at 256 registers instead of 512 per wave, one row less in
    // flight and two more barriers.  Measured slower at every batch size (profiles/r05_cpt16_64x64.txt): 67.3 against 62.5 us at 32 images, 88.5 against
    // 105.5 at 64 (the banded lanes kernel: 63.8)
    static constexpr bool ALIAS = TS == 16 && RCX_CPT16_ALIAS != 0;
    static constexpr int O2 = ALIAS ? O1 : O1 + P1 * P1 + 2;
    static constexpr int O3 = O2 + P2 * P2;
    static constexpr int O4 = O3 + P3 * P3;
    static constexpr int NPIX = ALIAS ? O1 + P1 * P1 + 2 : O4 + (NL >= 4 ? P4 * P4 : 0);
    static_assert(!ALIAS || (NL == 3 && O4 <= O1 + P1 * P1), "aliased small planes must fit the level-1 plane");
    // STG (round 4): the two streaming passes fetch their x rows by LDS-DMA into per-wave slots behind the level-1 plane -- over the planes of
    // the levels below, dead during both passes -- and read them back with the transposing read.  A slot = one row of the wave's window:
    // two sub-images [18 pixels: tile columns -2 .. 15][64 bytes = 32 channels] (the wave's two tiles at T = 4, the two channel halves of its
    // tile at T = 2): 2304 bytes = 144 16-byte pieces = three DMA instructions (the third on 16 lanes).  STG_ = slots per wave (rows in flight + 1).
    static constexpr int NSLOT = STG_;
    static constexpr int SLOTB = 2 * 18 * 64;
    static constexpr int STGOFF = O2 * PIXF * 4;            // byte offset of the slots
    static constexpr int STGEND = STGOFF + NW * NSLOT * SLOTB + 256;      // + what the last transposing read of the last slot reaches past its image
    static constexpr int LDS_BYTES = STG_ && STGEND > NPIX * PIXF * 4 ? STGEND : NPIX * PIXF * 4;
    static_assert(!STG_ || (sizeof(TIO) == 2 && ((T == 4 && HALVES == 2) || (T == 2 && HALVES == 1)) && LDS_BYTES <= 160 * 1024), "staged rows: 16-bit activations, 64-byte sub-images");
};

// T = 2, HALVES = 2 (round 3): a wave = 32 channels x the two tiles of one tile row, a workgroup = two waves = 32 channels of an image with
// 35.7 KB of LDS, four per CU -- for the channel counts that are not multiples of 64 (RecNeXt-M1: 96, M5: 160); it replaces round 2's
// image-pair variant (two half-waves = two images; slower than the banded kernel, profiles/archive/r02c_cpt_img2_variant.txt).
// TRAIN: the training-forward instantiation (saves the pyramid); the inference instantiations carry none of that code.
// HALVES = 4 (T = 4; round 3): a wave = 16 channels x the four tiles of one tile row (the quarters of the wave are the tile columns), a
// workgroup = four waves = 16 channels of an image with 69 KB of LDS -- TWO workgroups per CU, which run different units and so are
// in different phases: one's barrier-bound small-plane phases and load waits fill with the other's passes (a 32-channel workgroup
// alone on its CU serialises ~105 k cycles of phases that each leave most of the CU idle).  Used where cb16() says so.
template <int T, int HALVES, int MODE, int PIXB, typename TIO, bool TRAIN = false, int LV = (T == 4 ? 4 : 3), int STG = 0, int TS = 14>
__global__ __launch_bounds__(T * T / HALVES * 64, ((T == 4 && HALVES == 2) || (TS == 16 && RCX_CPT16_ALIAS == 0)) ? 1 : 2)   // 256 registers, 8 waves per CU (16-pixel tiles:
                                                                                              // one 4-wave workgroup per CU -- 88 KB of LDS -- with up to 512 registers a wave)
void k_recconv_cpt(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                   int N, int C, int has_bias, SavedPyr sv)
{
    using G = Geo<T, HALVES, MODE, TIO, LV, STG, TS>;
    static_assert(LV == (T == 4 ? 4 : 3) || !TRAIN, "the shorter ladder: inference");
    static_assert(TS == 14 || !TRAIN, "16-pixel tiles: inference");
    constexpr int NL = G::NL, PIXF = G::PIXF, NWORK = G::NWORK, P0 = G::P0, P1 = G::P1, P2 = G::P2, P3 = G::P3, P4 = G::P4;
    constexpr int TH = TS / 2;                                     // a tile of the level-1 plane is TH x TH
    constexpr int NCOL = TS + 4;                                   // columns of a level-0 input row held by a lane
    constexpr int ESZ = (int)sizeof(TIO);
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // Persistent workgroups: the grid is at most what the chip holds at once and a workgroup walks over its units (image,
    // channel block) -- no relaunch gap between the rounds, LDS zeroed once.  XCD-aware order: workgroups are dealt round-robin
    // over the 8 XCDs; each XCD gets a contiguous run of units, so the channel blocks of one image (the two halves of its
    // 128-byte lines) pass through the same L2 at about the same time.
    static_assert(HALVES != 4 || (T == 4 && !TRAIN), "quarter-wave tiles: the 56x56 inference kernel");
    static_assert(!(T == 2 && HALVES == 2) || !TRAIN, "two tiles per wave at 28x28: inference");
    constexpr int CHB = G::CB;                                     // channels per block
    const int nb = (C + CHB - 1) / CHB;
    const unsigned total = (unsigned)N * (unsigned)nb, GD = gridDim.x;
#ifdef RCX_CPT_NOXCD
    const bool xcd = false;
#else
    const bool xcd = (total & 7u) == 0 && (GD & 7u) == 0;
#endif
    const int tid = (int)threadIdx.x;
    // the wave index stays in a scalar register; the lane index is recomputed per unit from the execution mask (v_mbcnt: no input register), so
    // the workitem id is dead after the LDS clearing and nothing per-lane is kept live -- or spilled -- across the units
    const int w_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- zero the whole LDS image once (zero row, guards; and every later read is of finite data)
    for (int i = tid; i < G::LDS_BYTES / 16; i += G::NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (STG > 0) __syncthreads();                       // a row fetched by LDS-DMA before the first unit's barrier must not meet another wave's clearing stores
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

    int w = w_all;
    asm volatile("" : "+s"(w));                                   // per unit: what derives from the wave index is recomputed, not hoisted out of the unit loop and spilled
    int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_));                               // per unit: what derives from the lane index is recomputed, not kept live (or spilled) across units
    const int lane = lane_;
    constexpr int WPR = T / HALVES;                              // waves per tile row (a wave holds HALVES tiles of one tile row)
    const int h = HALVES == 4 ? (lane >> 4) : (HALVES == 2 ? (lane >> 5) : 0);
    const int ch = lane & (G::CB - 1);
    const int tr = w / WPR;
    const int tcb = w % WPR;
    const int tc = tcb + WPR * h;                                        // per lane (HALVES >= 2) / uniform
    const int q = tr * T + tc;                                          // this tile-lane's worker id
    const bool ledge = tc == 0, redge = tc == T - 1;
    const int c = cb * CHB + ch;
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;
    const int pix = C * ESZ;                                            // bytes between horizontally adjacent pixels
    // training forward: this lane's element (row, col) of the saved plane at byte offset `off` (P x P pixels per image)
    auto sv_ptr = [&](unsigned long long off, int P, int row, int col) -> float* {
        return reinterpret_cast<float*>(reinterpret_cast<char*>(sv.base) + off) + (((size_t)n * P + row) * P + col) * C + c;
    };
    const bool svon = TRAIN && sv.base != nullptr && cvalid;

    CPT_STAMP(0);
    CPT_STAMP_RT(9);
    float* const L = lds + ch;
    const float* const Lzero = L;
    float* const L1 = L + G::O1 * PIXF;
    float* const L2 = L + G::O2 * PIXF;
    float* const L3 = L + G::O3 * PIXF;
    float* const L4 = L + G::O4 * PIXF;


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
    // the lane's tile column, channel and edge flags for global memory
    const int tcG = tc;
    const int cG = c;
    const bool cvalidG = cvalid;
    const int ccG = cvalidG ? cG : C - 1;
    const bool ledgeG = tcG == 0, redgeG = tcG == T - 1;
    // this lane's three row-load offsets.  They are set here for pass 1 and AGAIN, from an opaque copy of the lane index, in front of pass 2:
    // recomputed, not kept live (or spilled: the training-forward instantiation did) across the chain of small-plane phases in between
    unsigned voffM, voffL, voffR;
    auto set_voffs = [&](int tcg, int ccg) {
        voffM = (unsigned)((TS * tcg) * pix + ccg * ESZ);
        voffL = tcg == 0 ? OOB : voffM - 2u * (unsigned)pix;               // columns -2, -1 of the tile
        voffR = tcg == T - 1 ? OOB : voffM + (unsigned)TS * (unsigned)pix; // columns TS, TS + 1
    };
    set_voffs(tcG, ccG);
    // row r (tile-local, -2 .. 15), all 18 columns; rows outside the image are redirected to a valid row (loaded, not used)
    // trp = the tile row as the PASS sees it: pass 2 hands in an opaque copy, or the row bases of pass 1 (one multiply each, the same rows) are kept live
    // -- spilled, with 16-pixel tiles -- across the chain of small-plane phases for pass 2 to reuse
    int trp = tr;
    auto load_row = [&](uint32_t (&raw)[NCOL], int r) {
        int ar = TS * trp + r;
        ar = ar < 0 ? 0 : (ar > P0 - 1 ? P0 - 1 : ar);
        const int rb = __builtin_amdgcn_readfirstlane(ar * (P0 * pix));     // uniform by construction; the asm below needs it in an SGPR
        trow_load<TIO, PIXB>(raw, voffL, voffM, voffR, rsrc, rb, pix);
    };
    auto row_valid = [&](int r) -> bool { const int ar = TS * tr + r; return ar >= 0 && ar < P0; };   // uniform
    // STG: this lane's three source offsets (piece j, lane i = chunk 64 j + i of the row image: sub-image s = chunk / 72, pixel (chunk % 72) / 4 =
    // tile column - 2, 16-byte quarter chunk % 4 of the sub-image's 32 channels; s = the wave's tile (T = 4) or the channel half (T = 2)), the
    // LDS byte address of the wave's first slot, and this lane's address for the transposing reads
    unsigned dv[3] = {0u, 0u, 0u};
    int stg_lds = 0;
    unsigned stg_tra = 0;
    if constexpr (STG > 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int item = 64 * j + lane, sI = item >= 72 ? 1 : 0, rem = item - 72 * sI, px = rem >> 2, q4 = rem & 3;
            const int col = 14 * (T == 4 ? tcb + WPR * sI : tc) + px - 2;
            const int c0 = cb * CHB + (T == 4 ? 0 : 32 * sI) + 8 * q4;                       // first of the chunk's eight channels
            dv[j] = (col >= 0 && col < P0 && c0 < C && item < 144) ? (unsigned)(col * pix + c0 * ESZ) : OOB;
        }
        const int lbase = G::STGOFF + w * (G::NSLOT * G::SLOTB);
        stg_lds = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(lds) + lbase));
        const int g = lane >> 4;
        stg_tra = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(lds) + lbase + (g >> 1) * (18 * 64) + ((lane >> 2) & 3) * 64 + (g & 1) * 32 + (lane & 3) * 8);
    }
    auto stg_request = [&](auto sc, int r) {                      // request row r (tile-local) into slot sc
        int ar = TS * tr + r;
        ar = ar < 0 ? 0 : (ar > P0 - 1 ? P0 - 1 : ar);
        const int rb = __builtin_amdgcn_readfirstlane(ar * (P0 * pix));
        stage_row(dv[0], dv[1], dv[2], rsrc, rb, stg_lds + decltype(sc)::value * G::SLOTB);
    };
    // pass 1's first rows are requested before the taps, the LDS clearing's tail and the barrier: their HBM latency runs behind those
    constexpr int AHEAD1 = TS == 16 && RCX_CPT16_ALIAS != 0 ? 1 : (RCX_CPT_PF > 0 ? 2 : RCX_CPT_AHEAD1), R01 = -2, NR1 = TS + 3;     // 16-pixel tiles: 20-register rows, one row less in flight

    uint32_t raw1[NR1][NCOL];
    constexpr int SAH = STG > 0 ? STG - 1 : 0;                 // staged rows in flight in front of the row being used
    if constexpr (STG > 0) sfor<SAH>([&](auto rc) { stg_request(IC<decltype(rc)::value % (STG > 0 ? STG : 1)>{}, R01 + decltype(rc)::value); });
    else if constexpr (RCX_CPT_PF == 0) sfor<AHEAD1>([&](auto rc) { load_row(raw1[decltype(rc)::value], R01 + decltype(rc)::value); });
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, (NL + 2) * 25 * C * 4, 0x00020000);
    // no bias: a buffer of zero records, every load returns 0 (no per-lane flag for an exec-masked load kept live -- or spilled -- across the units)
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bpack, 0, has_bias ? (NL + 2) * C * 4 : 0, 0x00020000);
    Taps td;
    load_taps(td, wsrc, bsrc, 0, C, cc);
    __syncthreads();
    CPT_STAMP(1);
    if constexpr (RCX_CPT_PRIO > 0) { if (w >= G::NW / 2) __builtin_amdgcn_s_setprio(RCX_CPT_PRIO); }
    if constexpr (RCX_CPT_STAGGER > 0) { if (w >= G::NW / 2) __builtin_amdgcn_s_sleep(RCX_CPT_STAGGER); }

    // ================= pass 1: F1 tile = down(x), rows -2 .. 14 of the tile, input-row stationary (tap pairs) =================
    float f1[TH][TH];                                        // this lane's F1 tile stays in registers until T1 is formed
    {
        constexpr int AHEAD = AHEAD1, R0 = R01, NR = NR1;
        const f32x2 b0 = f32x2{td.bias, 0.f};
        uint32_t (&raw)[NR][NCOL] = raw1;
        f32x2 facc[3][TH];
        // The per-lane loads move 2 bytes each and a wave holds at most 63 memory operations: 3 rows in flight do not cover the HBM
        // latency (stamps: this pass takes 20 k cycles for 11 k cycles of issue).  Tried (-DRCX_CPT_PF=2): pull each row into L2 first
        // with a few WIDE loads (16 bytes per lane, results discarded) PF rows ahead of the element loads.  Measured slower (pass 1:
        // 26 k cycles, 56x56 launch 111.7 vs 105.3 us): a wave's memory operations complete in issue order, so the element loads
        // queue behind the wide loads' HBM latency instead of overtaking them.  Off by default, kept for A/B builds.
        constexpr int PF = RCX_CPT_PF;
        constexpr int CPP = PIXF * ESZ / 16;                   // 16-byte chunks per pixel of the block
        constexpr int PPI = (64 / HALVES) / CPP;               // pixels per instruction and tile
        constexpr int NPF = PF > 0 ? (NCOL + PPI - 1) / PPI : 0; // instructions per row
        u32x4pf sink = {0u, 0u, 0u, 0u};                       // destination of the wide loads: kept live to the end of the pass
        const int pj = (lane & (64 / HALVES - 1)) / CPP, pchunk = lane & (CPP - 1);
        unsigned pvo[NPF > 0 ? NPF : 1];                       // this lane's offsets inside a row, one per instruction
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            int colp = TS * tc - 2 + i * PPI + pj;             // columns left of the image: re-read column 0; right of it: the next row or out of range
            colp = colp < 0 ? 0 : colp;
            pvo[i] = (unsigned)(colp * pix + (cb * PIXF) * ESZ + pchunk * 16);
        }
        auto prefetch_row = [&](int r) {
            if constexpr (PF > 0) {
                int ar = TS * tr + r;
                ar = ar < 0 ? 0 : (ar > P0 - 1 ? P0 - 1 : ar);
                const int rb = __builtin_amdgcn_readfirstlane(ar * (P0 * pix));
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    int t;
                    asm volatile("s_add_i32 %[t], %[rb], 0\n\tbuffer_load_dwordx4 %[d], %[vo], %[rs], %[t] offen"
                                 : [d] "+v"(sink), [t] "=&s"(t) : [vo] "v"(pvo[i]), [rs] "s"(rsrc), [rb] "s"(rb) : "scc");
                }
            }
        };
        if constexpr (STG == 0) sfor<AHEAD + PF>([&](auto rc) { prefetch_row(R0 + decltype(rc)::value); });
        if constexpr (PF > 0 && STG == 0) sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], R0 + decltype(rc)::value); });
        sfor<NR>([&](auto rc) {
            constexpr int ri = decltype(rc)::value, r = R0 + ri;
            if constexpr (STG > 0) {
                constexpr int SN = STG > 0 ? STG : 1;
                if constexpr (ri + SAH < NR) stg_request(IC<(ri + SAH) % SN>{}, r + SAH);          // its slot held row ri - 1: read and waited for in the last iteration
                constexpr int NYS = 3 * (NR - 1 - ri < SAH ? NR - 1 - ri : SAH);
                fetch_row<TIO, NYS, (ri % SN) * G::SLOTB>(raw[ri], stg_tra);
            } else {
            if constexpr (ri + AHEAD + PF < NR) prefetch_row(r + AHEAD + PF);
            if constexpr (ri + AHEAD < NR) load_row(raw[ri + AHEAD], r + AHEAD);
            // younger memory operations: what the iterations since row ri was requested have issued (wide loads first, then a row)
            constexpr int NY = [] {
                int k = 0;
                for (int j = 1; j <= AHEAD; ++j) k += (ri + j < NR ? NCOL : 0) + (ri + j + PF < NR ? NPF : 0);
                return k > 63 ? 63 : k;
            }();
            trow_pin<NY>(raw[ri]);
            }
            f32x2 xr[NCOL / 2];
#pragma unroll
            for (int k = 0; k < NCOL / 2; ++k) xr[k] = f32x2{raw_f32<TIO>(raw[ri][2 * k]), raw_f32<TIO>(raw[ri][2 * k + 1])};
            const bool rv = row_valid(r);
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                const int u = r - 2 * o + 2;
                if (u < 0 || u > 4) continue;
                f32x2(&a)[TH] = facc[o % 3];
                if (rv) {
                    // u == 0: the first contribution of output row o carries the initial value (bias, 0) as its addend
#pragma unroll
                    for (int i = 0; i < TH; ++i) a[i] = pfma(xr[i], td.p[u][0], u == 0 ? b0 : a[i]);
#pragma unroll
                    for (int i = 0; i < TH; ++i) a[i] = pfma(xr[i + 1], td.p[u][1], a[i]);
#pragma unroll
                    for (int i = 0; i < TH; ++i) a[i].x = fmaf(xr[i + 2].x, td.p[u][2].x, a[i].x);
                } else if (u == 0) {
#pragma unroll
                    for (int i = 0; i < TH; ++i) a[i] = b0;
                }
                if (u == 4) {
                    float* dst = L1 + ((TH * tr + o) * P1 + TH * tc) * PIXF;
#pragma unroll
                    for (int i = 0; i < TH; ++i) {
                        f1[o][i] = a[i].x + a[i].y;
                        dst[i * PIXF] = f1[o][i];
                    }
                    if constexpr (TRAIN) if (svon) {
#pragma unroll
                        for (int i = 0; i < TH; ++i) *sv_ptr(sv.f_off[1], P1, TH * tr + o, TH * tc + i) = f1[o][i];
                    }
                    pin(f1[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < TH; ++o) if (r - 2 * o + 2 >= 0 && r - 2 * o + 2 < 4) pin(facc[o % 3]);
            CPT_FENCE;
        });
        asm volatile("" : "+v"(sink));                         // every wide load has landed by now (the last rows' waits were vmcnt(0))
    }
    CPT_STAMP(2);
    __syncthreads();
    CPT_STAMP(3);

    // ================= chain: the small planes, pieces dealt over the T*T tile-lanes =================
    // conv j of the pack: 0 = down, 1 + (NL - l) = the conv of level l, 1 + NL = the final conv
    constexpr int PL[5] = {P0, P1, P2, P3, P4};
    float* const LP[5] = {nullptr, L1, L2, L3, L4};
    // piece rounds of a P-wide plane: P == 14 or 16 (16 workers): two rounds = the two P / 2-wide column segments, row = q; else full rows,
    // row = q + NWORK * round
    auto for_pieces = [&](auto pc, auto&& f) {
        constexpr int P = decltype(pc)::value;
        if constexpr (P > 8) {
            static_assert(NWORK == 16 && P <= 16 && (P & 1) == 0, "14- and 16-wide piece planes are dealt over 16 workers");
            const bool act = q < P;
            const int row = act ? q : 0;
            f(IC<0>{}, IC<0>{}, IC<P / 2>{}, row, act);
            f(IC<1>{}, IC<P / 2>{}, IC<P / 2>{}, row, act);
        } else {
            constexpr int RNDS = (P + NWORK - 1) / NWORK;
            sfor<RNDS>([&](auto rc) {
                constexpr int rnd = decltype(rc)::value;
                const int rr = q + NWORK * rnd;
                const bool act = rr < P;
                // the wave's smallest row of this round (its first tile-lane's; scalar): a wave with no row at all leaves the round to the others --
                // the LDS pipe and the SIMD partner see half (7-wide planes at T = 4) or a quarter (4-wide) of the instructions
                if constexpr (RCX_CPT_SKIPW != 0 && !TRAIN) {
                    if (tr * T + tcb + NWORK * rnd >= P) return;
                }
                f(rc, IC<0>{}, IC<P>{}, act ? rr : 0, act);
            });
        }
    };
    // down ladder: F_l = down(F_{l-1}), l = 2 .. NL
    sfor<NL - 1>([&](auto lc) {
        constexpr int l = 2 + decltype(lc)::value;
        constexpr int PIN = PL[l - 1], PO = PL[l];
        if constexpr (G::ALIAS && l == 2) {
            // F2 lands where F1 was read from: every piece is computed before the first one is written
            static_assert(PO > 8, "two column segments");
            float o2[2][PO / 2];
            for_pieces(IC<PO>{}, [&](auto rc, auto col0c, auto noutc, int row, bool) {
                down_piece<PIN, decltype(col0c)::value, decltype(noutc)::value, PIXF>(LP[1], Lzero, row, td, o2[decltype(rc)::value]);
            });
            __syncthreads();
            for_pieces(IC<PO>{}, [&](auto rc, auto col0c, auto noutc, int row, bool act) {
                if (act) {
                    float* dst = LP[2] + (row * PO + decltype(col0c)::value) * PIXF;
#pragma unroll
                    for (int i = 0; i < decltype(noutc)::value; ++i) dst[i * PIXF] = o2[decltype(rc)::value][i];
                }
            });
        } else
        for_pieces(IC<PO>{}, [&](auto, auto col0c, auto noutc, int row, bool act) {
            constexpr int COL0 = decltype(col0c)::value, NOUT = decltype(noutc)::value;
            float out[NOUT];
            down_piece<PIN, COL0, NOUT, PIXF>(LP[l - 1], Lzero, row, td, out);
            if (act) {
                float* dst = LP[l] + (row * PO + COL0) * PIXF;
#pragma unroll
                for (int i = 0; i < NOUT; ++i) dst[i * PIXF] = out[i];
                if constexpr (TRAIN) if (svon) {
#pragma unroll
                    for (int i = 0; i < NOUT; ++i) *sv_ptr(sv.f_off[l], PO, row, COL0 + i) = out[i];
                }
            }
        });
        __syncthreads();
    });
    CPT_STAMP(4);
    // up recursion on the piece planes: l = NL .. 2: T_l = F_l + resize(C_{l+1}) in place (l < NL), C_l = conv(T_l) in place.
    // (Requesting a level's taps one level ahead was measured: no gain -- the small planes are issue-bound -- and 20 VGPRs.)
    sfor<NL - 1>([&](auto lc) {
        constexpr int l = NL - decltype(lc)::value;
        constexpr int P = PL[l];
        Taps tc_;
        load_taps(tc_, wsrc, bsrc, 1 + (NL - l), C, cc);
        if constexpr (l < NL) {
            constexpr int PC = PL[l + 1];
            for_pieces(IC<P>{}, [&](auto, auto col0c, auto noutc, int row, bool act) {
                tform_piece<MODE, PC, P, decltype(col0c)::value, decltype(noutc)::value, PIXF>(LP[l], LP[l + 1], row, act);
            });
            __syncthreads();
        }
        constexpr int RN = P > 8 ? 2 : (P + NWORK - 1) / NWORK;
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
                if constexpr (TRAIN) if (svon) {
#pragma unroll
                    for (int i = 0; i < NOUT; ++i)
                        *sv_ptr(sv.c_off[l], P, row, COL0 + i) = (i & 1) ? res[decltype(rc)::value][i >> 1].y : res[decltype(rc)::value][i >> 1].x;
                }
            }
        });
        __syncthreads();
    });

    CPT_STAMP(5);
    // ================= level 1, per tile: T1 = F1 + resize(C2) (exact 2x), C1 = conv(T1) =================
    Taps t1;
    load_taps(t1, wsrc, bsrc, NL, C, cc);         // conv of level 1 = pack 1 + (NL - 1)
    {
        // columns: run of 7 starting at absolute column 7*tc (parity uniform), source columns b .. b+4 of C2, clamped
        const int d0 = TH * tc;
        const int bcol = MODE == 1 ? (d0 >> 1) : ((d0 - 1) >> 1);
        constexpr int NV = TH / 2 + 2;                           // source columns (and rows) of C2 a run of TH interpolates from
        int cofs[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            int cx = bcol + k;
            cx = cx < 0 ? 0 : (cx > P2 - 1 ? P2 - 1 : cx);
            cofs[k] = cx * PIXF;
        }
        // HALVES = 4: the four tile columns of a wave have both parities -- the same sums with per-lane weights and selected neighbours
        auto form_lane = [&]() {
            const bool par = (d0 & 1) != 0;
            const float le = MODE == 1 ? 0.f : (par ? 0.25f : 0.75f), lo = MODE == 1 ? 0.f : (par ? 0.75f : 0.25f);
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int dr = TH * tr + r;                     // uniform
                int i0, i1;
                float lam;
                if (MODE == 1) { i0 = i1 = dr >> 1; lam = 0.f; }
                else if (dr & 1) { i0 = (dr - 1) >> 1; i1 = i0 + 1; lam = 0.25f; }
                else { i0 = (dr >> 1) - 1; i1 = i0 + 1; lam = 0.75f; }
                i0 = i0 < 0 ? 0 : (i0 > P2 - 1 ? P2 - 1 : i0);
                i1 = i1 < 0 ? 0 : (i1 > P2 - 1 ? P2 - 1 : i1);
                const float* r0 = L2 + i0 * (P2 * PIXF);
                const float* r1 = L2 + i1 * (P2 * PIXF);
                float V[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) V[k] = MODE == 1 ? r0[cofs[k]] : fmaf(lam, r1[cofs[k]], (1.f - lam) * r0[cofs[k]]);
#pragma unroll
                for (int cI = 0; cI < TH; ++cI) {
                    float up;
                    if (cI & 1) {                                // rel2: par 0 -> (m, m+1; 0.25), par 1 -> (m-1, m; 0.75), m = (cI + 1) / 2
                        const int m = (cI + 1) / 2;
                        const float a0 = par ? V[m - 1] : V[m], a1 = par ? V[m] : V[m + 1];
                        up = MODE == 1 ? (par ? V[m] : V[m - 1]) : fmaf(lo, a1, (1.f - lo) * a0);
                    } else {                                     // rel2: (m, m+1), weight 0.75 (par 0) / 0.25 (par 1), m = cI / 2
                        const int m = cI / 2;
                        up = MODE == 1 ? V[m] : fmaf(le, V[m + 1], (1.f - le) * V[m]);
                    }
                    f1[r][cI] += up;
                }
            }
        };
        constexpr bool LANE_PARITY = HALVES > 1 && WPR == 1 && (TH & 1) != 0;      // the tiles of a wave are neighbours: both column parities in one wave (odd tile width)
        const int cpar = LANE_PARITY ? 0 : __builtin_amdgcn_readfirstlane(d0 & 1);
        auto form = [&](auto parc, auto rparc) {
            constexpr int PAR = decltype(parc)::value, RPAR = decltype(rparc)::value;      // parities of the tile's first column / first row (TH tc, TH tr)
            // the tile's TH rows interpolate from the NV C2 rows ib .. ib + NV - 1 (clamped), ib = (TH tr - 1) >> 1: NV rows of NV values, read once and
            // all of them before the first use (one LDS latency instead of one per row)
            const int ib = (TH * tr - 1) >> 1;                 // uniform
            float C2v[NV][NV];
#pragma unroll
            for (int m = 0; m < NV; ++m) {
                int im = ib + m;
                im = im < 0 ? 0 : (im > P2 - 1 ? P2 - 1 : im);
                const float* rm_ = L2 + im * (P2 * PIXF);
#pragma unroll
                for (int k = 0; k < NV; ++k) C2v[m][k] = rm_[cofs[k]];
            }
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                // row 7 tr + r: odd -> C2 rows ((d - 1)/2, (d + 1)/2), weight 0.25; even -> (d/2 - 1, d/2), weight 0.75; nearest: d >> 1 -- relative to ib
                const bool odd = ((RPAR + r) & 1) != 0;
                const int m0 = MODE == 1 ? (RPAR ? (r + 1) >> 1 : 1 + (r >> 1)) : (RPAR ? (odd ? r / 2 : (r - 1) / 2) : (odd ? (r + 1) / 2 : r / 2));
                const int m1 = MODE == 1 ? m0 : m0 + 1;
                const float lam = MODE == 1 ? 0.f : (odd ? 0.25f : 0.75f);
                float V[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) V[k] = MODE == 1 ? C2v[m0][k] : fmaf(lam, C2v[m1][k], (1.f - lam) * C2v[m0][k]);
                sfor<TH>([&](auto cic) {
                    constexpr int cI = decltype(cic)::value;
                    constexpr Rel rl = rel2(MODE, PAR, cI);
                    const float up = MODE == 1 ? V[rl.idx] : fmaf(rl.l, V[rl.idx + 1], (1.f - rl.l) * V[rl.idx]);
                    f1[r][cI] += up;
                });
            }
        };
        const int rpar = __builtin_amdgcn_readfirstlane((TH * tr) & 1);
        if constexpr (LANE_PARITY) form_lane();
        else if (cpar) { if (rpar) form(IC<1>{}, IC<1>{}); else form(IC<1>{}, IC<0>{}); }
        else { if (rpar) form(IC<0>{}, IC<1>{}); else form(IC<0>{}, IC<0>{}); }
        if constexpr (G::ALIAS) __syncthreads();             // C2 lives inside the level-1 plane: every lane has read it (form() reads it up front) before T1 overwrites it
        float* dst = L1 + ((TH * tr) * P1 + TH * tc) * PIXF;
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
            for (int cI = 0; cI < TH; ++cI) dst[(r * P1 + cI) * PIXF] = f1[r][cI];
    }
    __syncthreads();
    CPT_STAMP(6);
    // halo masks of the tile (per lane): columns outside the plane contribute nothing
    const float lmask = ledge ? 0.f : 1.f, rmask = redge ? 0.f : 1.f;
    {
        // C1 tile, input-row stationary over T1 rows -2 .. TH + 1, columns -2 .. TH + 1 (the guards before and after the plane make every
        // address valid; what a masked column reads is finite)
        f32x2 c1[TH][4];
        const f32x2 b1 = splat(t1.bias);
        const float* base = L1 + ((TH * tr) * P1 + TH * tc) * PIXF;
#pragma unroll
        for (int t = -2; t <= TH + 1; ++t) {
            const int ar = TH * tr + t;
            if (ar >= 0 && ar < P1) {                        // uniform
                const float* rp = base + t * (P1 * PIXF);
                f32x2 in[6], odd[5];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    in[k].x = rp[(2 * k - 2) * PIXF];
                    in[k].y = 2 * k - 1 <= TH + 1 ? rp[(2 * k - 1) * PIXF] : 0.f;
                }
                in[0] = in[0] * splat(lmask);
                if constexpr (TH & 1) { in[(TH + 2) / 2].y *= rmask; in[(TH + 3) / 2].x *= rmask; }     // columns TH, TH + 1: (4, y), (5, x) at TH = 7
                else in[(TH + 2) / 2] = in[(TH + 2) / 2] * splat(rmask);                                 // one pair at an even tile width
#pragma unroll
                for (int j = 0; j < 5; ++j) odd[j] = pkmov<1, 0>(in[j], in[j + 1]);       // one v_pk_mov_b32 (the compiler: two v_mov_b32 on LDS data)
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int o = t - u + 2;
                    if (o < 0 || o > TH - 1) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(in[j], splat(t1.at(u, 0)), u == 0 ? b1 : c1[o][j]);     // u == 0: first contribution of row o
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(odd[j], splat(t1.at(u, 1)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(in[j + 1], splat(t1.at(u, 2)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(odd[j + 1], splat(t1.at(u, 3)), c1[o][j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c1[o][j] = pfma(in[j + 2], splat(t1.at(u, 4)), c1[o][j]);
                }
            } else if (t + 2 <= TH - 1) {                    // a row above the plane: the output row it would have opened starts from the bias
#pragma unroll
                for (int j = 0; j < 4; ++j) c1[t + 2][j] = b1;
            }
            CPT_FENCE;
        }
        __syncthreads();                                     // every read of T1 is done
        float* dst = L1 + ((TH * tr) * P1 + TH * tc) * PIXF;
#pragma unroll
        for (int o = 0; o < TH; ++o)
#pragma unroll
            for (int cI = 0; cI < TH; ++cI) dst[(o * P1 + cI) * PIXF] = (cI & 1) ? c1[o][cI >> 1].y : c1[o][cI >> 1].x;
        if constexpr (TRAIN) if (svon) {
#pragma unroll
            for (int o = 0; o < TH; ++o)
#pragma unroll
                for (int cI = 0; cI < TH; ++cI) *sv_ptr(sv.c_off[1], P1, TH * tr + o, TH * tc + cI) = (cI & 1) ? c1[o][cI >> 1].y : c1[o][cI >> 1].x;
        }
    }
    Taps tf;
    load_taps(tf, wsrc, bsrc, 1 + NL, C, cc);
    __syncthreads();
    CPT_STAMP(7);
    if constexpr (RCX_CPT_STAGGER > 0) { if (w >= G::NW / 2) __builtin_amdgcn_s_sleep(RCX_CPT_STAGGER); }

    // ================= pass 2: y tile = conv(x + resize(C1)), input rows -2 .. 15, five accumulator rows in flight =================
    {
        constexpr int AHEAD = (std::is_same<TIO, f16_t>::value || (TS == 16 && RCX_CPT16_ALIAS != 0)) && RCX_CPT_AHEAD2 > 1 ? 1 : RCX_CPT_AHEAD2, R0 = -2, NR = TS + 4;   // float16: one row less in flight (its
                                                                                  // per-element conversions otherwise spill eight registers at 256)
        {                                                     // see set_voffs
            int l2 = lane;
            asm volatile("" : "+v"(l2));
            const int h2 = HALVES == 4 ? (l2 >> 4) : (HALVES == 2 ? (l2 >> 5) : 0), c2 = cb * CHB + (l2 & (G::CB - 1));
            set_voffs(tcb + WPR * h2, c2 < C ? c2 : C - 1);
            if constexpr (TS == 16) asm volatile("" : "+s"(trp));
        }
        // C1 columns -2 .. 8 of the tile: the two on each side may lie outside the plane (clamped: ATen's border rule)
        const int cb0 = TH * tcG;
        const int cL0 = (ledgeG ? 0 : cb0 - 2) * PIXF, cL1 = (ledgeG ? 0 : cb0 - 1) * PIXF;
        const int cR0 = (redgeG ? P1 - 1 : cb0 + TH) * PIXF, cR1 = (redgeG ? P1 - 1 : cb0 + TH + 1) * PIXF;
        const float lmaskG = ledgeG ? 0.f : 1.f, rmaskG = redgeG ? 0.f : 1.f;
        const float* const L1G = L + G::O1 * PIXF;
        // horizontal weights; the pairs that lie outside the image (columns -2, -1 at the left edge, 14, 15 at the right) are zeroed here
        const f32x2 wq = MODE == 1 ? splat(0.f) : splat(0.25f), wt = MODE == 1 ? splat(1.f) : splat(0.75f);
        uint32_t raw[NR][NCOL];
        constexpr int NHP = NCOL / 2;                            // pairs of columns of the window
        f32x2 H[2][NHP];
        f32x2 acc[5][TH];
        const f32x2 bf = splat(tf.bias);
        i32x4 ysrc;                                           // y image as a raw buffer; lanes past the last channel store out of range (dropped)
        {
            const unsigned long long a = (unsigned long long)(reinterpret_cast<char*>(y) + (size_t)n * P0 * P0 * pix);
            ysrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            ysrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
            ysrc.z = P0 * P0 * pix;
            ysrc.w = 0x00020000;
        }
        const unsigned yoff = cvalidG ? voffM : OOB;          // a valid lane's own column 0 = where its row loads start
        // H[i]: C1 row i (tile-local, -2 .. 8; clamped into the plane) resized horizontally to the 18 columns -2 .. 15
        auto build_H = [&](f32x2 (&Hs)[NHP], int i) {
            int ar = TH * tr + i;
            ar = ar < 0 ? 0 : (ar > P1 - 1 ? P1 - 1 : ar);
            const float* rp = L1G + ar * (P1 * PIXF);
            float cv[TH + 4];
            cv[0] = rp[cL0];
            cv[1] = rp[cL1];
#pragma unroll
            for (int k = 0; k < TH; ++k) cv[2 + k] = rp[(cb0 + k) * PIXF];
            cv[TH + 2] = rp[cR0];
            cv[TH + 3] = rp[cR1];
            f32x2 P[6], Pq[6];                                   // pairs of C1 pixels, and the same times the outer weight (0.25)
#pragma unroll
            for (int m = 0; m < 6; ++m) { P[m] = f32x2{cv[2 * m], 2 * m + 1 < TH + 4 ? cv[2 * m + 1] : 0.f}; Pq[m] = P[m] * wq; }
            sfor<NHP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // columns 2j-2 (even) and 2j-1 (odd): 0.25 c[j] + 0.75 c[j+1] and 0.75 c[j+1] + 0.25 c[j+2]; nearest: c[j+1] twice.
                // (c[j], c[j+2]) = the same halves of two neighbouring pairs: one v_pk_mov_b32
                const f32x2 e = pkmov<(j & 1), (j & 1)>(Pq[j >> 1], Pq[(j >> 1) + 1]);
                const float mid = ((j + 1) & 1) ? P[(j + 1) >> 1].y : P[(j + 1) >> 1].x;
                Hs[j] = pfma(splat(mid), wt, e);
            });
            Hs[0] = Hs[0] * splat(lmaskG);
            Hs[NHP - 1] = Hs[NHP - 1] * splat(rmaskG);
        };
        constexpr bool STG2 = STG > 0 && RCX_CPT_STG_P2 != 0;
        if constexpr (STG2) sfor<SAH>([&](auto rc) { stg_request(IC<decltype(rc)::value % (STG > 0 ? STG : 1)>{}, R0 + decltype(rc)::value); });
        else sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], R0 + decltype(rc)::value); });
        build_H(H[0], -2);
        build_H(H[1], -1);
        sfor<NR>([&](auto rc) {
            constexpr int ri = decltype(rc)::value, t = R0 + ri;
            if constexpr (STG2) {
                if constexpr (ri + SAH < NR) stg_request(IC<(ri + SAH) % (STG > 0 ? STG : 1)>{}, t + SAH);
            } else if constexpr (ri + AHEAD < NR) load_row(raw[ri + AHEAD], t + AHEAD);
            // vertical source rows (tile origin is even): t even -> (t/2 - 1, t/2) weight 0.75; t odd -> ((t-1)/2, (t+1)/2) weight 0.25
            constexpr int te = (t + 2) & 1;                  // parity of t (t + 2 >= 0)
            constexpr int i0 = MODE == 1 ? ((t + 2) >> 1) - 1 : (te ? (t - 1) / 2 : t / 2 - 1);
            constexpr int i1 = MODE == 1 ? i0 : i0 + 1;
            constexpr float lam = MODE == 1 ? 0.f : (te ? 0.25f : 0.75f);
            // H[i1] is first needed here when t is odd (H[-2], H[-1] were built up front)
            if constexpr (MODE == 0 && te && t >= -1) build_H(H[(i1 + 2) & 1], i1);
            if constexpr (MODE == 1 && !te && t >= 0) build_H(H[(i0 + 2) & 1], i0);
            // younger memory operations at this point: the rows requested since (18 loads each) and the output rows stored at the
            // end of the iterations in between (14 stores each; iteration i stores a row for 4 <= i <= 17); the counter holds 63
            constexpr int NLD = NR - 1 - ri < AHEAD ? NR - 1 - ri : AHEAD;
            constexpr int NST = [] { int k = 0; for (int j = 1; j <= AHEAD; ++j) k += (ri - j >= 4 && ri - j <= TS + 3) ? 1 : 0; return k; }();
            if constexpr (STG2) {
                // younger: the pieces of the rows requested since (3 each) and the output rows stored at the end of the iterations in between
                constexpr int SLD = NR - 1 - ri < SAH ? NR - 1 - ri : SAH;
                constexpr int SST = [] { int k = 0; for (int j = 1; j <= SAH; ++j) k += (ri - j >= 4 && ri - j <= 17) ? 1 : 0; return k; }();
                constexpr int SNY = 3 * SLD + (sizeof(TIO) == 2 ? 14 : 14) * SST;
                fetch_row<TIO, (SNY > 63 ? 63 : SNY), (ri % (STG > 0 ? STG : 1)) * G::SLOTB>(raw[ri], stg_tra);
            }
            if constexpr (!STG2) trow_pin<(NCOL * NLD + TS * NST > 63 ? 63 : NCOL * NLD + TS * NST)>(raw[ri]);
            if (row_valid(t)) {
                f32x2 row[NHP], odd[NHP - 1];
#pragma unroll
                for (int k = 0; k < NHP; ++k) {
                    const f32x2 xv = f32x2{raw_f32<TIO>(raw[ri][2 * k]), raw_f32<TIO>(raw[ri][2 * k + 1])};
                    if (MODE == 1) row[k] = xv + H[(i0 + 2) & 1][k];
                    else row[k] = pfma(splat(lam), H[(i1 + 2) & 1][k], pfma(splat(1.f - lam), H[(i0 + 2) & 1][k], xv));
                }
#pragma unroll
                for (int j = 0; j < NHP - 1; ++j) odd[j] = shift1(row[j], row[j + 1]);
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int o = t - u + 2;
                    if (o < 0 || o > TS - 1) continue;
                    f32x2(&a)[TH] = acc[o % 5];
#pragma unroll
                    for (int j = 0; j < TH; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), u == 0 ? bf : a[j]);     // u == 0: output row t + 2 enters the window
#pragma unroll
                    for (int j = 0; j < TH; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
                    for (int j = 0; j < TH; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
                    for (int j = 0; j < TH; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
                    for (int j = 0; j < TH; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
                }
            } else if constexpr (t + 2 >= 0 && t + 2 <= TS - 1) {  // a row outside the image: the output row it would have opened starts from the bias
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[(t + 2) % 5][j] = bf;
            }
            // output row t - 2 has seen its last input row
            if constexpr (t - 2 >= 0 && t - 2 <= TS - 1) {
                constexpr int o = t - 2;
                const int yrb = __builtin_amdgcn_readfirstlane((TS * trp + o) * (P0 * pix));
                trow_store<TIO, PIXB>(acc[o % 5], yoff, ysrc, yrb, pix);
            }
#pragma unroll
            for (int o = 0; o < TS; ++o) if (o > t - 2 && o <= t + 2) pin(acc[o % 5]);
            pin(H[0]);
            pin(H[1]);
            CPT_FENCE;
        });
    }
    CPT_STAMP(8);
    CPT_STAMP_RT(10);
    if constexpr (RCX_CPT_ENDBAR != 0) __syncthreads();      // the next unit's pass 1 writes F1 where this unit's pass 2 read C1
  }
}

static inline bool enabled()
{
    const char* v = rcx::opt::value(rcx::opt::CPT);
    const char* l = rcx::opt::value(rcx::opt::LANES);
    return !(v && *v == '0') && !(l && *l == '0');
}

// x rows by LDS-DMA + transposing reads (STG) where the kernel has the form and the data allow it: 16-bit activations, whole 16-byte chunks of
// eight channels (C % 8 == 0: every RecNeXt width) at 16-byte-aligned addresses, inference.  Parity-green (the 820 bf16 / float16 / golden cases of
// tests/test_recconv_gpu.py) and NOT faster inside a model -- 56x56 x 64: 123.3 us either way (a loop over one input: 112.2 against 113.8), 28x28 x 128:
// 55.6 against 53.6 (loop: 49.2 against 53.7), profiles/archive/r04_staged_rows.txt: with two waves per SIMD these kernels are bound by their vector-ALU
// issue, not by the 976 -> 451 vector-memory instructions per wave this removes.  So the instantiations exist in the diagnostic build only
// (make diag, RCX_AB_VARIANTS; RCX_CPT_STG=0 there: element loads).
template <int T, int HALVES, typename TIO, bool TRAIN> constexpr int stg_slots()
{
#ifndef RCX_AB_VARIANTS
    return 0;
#endif
    if (TRAIN || sizeof(TIO) != 2) return 0;
    if (T == 4 && HALVES == 2) return 3;
    if (T == 2 && HALVES == 1) return 2;
    return 0;
}
static inline bool stg_enabled() { return !rcx::opt::is_zero(rcx::opt::CPT_STG); }

template <int T, int HALVES, int MODE, int PIXB, typename TIO, bool TRAIN = false, int LV = (T == 4 ? 4 : 3), int STG = 0, int TS = 14>
static hipError_t launch(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    using G = Geo<T, HALVES, MODE, TIO, LV, STG, TS>;
    if constexpr (TS == 14 && STG == 0 && stg_slots<T, HALVES, TIO, TRAIN>() > 0) {
        if (!sv.base && C % 8 == 0 && ((size_t)x & 15) == 0 && stg_enabled())
            return launch<T, HALVES, MODE, PIXB, TIO, TRAIN, LV, stg_slots<T, HALVES, TIO, TRAIN>()>(x, y, wpack, bpack, N, C, s, sv);
    }
    if constexpr (TS == 14 && !TRAIN && MODE == 0 && HALVES != 4 && !(T == 2 && HALVES == 2) && LV == (T == 4 ? 4 : 3)) {      // training forward: bilinear only (what RecConv2d trains with), whole-block variants, full ladder
        if (sv.base) return launch<T, HALVES, MODE, PIXB, TIO, true>(x, y, wpack, bpack, N, C, s, sv);
    }
    if (!TRAIN && sv.base) return hipErrorInvalidConfiguration;
    auto kfn = k_recconv_cpt<T, HALVES, MODE, PIXB, TIO, TRAIN, LV, STG, TS>;
    RCX_SET_LDS_ONCE(kfn, G::LDS_BYTES);                       // once per instantiation and device
    static std::atomic<int> cus_cache{0};
    int cus = cus_cache.load(std::memory_order_relaxed);
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus = v;
        cus_cache.store(v, std::memory_order_relaxed);
    }
    const unsigned total = (unsigned)(N * ((C + G::CB - 1) / G::CB));
    // workgroups resident at once: 8 waves per CU (256 registers each) and 160 KB of LDS
    constexpr unsigned PER_CU = ((T == 4 && HALVES == 2) || (TS == 16 && RCX_CPT16_ALIAS == 0)) ? 1u : ((T == 2 && HALVES == 2) ? 4u : 2u);
    static_assert(PER_CU * G::LDS_BYTES <= 160 * 1024 && PER_CU * G::NW <= 8, "residency");
    unsigned cap = (unsigned)cus * PER_CU;
    if (const char* e = rcx::opt::value(rcx::opt::CPT_GRID)) { const int g = atoi(e); if (g > 0) cap = (unsigned)g; }    // A/B knob
    cap &= ~7u;
    const unsigned grid = total <= cap || cap == 0 ? total : cap;
    RCX_LAUNCH_TIMED(kfn, dim3(grid), dim3(G::NT), G::LDS_BYTES, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
    return hipGetLastError();
}

// the channel counts of RecNeXt-M3 / M4 get the compile-time pixel pitch (immediate column offsets), the rest the run-time one
template <int T, int HALVES, int MODE, typename TIO, int LV = (T == 4 ? 4 : 3)>
static hipError_t launch_c(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    constexpr int CM3 = T == 4 ? 64 : 128;
    if constexpr (LV != (T == 4 ? 4 : 3)) {                    // the shorter ladder (other resolutions, inner blocks): the run-time pitch only
        return launch<T, HALVES, MODE, 0, TIO, false, LV>(x, y, wpack, bpack, N, C, s, sv);
    } else {
    if (C == CM3) return launch<T, HALVES, MODE, CM3 * (int)sizeof(TIO), TIO>(x, y, wpack, bpack, N, C, s, sv);
    if constexpr (T == 2 && HALVES == 1) {
        // channel counts that are not multiples of 64 (RecNeXt-M1: 96, M5: 160, M0: 80): 32-channel workgroups of two waves, a wave = the two
        // tiles of a tile row, four workgroups per CU (round 3)
        if (C % 64 != 0 && !sv.base) return launch<2, 2, MODE, 0, TIO>(x, y, wpack, bpack, N, C, s, sv);
    }
    return launch<T, HALVES, MODE, 0, TIO>(x, y, wpack, bpack, N, C, s, sv);
    }
}

// The 56x56 block with 16-channel workgroups, two per CU (HALVES = 4), or with 32-channel workgroups (HALVES = 2).  Measured INSIDE the
// models (bench.py per-kernel event times, batch 256, bf16; profiles/archive/r03_cpt_cb16.txt): 64 channels 108.1 us with 32-channel blocks, 117.5
// with 16 (32-byte runs per cache-line access); 48 channels (RecNeXt-M1) 111.6 -> 92.5 us and 80 channels (M5) 175.6 -> 157.2 us with
// 16 (no half-empty last block); few units (3 images x 48 channels) 53.7 -> 37.2 us.  So: 16 where the channel count is not a multiple
// of 32 or where 32-channel units would not fill the chip; RCX_CPT_CB=16 / 32 pins either (A/B).
static inline bool cb16(int N, int C)
{
    const char* v = rcx::opt::value(rcx::opt::CPT_CB);
    if (v && v[0] == '1') return true;
    if (v && v[0] == '3') return false;
    return C % 32 != 0 || (long long)N * ((C + 31) / 32) < 256;
}

template <int T, int HALVES, int LV = (T == 4 ? 4 : 3)>
static hipError_t launch_md(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s, const SavedPyr& sv)
{
    if (dtype == 1) return mode == 1 ? launch_c<T, HALVES, 1, bf16_t, LV>(x, y, wpack, bpack, N, C, s, sv) : launch_c<T, HALVES, 0, bf16_t, LV>(x, y, wpack, bpack, N, C, s, sv);
    if (dtype == 2) return mode == 1 ? launch_c<T, HALVES, 1, f16_t, LV>(x, y, wpack, bpack, N, C, s, sv) : launch_c<T, HALVES, 0, f16_t, LV>(x, y, wpack, bpack, N, C, s, sv);
    return mode == 1 ? launch_c<T, HALVES, 1, float, LV>(x, y, wpack, bpack, N, C, s, sv) : launch_c<T, HALVES, 0, float, LV>(x, y, wpack, bpack, N, C, s, sv);
}

// the 64 x 64 / level 3 block on 16-pixel tiles (TS = 16; rcx_cpt4.hip): 16-channel workgroups of four waves, run-time pixel pitch
template <int MODE, typename TIO>
static hipError_t launch16(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s)
{
    const SavedPyr sv{};
    return launch<4, 4, MODE, 0, TIO, false, 3, 0, 16>(x, y, wpack, bpack, N, C, s, sv);
}

}  // namespace cpt
}  // namespace rcx
