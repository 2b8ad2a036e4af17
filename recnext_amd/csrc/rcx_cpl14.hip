// Channel-per-lane RecConv2d for the 14x14 / level 2 block (model/recnext.py:24-34 at stage 2 of RecNeXt-M*: 13 of M3's 21
// blocks).  One LANE owns a whole (image, channel) plane, a wave owns 64 consecutive channels of one image (128 contiguous
// bytes per pixel in NHWC bf16), so nothing is ever exchanged between lanes: no DPP (rcx_lanes.h: a DPP move puts the SIMD
// into its slow issue mode), no LDS, no barrier; the zero padding is resolved at compile time (border taps are not issued).
//
// Everything lives in registers as float32 PAIRS of horizontally adjacent pixels, (x[2j], x[2j+1]) in an even-aligned
// register pair, so that every convolution is v_pk_fma_f32 on operands that already sit where the instruction wants them:
//   * stride-1 convs (14x14, 7x7, 4x4) pair COLUMNS: acc(2j,2j+1) += in(2j+d, 2j+1+d) * splat(w); even d reads the plane's
//     own pairs, odd d reads a second copy of the row shifted by one pixel (one v_pk_mov_b32 per pair, built once per row);
//   * stride-2 convs (14->7, 7->4) pair TAPS: acc += in(2i-2,2i-1)*(w0,w1) + in(2i,2i+1)*(w2,w3), the two halves added at
//     the end -- the de-interleave a strided conv needs is free in this layout;
//   * both 14x14 passes stream x one row at a time and are input-row stationary: a row is scattered into the F1 rows (pass 1)
//     or into a ring of five output accumulator rows (pass 2) it feeds; finished rows leave as bf16 straight from registers.
// x is read from HBM once: between its two uses it waits in the ACCUMULATOR half of the unified 512-entry register file
// (v_accvgpr_write / v_accvgpr_read, one move each way).  The kernel therefore runs ONE wave per SIMD by design -- 256 x 256
// planes / 64 = 1024 waves = the chip's 1024 SIMDs -- where a lone wave issues one instruction per ~4 cycles of any kind,
// which is exactly what a packed FMA costs: the arithmetic runs at the vector peak as long as the instruction stream is
// (almost) nothing but packed FMAs.  All addressing is scalar: uniform base (SGPR pair) + per-lane 32-bit byte offset +
// immediate, no vector address arithmetic.  Same arithmetic as the other schedules: float32 throughout, one rounding at the
// final store.
#include "rcx_cpl14_pieces.h"
#include "rcx_launch.h"
#include "rcx_opts.h"

namespace rcx {
namespace cpl14 {

// ======== x through LDS (16-bit I/O, whole 64-channel blocks): k_recconv_cpl14<..., XL = true> ========
// A wave can keep at most 63 memory operations in flight, and a per-lane 2-byte load moves 128 bytes per wave: 8 KB in flight per
// SIMD, which at the chip's ~2 us loaded latency makes pass 1 latency-bound (rocprofv3: 19.3 us, a third of the wave's cycles in
// s_waitcnt).  LDS-DMA moves 1 KB per instruction: the wave requests its whole 14 x 14 x 64-channel plane (25 KB) with 25
// instructions up front and touches the rows as they land (counted vmcnt waits); a lane then reads its channel's pixels with
// ds_read_u16_d16_hi (the 16 bits arrive in float32 position, the other half zero-filled: tools/ubench/d16_probe.hip).  The same
// LDS image serves pass 2: no AGPR stash, no second HBM read.  One wave = one workgroup = 25 KB of LDS.
typedef int i32x4 __attribute__((ext_vector_type(4)));

// the whole plane: 25 x buffer_load_dwordx4 ... lds (lane l: pixel 8 i + l / 8, 16-byte chunk l % 8 -> LDS byte 1024 i + 16 l, i.e. the
// LDS image is [pixel][128 bytes]).  M0 carries the LDS destination; it is saved and restored (the compiler does not know).
// v1 = the lanes' offsets for the last instruction, whose pixels past 195 are redirected to pixel 195.
__device__ __forceinline__ void dma_plane(i32x4 xsrc, unsigned v0, unsigned v1, unsigned lds_base, int step)
{
    unsigned keep;
    int t;
    asm volatile("s_mov_b32 %[keep], m0\n\t"
                 "s_mov_b32 %[t], 0\n\t"
                 "s_mov_b32 m0, %[lds]\n\t"
                 ".rept 24\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %[v0], %[rs], %[t] offen lds\n\t"
                 "s_add_i32 %[t], %[t], %[step]\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 ".endr\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %[v1], %[rs], %[t] offen lds\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep), [t] "=&s"(t)
                 : [rs] "s"(xsrc), [lds] "s"(lds_base), [step] "s"(step), [v0] "v"(v0), [v1] "v"(v1)
                 : "scc", "memory");
}

template <typename TIO> struct LdsPix {      // float32 I/O never takes the LDS form (50 KB per wave)
    template <int OFF> static __device__ __forceinline__ void ld(uint32_t&, unsigned) {}
};
template <> struct LdsPix<bf16_t> {
    template <int OFF> static __device__ __forceinline__ void ld(uint32_t& dst, unsigned laddr)
    {
        asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(dst) : "v"(laddr), "n"(OFF));
    }
};
template <> struct LdsPix<f16_t> {
    template <int OFF> static __device__ __forceinline__ void ld(uint32_t& dst, unsigned laddr)
    {
        asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(laddr), "n"(OFF));
    }
};
// first touch of a row read from LDS: LDS operations complete in order, PENDING = the reads issued after this row's
template <int PENDING>
__device__ __forceinline__ void pin_lds_row(uint32_t (&v)[14])
{
    asm volatile("s_waitcnt lgkmcnt(%14)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]) : "n"(PENDING));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// the 25 taps of the down conv, requested by hand BEFORE the plane (they complete first; a compiler-visible load would make hipcc
// drain the DMAs at its first use): two statements of 13 and 12 loads, scalar offset walked inside
#define RCX_TL "buffer_load_dword %"
#define RCX_TSTEP "s_add_i32 %[t], %[t], %[st]\n\t"
__device__ __forceinline__ void load_taps_asm(float (&w)[25], i32x4 wsrc, unsigned vow, int base, int stride)
{
    int t;
    asm volatile("s_add_i32 %[t], %[b], 0\n\t"
                 RCX_TL "0, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "1, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "2, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "3, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "4, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "5, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "6, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "7, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "8, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "9, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "10, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "11, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "12, %[vo], %[rs], %[t] offen"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]), "=&v"(w[8]), "=&v"(w[9]),
                   "=&v"(w[10]), "=&v"(w[11]), "=&v"(w[12]), [t] "=&s"(t)
                 : [vo] "v"(vow), [rs] "s"(wsrc), [b] "s"(base), [st] "s"(stride) : "scc");
    asm volatile("s_add_i32 %[t], %[b], 0\n\t"
                 RCX_TL "0, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "1, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "2, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "3, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "4, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "5, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "6, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "7, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "8, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "9, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP
                 RCX_TL "10, %[vo], %[rs], %[t] offen\n\t" RCX_TSTEP RCX_TL "11, %[vo], %[rs], %[t] offen"
                 : "=&v"(w[13]), "=&v"(w[14]), "=&v"(w[15]), "=&v"(w[16]), "=&v"(w[17]), "=&v"(w[18]), "=&v"(w[19]), "=&v"(w[20]), "=&v"(w[21]),
                   "=&v"(w[22]), "=&v"(w[23]), "=&v"(w[24]), [t] "=&s"(t)
                 : [vo] "v"(vow), [rs] "s"(wsrc), [b] "s"(base + 13 * stride), [st] "s"(stride) : "scc");
}
#undef RCX_TL
#undef RCX_TSTEP

// LV = levels of the block: 2 (14 -> 7 -> 4: RecNeXt's stage 2 at 224 x 224) or 1 (14 -> 7: stage 3 of a 448 x 448 input; inference only)
// RL = "reload" (round 3): x is not kept in the accumulator registers between the passes but read a second time (from L2) in pass 2, so a
// wave fits 256 registers and TWO waves share a SIMD.  Same arithmetic on the same values: bit-identical to the stash form.  It pays
// where the launch has more waves than the chip has SIMDs (N * ceil(C / 64) > 1 024: RecNeXt-M5's 256 x 320, any batch above 256 of M3):
// the stash form then runs a second round on a fraction of the chip.
template <int MODE, int CT, typename TIO, bool XL = false, int LV = 2, bool RL = false>
__global__ __launch_bounds__(64, RL ? 2 : 1)
void k_recconv_cpl14(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                     int N, int C_rt, int has_bias, SavedPyr sv)
{
    constexpr int W = 14, P = 7, W1 = 7, P1 = 4, W2 = 4, P2 = 2;
    const int C = CT > 0 ? CT : C_rt;
    const int nb = (C + 63) / 64;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so b and b+8 share an L2; give each XCD a contiguous
    // run of (image, channel block) units -- the channel blocks of one image then stream the same 128-byte-interleaved lines
    // through one L2 at about the same time
    unsigned b = blockIdx.x;
    const unsigned G = gridDim.x;
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;                                     // tail lanes of a ragged last block: EXEC-masked for the whole kernel
    const size_t pix = (size_t)C * sizeof(TIO);             // bytes between horizontally adjacent pixels (uniform)
    const gcptr xb = (gcptr)x + (size_t)n * W * W * pix;
    const gcptr yb = (gcptr)y + (size_t)n * W * W * pix;
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);

    // ---- pass 1, one x row at a time (loads run AHEAD rows in front: a wave cannot have more than 64 memory operations in
    // flight anyway): 14 elements straight from HBM (each load instruction moves 128 contiguous bytes per wave), kept for pass 2
    // in the accumulator half of the register file, and scattered into the F1 = down(x) rows they feed (14 -> 7, tap pairs)
    constexpr int AHEAD = 3;                                // 3 * 14 + 14 = 56 loads in flight at most (the counter holds 63)
    uint32_t raw[W][W];                                     // as loaded; only AHEAD + 1 rows are ever live
    float S[W][W];                                          // the stash (AGPRs; not with XL)
    f32x2 F1[W1][P1];
    f32x2 facc[3][W1];                                      // F1 rows in flight: (sum over even taps, sum over odd taps) per output
    auto load_row = [&](int r) {
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            PixLd<TIO>::template ld<decltype(immc)::value>(raw[r][decltype(qc)::value], base, voff);
        });
    };
    Taps td;                                                // the shared down conv (model/recnext.py:21, :28); its taps come from L2
    // XL: this lane's LDS byte address of pixel 0 of its channel; row r, column q is at + (14 r + q) * 128 (an immediate)
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];
    unsigned laddr = 0;
    uint32_t xl[2][W];                                      // XL: rows as read from LDS, one row ahead
    auto lds_row = [&](uint32_t (&v)[W], auto rc) {
        constexpr int r = decltype(rc)::value;
        lanes::sfor<W>([&](auto qc) { LdsPix<TIO>::template ld<(W * r + decltype(qc)::value) * 128>(v[decltype(qc)::value], laddr); });
    };
    if constexpr (XL) {
        const int lane = (int)threadIdx.x;
        const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xlds);
        laddr = lds_base + (unsigned)lane * 2u;
        i32x4 xsrc, wsrc;
        {
            const unsigned long long a = (unsigned long long)xb;
            xsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            xsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
            xsrc.z = W * W * (int)pix;
            xsrc.w = 0x00020000;
            const unsigned long long wa = (unsigned long long)wpack;
            wsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)wa);
            wsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(wa >> 32) & 0xffff);
            wsrc.z = 5 * 25 * C * 4;
            wsrc.w = 0x00020000;
        }
        td.bias = 0.f;
        if (has_bias) {                                      // rare: fetched and waited for before anything is in flight
            td.bias = gload<float>((gcptr)bpack + vow);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(td.bias));
        }
        float tw[25];
        load_taps_asm(tw, wsrc, vow, 0, C * 4);
        const int pl = lane >> 3, ch16 = (lane & 7) * 16 + cb * 128;
        const int plast = 192 + pl > 195 ? 3 : pl;          // the last instruction covers pixels 192 .. 199: past 195 re-read 195
        dma_plane(xsrc, (unsigned)(pl * (int)pix + ch16), (unsigned)(plast * (int)pix + ch16), lds_base, 8 * (int)pix);
        // row 0 = pixels 0 .. 13 = DMA 0 and 1: 23 younger ones may still be in flight; the taps were requested before and are in
        wait_vm<23>();
#pragma unroll
        for (int i = 0; i < 25; ++i) asm volatile("" : "+v"(tw[i]));
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            td.p[u][0] = f32x2{tw[u * 5], tw[u * 5 + 1]};
            td.p[u][1] = f32x2{tw[u * 5 + 2], tw[u * 5 + 3]};
            td.p[u][2] = f32x2{tw[u * 5 + 4], 0.f};
        }
        lds_row(xl[0], lanes::IC<0>{});
    } else {
        lanes::sfor<AHEAD>([&](auto rc) { load_row(decltype(rc)::value); });
        load_taps<CT>(td, wpack, bpack, 0, C, vow, has_bias);   // behind the first x rows: the compiler's wait for them covers those too
    }
    lanes::sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        f32x2 xr[P];
        if constexpr (XL) {
            if constexpr (r + 1 < W) {
                // row r + 1 = pixels up to 14 r + 27 must have landed before it is read: DMA index (14 r + 27) / 8
                wait_vm<24 - (W * (r + 1) + W - 1) / 8>();
                lds_row(xl[(r + 1) & 1], lanes::IC<r + 1>{});
            }
            pin_lds_row<(r + 1 < W ? W : 0)>(xl[r & 1]);
#pragma unroll
            for (int j = 0; j < P; ++j) xr[j] = f32x2{PixLd<TIO>::cvt(xl[r & 1][2 * j]), PixLd<TIO>::cvt(xl[r & 1][2 * j + 1])};
        } else {
            if constexpr (r + AHEAD < W) load_row(r + AHEAD);
            // the row requested AHEAD rows ago is first touched HERE: one counted wait for the whole row (the younger loads stay in flight)
            pin_row<14 * (W - 1 - r < AHEAD ? W - 1 - r : AHEAD)>(raw[r]);
#pragma unroll
            for (int j = 0; j < P; ++j) xr[j] = f32x2{PixLd<TIO>::cvt(raw[r][2 * j]), PixLd<TIO>::cvt(raw[r][2 * j + 1])};
            if constexpr (!RL) {
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    S[r][2 * j] = stash(xr[j].x);
                    S[r][2 * j + 1] = stash(xr[j].y);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < W1; ++o) {
            const int u = r - 2 * o + 2;
            if (u < 0 || u > 4) continue;
            f32x2(&a)[W1] = facc[o % 3];
            const bool first = (u == 0) || (r == 0);                     // first input row of output row o
            if (first) {
#pragma unroll
                for (int i = 0; i < W1; ++i) a[i] = f32x2{td.bias, 0.f};
            }
#pragma unroll
            for (int i = 1; i < W1; ++i) a[i] = pfma(xr[i - 1], td.p[u][0], a[i]);
#pragma unroll
            for (int i = 0; i < W1; ++i) a[i] = pfma(xr[i], td.p[u][1], a[i]);
#pragma unroll
            for (int i = 0; i + 1 < P; ++i) a[i].x = fmaf(xr[i + 1].x, td.p[u][2].x, a[i].x);
            const bool last = (u == 4) || (r == W - 1);
            if (last) {
#pragma unroll
                for (int i = 0; i < W1; ++i) {
                    const float v = a[i].x + a[i].y;
                    if (i & 1) F1[o][i >> 1].y = v;
                    else F1[o][i >> 1].x = v;
                }
                F1[o][P1 - 1].y = 0.f;
                pin(F1[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < W1; ++o) if (r - 2 * o + 2 >= 0 && r - 2 * o + 2 <= 4 && !((r - 2 * o + 2 == 4) || (r == W - 1))) pin(facc[o % 3]);
        RCX_FENCE;
    });
    RCX_FENCE;

    // RL: the first rows of x again, requested before the level-1 block so that they arrive behind it (ordered loads the compiler counts itself)
    constexpr int AHEAD2 = 2;
    uint32_t raw2[W][W];
    auto load_row2 = [&](int r) {
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            raw2[r][decltype(qc)::value] = SafeLd<TIO>::ld(base + decltype(immc)::value + voff);
        });
    };
    if constexpr (RL) lanes::sfor<AHEAD2>([&](auto rc) { load_row2(decltype(rc)::value); });
    if (sv.base) save_plane<W1>(sv.base, sv.f_off[0], n, C, c, F1);                               // training forward: F_1
    // ---- the level-1 block on the 7x7 plane: C1 = conv_1(F1 + resize(conv_0(down(F1))))          (:27-33)
    f32x2 C1[W1][P1];
    Taps t2;
    if constexpr (LV == 1) {                                // one level: C1 = conv_0(F1), the final conv is pack 2
        Taps t0;
        load_taps<CT>(t0, wpack, bpack, 1, C, vow, has_bias);
        RCX_FENCE;
        load_taps<CT>(t2, wpack, bpack, 2, C, vow, has_bias);
        RCX_FENCE;
        conv5_plane<W1>(F1, C1, t0);
    } else {
        Taps t0;                                            // tap sets are fetched one stage ahead of their use and no earlier
        load_taps<CT>(t0, wpack, bpack, 1, C, vow, has_bias);
        f32x2 F2[W2][P2];
        down5<W1, W2>(F1, F2, td);
        if (sv.base) save_plane<W2>(sv.base, sv.f_off[1], n, C, c, F2);                           // F_2
        Taps t1;
        load_taps<CT>(t1, wpack, bpack, 2, C, vow, has_bias);
        RCX_FENCE;
        f32x2 C2[W2][P2];
        conv5_plane<W2>(F2, C2, t0);
        if (sv.base) save_plane<W2>(sv.base, sv.c_off[1], n, C, c, C2);                           // C_2
        f32x2 H2[W2][P1];
#pragma unroll
        for (int i = 0; i < W2; ++i) resize_row<MODE, W2, W1>(C2[i], H2[i]);
#pragma unroll
        for (int r = 0; r < W1; ++r) add_resized_row<MODE, W2, W1, P1>(F1[r], H2, r);       // F1 becomes T1
        pin(F1);
        RCX_FENCE;
        load_taps<CT>(t2, wpack, bpack, 3, C, vow, has_bias);
        RCX_FENCE;
        conv5_plane<W1>(F1, C1, t1);
        if (sv.base) save_plane<W1>(sv.base, sv.c_off[0], n, C, c, C1);                           // C_1
    }

    // ---- y = conv_2(x + resize(C1)): input-row stationary, five accumulator rows in flight                     (:34)
    f32x2 H1[W1][P];            // C1 rows resized horizontally, each computed just before its first use
    f32x2 acc[5][P];
#pragma unroll
    for (int t = 0; t < W; ++t) {
        // accumulator rows entering the window (rows 0..2 with the first input row, then row t+2); constant trip counts with
        // constant-folded conditions: a loop whose bounds depend on t is not unrolled and would push acc[] into scratch
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const bool enters = (t == 0) ? (s <= 2) : (s == (t + 2) % 5 && t + 2 < W);
            if (enters) {
#pragma unroll
                for (int j = 0; j < P; ++j) acc[s][j] = splat(t2.bias);
            }
        }
        const VT vt = vtab(MODE, W1, W, t);
#pragma unroll
        for (int i = 0; i < W1; ++i) {
            const bool first_use = (i == vt.i0 || i == vt.i1) && (t == 0 || (i != vtab(MODE, W1, W, t - 1).i0 && i != vtab(MODE, W1, W, t - 1).i1));
            if (first_use) resize_row<MODE, W1, W>(C1[i], H1[i]);
        }
        f32x2 row[P];
        if constexpr (XL) {
            // x a second time, from the LDS image (one row ahead; the row base costs one add, the columns are immediates)
            auto rd = [&](uint32_t (&v)[W], int tr) {
                const unsigned ra_ = laddr + (unsigned)(tr * W * 128);
                lanes::sfor<W>([&](auto qc) { LdsPix<TIO>::template ld<decltype(qc)::value * 128>(v[decltype(qc)::value], ra_); });
            };
            if (t == 0) rd(xl[0], 0);
            if (t + 1 < W) { rd(xl[(t + 1) & 1], t + 1); pin_lds_row<W>(xl[t & 1]); }
            else pin_lds_row<0>(xl[t & 1]);
#pragma unroll
            for (int j = 0; j < P; ++j) row[j] = f32x2{PixLd<TIO>::cvt(xl[t & 1][2 * j]), PixLd<TIO>::cvt(xl[t & 1][2 * j + 1])};
        } else if constexpr (RL) {
            if (t + AHEAD2 < W) load_row2(t + AHEAD2);
            pin_raw(raw2[t]);
#pragma unroll
            for (int j = 0; j < P; ++j) row[j] = f32x2{SafeLd<TIO>::cvt(raw2[t][2 * j]), SafeLd<TIO>::cvt(raw2[t][2 * j + 1])};
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) row[j] = f32x2{unstash(S[t][2 * j]), unstash(S[t][2 * j + 1])};
        }
        add_resized_row<MODE, W1, W, P>(row, H1, t);                                         // T0 row t
        conv5_row<W>(row, t, t2, [&](int o) -> f32x2(&)[P] { return acc[o % 5]; });
        // rows that have seen their last input row leave: row t-2, and with the last input row also rows 12 and 13
#pragma unroll
        for (int d = 2; d >= 0; --d) {
            const int o = t - d;
            if (o < 0 || (d < 2 && t != W - 1)) continue;
            typename PixSt<TIO>::packed pk[P];
#pragma unroll
            for (int j = 0; j < P; ++j) pk[j] = PixSt<TIO>::prep(acc[o % 5][j]);
            ra.row(yb, o, [&](auto qc, gcptr base, unsigned voff, auto immc) {
                constexpr int q = decltype(qc)::value;
                PixSt<TIO>::st(base + decltype(immc)::value + voff, pk[q >> 1], q & 1);
            });
        }
#pragma unroll
        for (int o = 0; o < W; ++o) if (o > t - 2 && o <= t + 2 && t != W - 1) pin(acc[o % 5]);
        RCX_FENCE;
    }
}

// ---- the 7x7 / level 1 block (last stage of RecNeXt-M*) from the same pieces: y = conv_1(x + resize(conv_0(down(x)))).
// 2048 waves at N = 256, C = 512: two per SIMD, so the register budget is 256 and nothing is stashed.
template <int MODE, int CT, typename TIO>
__global__ __launch_bounds__(64, 2)
void k_recconv_cpl7b(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack,
                     int N, int C_rt, int has_bias, SavedPyr sv)
{
    constexpr int W = 7, P = 4, W1 = 4, P1 = 2;
    const int C = CT > 0 ? CT : C_rt;
    const int nb = (C + 63) / 64;
    unsigned b = blockIdx.x;
    const unsigned G = gridDim.x;
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);                 // XCD-aware order, as above
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;
    const size_t pix = (size_t)C * sizeof(TIO);
    const gcptr xb = (gcptr)x + (size_t)n * W * W * pix;
    const gcptr yb = (gcptr)y + (size_t)n * W * W * pix;
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);

    uint32_t raw[W][W];
    lanes::sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            PixLd<TIO>::template ld<decltype(immc)::value>(raw[r][decltype(qc)::value], base, voff);
        });
    });
    Taps td, t0;
    load_taps<CT>(td, wpack, bpack, 0, C, vow, has_bias);
    load_taps<CT>(t0, wpack, bpack, 1, C, vow, has_bias);
    f32x2 X[W][P];
    lanes::sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        pin_row<7 * (W - 1 - r)>(raw[r]);
#pragma unroll
        for (int j = 0; j < P; ++j) X[r][j] = f32x2{PixLd<TIO>::cvt(raw[r][2 * j]), 2 * j + 1 < W ? PixLd<TIO>::cvt(raw[r][2 * j + 1]) : 0.f};
    });
    RCX_FENCE;
    f32x2 F1[W1][P1];
    down5<W, W1>(X, F1, td);
    if (sv.base) save_plane<W1>(sv.base, sv.f_off[0], n, C, c, F1);                               // training forward: F_1
    Taps t1;
    load_taps<CT>(t1, wpack, bpack, 2, C, vow, has_bias);
    RCX_FENCE;
    f32x2 C1[W1][P1];
    conv5_plane<W1>(F1, C1, t0);
    if (sv.base) save_plane<W1>(sv.base, sv.c_off[0], n, C, c, C1);                               // C_1
    f32x2 H1[W1][P];
#pragma unroll
    for (int i = 0; i < W1; ++i) resize_row<MODE, W1, W>(C1[i], H1[i]);
#pragma unroll
    for (int r = 0; r < W; ++r) add_resized_row<MODE, W1, W, P>(X[r], H1, r);                  // X becomes T0
    pin(X);
    RCX_FENCE;
    f32x2 Y[W][P];
    conv5_plane<W>(X, Y, t1);
    lanes::sfor<W>([&](auto rc) {
        constexpr int o = decltype(rc)::value;
        typename PixSt<TIO>::packed pk[P];
#pragma unroll
        for (int j = 0; j < P; ++j) pk[j] = PixSt<TIO>::prep(Y[o][j]);
        ra.row(yb, o, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            constexpr int q = decltype(qc)::value;
            PixSt<TIO>::st(base + decltype(immc)::value + voff, pk[q >> 1], q & 1);
        });
    });
}

// ---- one step on its own: y = conv5(x + resize(coarse)) + bias on the 14x14 plane (coarse 7x7) -- RecAttn2d's fused
// "upsample + add + depthwise conv" (model/recattn.py:54-67 after ConvNorm.fuse: rcx_upadd_dwconv_fwd) and the training forward's
// last step.  k_recconv_cpl14's pass 2 with the coarse plane read from memory: x streams once, nothing is stashed.  One wave per SIMD
// like its parent: under a 256-register cap the compiler spills, and what it spills are destinations of loads still in flight
// (tools/check_asm_hazards.py flags exactly that).
template <int MODE, int CT, typename TIO, typename TC>
__global__ __launch_bounds__(64)
void k_upadd_cpl14(const TIO* __restrict__ x, const TC* __restrict__ coarse, TIO* __restrict__ y, const float* __restrict__ w,
                   const float* __restrict__ bias, int N, int C_rt)
{
    constexpr int W = 14, P = 7, W1 = 7, P1 = 4;
    const int C = CT > 0 ? CT : C_rt;
    const int nb = (C + 63) / 64;
    unsigned b = blockIdx.x;
    const unsigned G = gridDim.x;
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);                 // XCD-aware order, as above
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;
    const size_t pix = (size_t)C * sizeof(TIO);
    const gcptr xb = (gcptr)x + (size_t)n * W * W * pix;
    const gcptr yb = (gcptr)y + (size_t)n * W * W * pix;
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);
    constexpr int AHEAD = 3;
    uint32_t raw[W][W];
    // ordered loads the compiler counts itself (SafeLd): in this kernel it copies freshly loaded registers around, which hand-issued
    // loads do not survive (tools/check_asm_hazards.py)
    auto load_row = [&](int r) {
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            raw[r][decltype(qc)::value] = SafeLd<TIO>::ld(base + decltype(immc)::value + voff);
        });
    };
    lanes::sfor<AHEAD>([&](auto rc) { load_row(decltype(rc)::value); });
    // the coarse plane and the taps behind the first x rows
    f32x2 C1[W1][P1];
    {
        const TC* q = coarse + ((size_t)n * W1 * W1) * C + c;
#pragma unroll
        for (int o = 0; o < W1; ++o)
#pragma unroll
            for (int j = 0; j < P1; ++j)
                C1[o][j] = f32x2{elem_to_f32(q[(size_t)(o * W1 + 2 * j) * C]), 2 * j + 1 < W1 ? elem_to_f32(q[(size_t)(o * W1 + 2 * j + 1) * C]) : 0.f};
    }
    Taps t2;
    load_taps<CT>(t2, w, bias, 0, C, vow, bias != nullptr);
    f32x2 H1[W1][P];
    f32x2 acc[5][P];
    lanes::sfor<W>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const bool enters = (t == 0) ? (s <= 2) : (s == (t + 2) % 5 && t + 2 < W);
            if (enters) {
#pragma unroll
                for (int j = 0; j < P; ++j) acc[s][j] = splat(t2.bias);
            }
        }
        if constexpr (t + AHEAD < W) load_row(t + AHEAD);
        const VT vt = vtab(MODE, W1, W, t);
#pragma unroll
        for (int i = 0; i < W1; ++i) {
            const bool first_use = (i == vt.i0 || i == vt.i1) && (t == 0 || (i != vtab(MODE, W1, W, t - 1).i0 && i != vtab(MODE, W1, W, t - 1).i1));
            if (first_use) resize_row<MODE, W1, W>(C1[i], H1[i]);
        }
        pin_raw(raw[t]);
        f32x2 row[P];
#pragma unroll
        for (int j = 0; j < P; ++j) row[j] = f32x2{SafeLd<TIO>::cvt(raw[t][2 * j]), SafeLd<TIO>::cvt(raw[t][2 * j + 1])};
        add_resized_row<MODE, W1, W, P>(row, H1, t);
        conv5_row<W>(row, t, t2, [&](int o) -> f32x2(&)[P] { return acc[o % 5]; });
#pragma unroll
        for (int d = 2; d >= 0; --d) {
            const int o = t - d;
            if (o < 0 || (d < 2 && t != W - 1)) continue;
            typename PixSt<TIO>::packed pk[P];
#pragma unroll
            for (int j = 0; j < P; ++j) pk[j] = PixSt<TIO>::prep(acc[o % 5][j]);
            ra.row(yb, o, [&](auto qc, gcptr base, unsigned voff, auto immc) {
                constexpr int q = decltype(qc)::value;
                PixSt<TIO>::st(base + decltype(immc)::value + voff, pk[q >> 1], q & 1);
            });
        }
#pragma unroll
        for (int o = 0; o < W; ++o) if (o > t - 2 && o <= t + 2 && t != W - 1) pin(acc[o % 5]);
        RCX_FENCE;
    });
}

// ---- the same two single steps on the 7x7 plane (coarse 4x4): RecAttn2d's last stage (model/recattn.py:61 / :67 at 7x7; RecNeXt-A's stage 3)
// ran them on the any-shape kernel (43.7 / 17.7 us at 256 x 512 against 20.5 / 13.6 for the 14x14 steps of four times the pixels).  The plane
// is held whole, as in k_recconv_cpl7b.
template <int MODE, int CT, typename TIO, typename TC>
__global__ __launch_bounds__(64)
void k_upadd_cpl7(const TIO* __restrict__ x, const TC* __restrict__ coarse, TIO* __restrict__ y, const float* __restrict__ w,
                  const float* __restrict__ bias, int N, int C_rt)
{
    constexpr int W = 7, P = 4, W1 = 4, P1 = 2;
    const int C = CT > 0 ? CT : C_rt;
    const int nb = (C + 63) / 64;
    unsigned b = blockIdx.x;
    const unsigned G = gridDim.x;
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);                 // XCD-aware order, as above
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;
    const size_t pix = (size_t)C * sizeof(TIO);
    const gcptr xb = (gcptr)x + (size_t)n * W * W * pix;
    const gcptr yb = (gcptr)y + (size_t)n * W * W * pix;
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);
    uint32_t raw[W][W];
    lanes::sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            raw[r][decltype(qc)::value] = SafeLd<TIO>::ld(base + decltype(immc)::value + voff);
        });
    });
    f32x2 C1[W1][P1];
    {
        const TC* q = coarse + ((size_t)n * W1 * W1) * C + c;
#pragma unroll
        for (int o = 0; o < W1; ++o)
#pragma unroll
            for (int j = 0; j < P1; ++j)
                C1[o][j] = f32x2{elem_to_f32(q[(size_t)(o * W1 + 2 * j) * C]), elem_to_f32(q[(size_t)(o * W1 + 2 * j + 1) * C])};
    }
    Taps t2;
    load_taps<CT>(t2, w, bias, 0, C, vow, bias != nullptr);
    f32x2 X[W][P];
#pragma unroll
    for (int r = 0; r < W; ++r)
#pragma unroll
        for (int j = 0; j < P; ++j) X[r][j] = f32x2{SafeLd<TIO>::cvt(raw[r][2 * j]), 2 * j + 1 < W ? SafeLd<TIO>::cvt(raw[r][2 * j + 1]) : 0.f};
    f32x2 H1[W1][P];
#pragma unroll
    for (int i = 0; i < W1; ++i) resize_row<MODE, W1, W>(C1[i], H1[i]);
#pragma unroll
    for (int r = 0; r < W; ++r) add_resized_row<MODE, W1, W, P>(X[r], H1, r);
    f32x2 Y[W][P];
    conv5_plane<W>(X, Y, t2);
    lanes::sfor<W>([&](auto rc) {
        constexpr int o = decltype(rc)::value;
        typename PixSt<TIO>::packed pk[P];
#pragma unroll
        for (int j = 0; j < P; ++j) pk[j] = PixSt<TIO>::prep(Y[o][j]);
        ra.row(yb, o, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            constexpr int q = decltype(qc)::value;
            PixSt<TIO>::st(base + decltype(immc)::value + voff, pk[q >> 1], q & 1);
        });
    });
}

// stride-2 conv5 of the 7x7 plane -> 4x4 (W = 7) or of the 14x14 plane -> 7x7 (W = 14), float32 out (the input of RecAttn2d's coarse chain)
template <int W, int CT, typename TIO>
__global__ __launch_bounds__(64)
void k_down5_cpl7(const TIO* __restrict__ x, float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias, int N, int C_rt)
{
    constexpr int P = (W + 1) / 2, W1 = (W + 1) / 2, P1 = (W1 + 1) / 2;
    const int C = CT > 0 ? CT : C_rt;
    const int nb = (C + 63) / 64;
    unsigned b = blockIdx.x;
    const unsigned G = gridDim.x;
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;
    const size_t pix = (size_t)C * sizeof(TIO);
    const gcptr xb = (gcptr)x + (size_t)n * W * W * pix;
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);
    uint32_t raw[W][W];
    lanes::sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            raw[r][decltype(qc)::value] = SafeLd<TIO>::ld(base + decltype(immc)::value + voff);
        });
    });
    Taps td;
    load_taps<CT>(td, w, bias, 0, C, vow, bias != nullptr);
    f32x2 X[W][P];
#pragma unroll
    for (int r = 0; r < W; ++r)
#pragma unroll
        for (int j = 0; j < P; ++j) X[r][j] = f32x2{SafeLd<TIO>::cvt(raw[r][2 * j]), 2 * j + 1 < W ? SafeLd<TIO>::cvt(raw[r][2 * j + 1]) : 0.f};
    f32x2 F1[W1][P1];
    down5<W, W1>(X, F1, td);
    save_plane<W1>(y, 0ull, n, C, c, F1);
}

template <int MODE, int CT, typename TIO, typename TC>
static hipError_t launch_up(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
    hipLaunchKernelGGL((k_upadd_cpl14<MODE, CT, TIO, TC>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (const TC*)coarse, (TIO*)y, w, b, N, C);
    return hipGetLastError();
}
template <int MODE, typename TIO, typename TC>
static hipError_t launch_up_c(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, hipStream_t s)
{
    if (C == 256) return launch_up<MODE, 256, TIO, TC>(x, coarse, y, w, b, N, C, s);
    return launch_up<MODE, 0, TIO, TC>(x, coarse, y, w, b, N, C, s);
}

// A/B switches, read per call like the other schedules' (tests flip them inside one process): RCX_CPL14=0 gives the block
// back to the lanes kernel, RCX_LANES=0 / RCX_FORCE_GENERIC=1 switch every register-resident schedule off
static inline bool enabled()
{
    const char* v = rcx::opt::value(rcx::opt::CPL14);
    const char* l = rcx::opt::value(rcx::opt::LANES);
    return !(v && *v == '0') && !(l && *l == '0');
}

// more waves than SIMDs (4 x 256 CUs): the reload form, two waves per SIMD, takes them in one round.  RCX_CPL14_RL=0 / 1 pins either form.
static inline bool use_reload(unsigned waves)
{
    const char* v = rcx::opt::value(rcx::opt::CPL14_RL);
    if (v && *v == '0') return false;
    if (v && *v == '1') return true;
    return waves > 1024u;
}

template <int CT, typename TIO> constexpr bool reload_built() { return CT > 0 && sizeof(TIO) == 2; }

template <int MODE, int CT, typename TIO>
static hipError_t launch(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
    // (the run-time-pitch instantiations spend 14 registers on column offsets, the float32 ones convert nothing in place: neither fits 256
    // registers without scratch, and they keep the stash form)
    if constexpr (reload_built<CT, TIO>()) {
        if (!sv.base && use_reload(grid)) {
            RCX_LAUNCH_TIMED((k_recconv_cpl14<MODE, CT, TIO, false, 2, true>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
            return hipGetLastError();
        }
    }
    RCX_LAUNCH_TIMED((k_recconv_cpl14<MODE, CT, TIO>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
    return hipGetLastError();
}

// x through LDS: 16-bit I/O and whole 64-channel blocks.  OFF by default (RCX_CPL14_LDS=1 switches it on for A/B runs): measured
// 19.96 us against 19.27 us for the register / AGPR-stash form at 256 x 256 x 14 x 14 bf16 (profiles/archive/r02c_cpl14_lds_variant.txt) --
// the kernel is bound by its ~5.3 k instructions at one wave per SIMD, not by the latency of its x loads.
// It is an A/B variant: only the diagnostic library (make diag: -DRCX_AB_VARIANTS) carries it (round 3: the shipped library has the one
// form that is used, and its asm-hazard scan -- part of `make all` -- reported this variant's f16 / nearest instantiation).
static inline bool use_xl(int C, int esz)
{
#ifdef RCX_AB_VARIANTS
    const char* v = rcx::opt::value(rcx::opt::CPL14_LDS);
    return esz == 2 && C % 64 == 0 && v && *v == '1';
#else
    (void)C; (void)esz;
    return false;
#endif
}

template <int MODE, int CT, typename TIO>
static hipError_t launch_xl(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    const unsigned grid = (unsigned)(N * (C / 64));
    RCX_LAUNCH_TIMED((k_recconv_cpl14<MODE, CT, TIO, true>), dim3(grid), dim3(64), 25 * 1024, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
    return hipGetLastError();
}

template <int MODE, typename TIO>
static hipError_t launch_c(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
#ifdef RCX_AB_VARIANTS
    if constexpr (sizeof(TIO) == 2) {
        if (use_xl(C, 2)) return C == 256 ? launch_xl<MODE, 256, TIO>(x, y, wpack, bpack, N, C, s, sv) : launch_xl<MODE, 0, TIO>(x, y, wpack, bpack, N, C, s, sv);
    }
#endif
    if (C == 256) return launch<MODE, 256, TIO>(x, y, wpack, bpack, N, C, s, sv);          // RecNeXt-M3/M4 stage 2: immediates instead of scalar adds
    if (C == 320) return launch<MODE, 320, TIO>(x, y, wpack, bpack, N, C, s, sv);          // RecNeXt-M5 stage 2 (BASELINE config 3): the reload form needs them to fit 256 registers
    if (C == 192) return launch<MODE, 192, TIO>(x, y, wpack, bpack, N, C, s, sv);          // RecNeXt-M1 stage 2 (BASELINE config 2)
    return launch<MODE, 0, TIO>(x, y, wpack, bpack, N, C, s, sv);
}

template <int MODE, int CT, typename TIO>
static hipError_t launch7(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
    RCX_LAUNCH_TIMED((k_recconv_cpl7b<MODE, CT, TIO>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
    return hipGetLastError();
}

// 14 x 14 / level 1 (stage 3 of a 448 x 448 input): the same kernel without its 4 x 4 level
template <int MODE, typename TIO>
static hipError_t launch_short(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s)
{
    const SavedPyr sv{};
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
    RCX_LAUNCH_TIMED((k_recconv_cpl14<MODE, 0, TIO, false, 1>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (TIO*)y, wpack, bpack, N, C, bpack != nullptr, sv);
    return hipGetLastError();
}

template <int MODE, typename TIO>
static hipError_t launch7_c(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, hipStream_t s, const SavedPyr& sv)
{
    if (C == 512) return launch7<MODE, 512, TIO>(x, y, wpack, bpack, N, C, s, sv);         // RecNeXt-M3/M4 stage 3
    return launch7<MODE, 0, TIO>(x, y, wpack, bpack, N, C, s, sv);
}

}  // namespace cpl14

// 14 x 14 / level 1: inference only (no saved pyramid, no fused backward behind it: cpl14_applicable stays the level-2 block)
bool cpl14_short_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    (void)N;
    return cpl14::enabled() && H == 14 && W == 14 && level == 1 && k == 5 && C >= 1 && (dtype == 0 || dtype == 1 || dtype == 2);
}

int cpl14_short_describe(int N, int C, int mode, char* buf, int len)
{
    return snprintf(buf, len, "cpl(k_recconv_cpl14<%d, 0>,levels-1,cb=64,nt=64,blocks=%d,lds=0)", mode, N * ((C + 63) / 64));
}

hipError_t cpl14_short_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s)
{
    if (dtype == 1) return mode == 1 ? cpl14::launch_short<1, bf16_t>(x, y, wpack, bpack, N, C, s) : cpl14::launch_short<0, bf16_t>(x, y, wpack, bpack, N, C, s);
    if (dtype == 2) return mode == 1 ? cpl14::launch_short<1, f16_t>(x, y, wpack, bpack, N, C, s) : cpl14::launch_short<0, f16_t>(x, y, wpack, bpack, N, C, s);
    return mode == 1 ? cpl14::launch_short<1, float>(x, y, wpack, bpack, N, C, s) : cpl14::launch_short<0, float>(x, y, wpack, bpack, N, C, s);
}

// the 7x7 / level 1 block on the pieces of this file (round 1's first version, rcx_cpl.hip, left the tree in round 3: profiles/archive/r01*, r02a_*)
bool cpl7b_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    (void)N;
    const char* all = rcx::opt::value(rcx::opt::CPL);                          // RCX_CPL=0: no channel-per-lane kernel on 7x7 (the lanes kernel instead)
    return cpl14::enabled() && !(all && *all == '0') && H == 7 && W == 7 && level == 1 && k == 5 && C >= 1 && (dtype == 0 || dtype == 1 || dtype == 2);
}

int cpl7b_describe(int N, int C, int mode, char* buf, int len)
{
    return snprintf(buf, len, "cpl(k_recconv_cpl7b<%d, %d>,cb=64,nt=64,blocks=%d,lds=0)", mode, C == 512 ? 512 : 0, N * ((C + 63) / 64));
}

hipError_t cpl7b_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s,
                         float* saved, const size_t* f_off, const size_t* c_off)
{
    cpl14::SavedPyr sv{};
    sv.base = saved;
    if (saved) { sv.f_off[0] = f_off[1]; sv.c_off[0] = c_off[1]; }
    if (dtype == 1) return mode == 1 ? cpl14::launch7_c<1, bf16_t>(x, y, wpack, bpack, N, C, s, sv) : cpl14::launch7_c<0, bf16_t>(x, y, wpack, bpack, N, C, s, sv);
    if (dtype == 2) return mode == 1 ? cpl14::launch7_c<1, f16_t>(x, y, wpack, bpack, N, C, s, sv) : cpl14::launch7_c<0, f16_t>(x, y, wpack, bpack, N, C, s, sv);
    return mode == 1 ? cpl14::launch7_c<1, float>(x, y, wpack, bpack, N, C, s, sv) : cpl14::launch7_c<0, float>(x, y, wpack, bpack, N, C, s, sv);
}

// y = conv5(x + resize(coarse)) on the 14x14 plane: coarse in the I/O type or float32
// RCX_UPADD_CPL=0: neither plane; RCX_UPADD_CPL=14: the 14x14 plane only (A/B of the 7x7 step kernels, round 4)
static bool cpl7_steps()
{
    const char* v = rcx::opt::value(rcx::opt::UPADD_CPL);
    return !(v && v[0] == '1' && v[1] == '4');
}

bool upadd_cpl14_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt)
{
    (void)N;
    const char* v = rcx::opt::value(rcx::opt::UPADD_CPL);
    const bool plane = (H == 14 && W == 14 && Hc == 7 && Wc == 7) || (H == 7 && W == 7 && Hc == 4 && Wc == 4 && cpl7_steps());
    return cpl14::enabled() && !(v && *v == '0') && plane && k == 5 && C >= 1 && out_dt == x_dt &&
           (x_dt == 0 || x_dt == 1 || x_dt == 2) && (c_dt == x_dt || c_dt == 0);
}

// RCX_UPADD_CPL=7: the 14x14 -> 7x7 step stays on the lanes kernel (A/B)
bool down5_cpl7_applicable(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt)
{
    (void)N;
    const char* v = rcx::opt::value(rcx::opt::UPADD_CPL);
    const bool plane = (H == 7 && W == 7) || (H == 14 && W == 14 && !(v && *v == '7') && in_dt != 0);
    return cpl14::enabled() && !(v && *v == '0') && cpl7_steps() && plane && k == 5 && stride == 2 && C >= 1 && out_dt == 0 &&
           (in_dt == 0 || in_dt == 1 || in_dt == 2);
}

hipError_t down5_cpl7(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int in_dt, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
#define RCX_D7(W_, CT_, T_) hipLaunchKernelGGL((cpl14::k_down5_cpl7<W_, CT_, T_>), dim3(grid), dim3(64), 0, s, (const T_*)x, (float*)y, w, b, N, C)
    if (H == 14) {
        if (C == 256) { if (in_dt == 1) RCX_D7(14, 256, bf16_t); else RCX_D7(14, 256, f16_t); }
        else { if (in_dt == 1) RCX_D7(14, 0, bf16_t); else RCX_D7(14, 0, f16_t); }
    } else if (C == 512) { if (in_dt == 0) RCX_D7(7, 512, float); else if (in_dt == 1) RCX_D7(7, 512, bf16_t); else RCX_D7(7, 512, f16_t); }
    else { if (in_dt == 0) RCX_D7(7, 0, float); else if (in_dt == 1) RCX_D7(7, 0, bf16_t); else RCX_D7(7, 0, f16_t); }
#undef RCX_D7
    return hipGetLastError();
}

namespace cpl14 {
template <int MODE, typename TIO, typename TC>
static hipError_t launch_up7(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
    if (C == 512) hipLaunchKernelGGL((k_upadd_cpl7<MODE, 512, TIO, TC>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (const TC*)coarse, (TIO*)y, w, b, N, C);
    else hipLaunchKernelGGL((k_upadd_cpl7<MODE, 0, TIO, TC>), dim3(grid), dim3(64), 0, s, (const TIO*)x, (const TC*)coarse, (TIO*)y, w, b, N, C);
    return hipGetLastError();
}
}  // namespace cpl14

hipError_t upadd_cpl14(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, int H, int mode, int x_dt, int c_dt,
                       hipStream_t s)
{
    if (H == 7) {
#define RCX_U7(MD_)                                                                                                                  \
    (x_dt == 0 ? cpl14::launch_up7<MD_, float, float>(x, coarse, y, w, b, N, C, s)                                                   \
     : x_dt == 1 ? (c_dt == 1 ? cpl14::launch_up7<MD_, bf16_t, bf16_t>(x, coarse, y, w, b, N, C, s) : cpl14::launch_up7<MD_, bf16_t, float>(x, coarse, y, w, b, N, C, s)) \
                 : (c_dt == 2 ? cpl14::launch_up7<MD_, f16_t, f16_t>(x, coarse, y, w, b, N, C, s) : cpl14::launch_up7<MD_, f16_t, float>(x, coarse, y, w, b, N, C, s)))
        return mode == 1 ? RCX_U7(1) : RCX_U7(0);
#undef RCX_U7
    }
#define RCX_UC(MD_)                                                                                                                  \
    (x_dt == 0 ? cpl14::launch_up_c<MD_, float, float>(x, coarse, y, w, b, N, C, s)                                                  \
     : x_dt == 1 ? (c_dt == 1 ? cpl14::launch_up_c<MD_, bf16_t, bf16_t>(x, coarse, y, w, b, N, C, s) : cpl14::launch_up_c<MD_, bf16_t, float>(x, coarse, y, w, b, N, C, s)) \
                 : (c_dt == 2 ? cpl14::launch_up_c<MD_, f16_t, f16_t>(x, coarse, y, w, b, N, C, s) : cpl14::launch_up_c<MD_, f16_t, float>(x, coarse, y, w, b, N, C, s)))
    return mode == 1 ? RCX_UC(1) : RCX_UC(0);
#undef RCX_UC
}

bool cpl14_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    (void)N;
    return cpl14::enabled() && H == 14 && W == 14 && level == 2 && k == 5 && C >= 1 && (dtype == 0 || dtype == 1 || dtype == 2);
}

int cpl14_describe(int N, int C, int mode, int dtype, char* buf, int len)
{
    const bool xl = cpl14::use_xl(C, dtype == 0 ? 4 : 2);
    const int ct = C == 256 || C == 320 || C == 192 ? C : 0;
    const bool rl = !xl && ct > 0 && dtype != 0 && cpl14::use_reload((unsigned)(N * ((C + 63) / 64)));       // inference launches (the training forward keeps the stash form)
    return snprintf(buf, len, "cpl(k_recconv_cpl14<%d, %d%s>,cb=64,nt=64,blocks=%d,lds=%d)", mode, ct, xl ? ", XL" : (rl ? ", RL" : ""), N * ((C + 63) / 64), xl ? 25 * 1024 : 0);
}

hipError_t cpl14_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s,
                         float* saved, const size_t* f_off, const size_t* c_off)
{
    cpl14::SavedPyr sv{};
    sv.base = saved;
    if (saved) { sv.f_off[0] = f_off[1]; sv.f_off[1] = f_off[2]; sv.c_off[0] = c_off[1]; sv.c_off[1] = c_off[2]; }
    if (dtype == 1) return mode == 1 ? cpl14::launch_c<1, bf16_t>(x, y, wpack, bpack, N, C, s, sv) : cpl14::launch_c<0, bf16_t>(x, y, wpack, bpack, N, C, s, sv);
    if (dtype == 2) return mode == 1 ? cpl14::launch_c<1, f16_t>(x, y, wpack, bpack, N, C, s, sv) : cpl14::launch_c<0, f16_t>(x, y, wpack, bpack, N, C, s, sv);
    return mode == 1 ? cpl14::launch_c<1, float>(x, y, wpack, bpack, N, C, s, sv) : cpl14::launch_c<0, float>(x, y, wpack, bpack, N, C, s, sv);
}

}  // namespace rcx
