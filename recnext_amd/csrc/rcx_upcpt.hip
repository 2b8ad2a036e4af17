// Single steps of the recursion on ANY even plane, channel per lane (round 3):
//
//   k_upadd_cpt   y = conv5(x + resize2x(coarse))   -- the last line of RecConv2d.forward (model/recnext.py:31-34) and of
//                                                      RecAttn2d.forward (model/recattn.py:67) as one launch
//
// It is pass 2 of k_recconv_cpt (rcx_cpt_kernel.h) made a kernel of its own: a wave = 64 channels of one 14 x 14 tile, a lane = one
// channel; x streams one row at a time from global memory into registers through the same hand-issued row statements (18 columns with
// the halo), the conv is input-row stationary on v_pk_fma_f32 with five accumulator rows in flight, a row of 14 outputs leaves as one
// statement.  What the fused kernel reads from its LDS planes -- the rows of the coarse plane, 11 columns per tile -- comes from global
// memory here, one row every second input row, hand-issued like x (the coarse plane is a quarter of x and stays in L2).  No LDS, no
// barrier, nothing shared between waves.
//
// Rows of x and y are addressed through PER-ROW buffer descriptors (row_desc below): a column left of the plane or right of it is out of
// range and reads 0, a store past the row's end is dropped.  So there are no edge flags, the last tile column may be ragged, and ANY even
// plane works (56, 28, 112, 128; the 200 x 336, 100 x 168, 50 x 84 stages of a COCO input; 200 x 334); the coarse columns are clamped
// into the plane one by one (ATen's border rule), the resized coarse row is masked to zero outside the plane pair by pair.  Any even
// height (rows past the plane are loaded from a valid row and not used, their stores go to an empty descriptor), any channel count
// (lanes past the last channel load channel C - 1 and store out of range).
//
// Every wave issues the SAME sequence of memory instructions whatever its tile (skipped rows still load, stores of absent rows are
// dropped by the hardware), so the s_waitcnt counts are exact compile-time numbers: `Sched` replays the issue order at compile time.
//
// TW = tile width: 14, or 16 where 16 divides the width and 14 does not (the 16 * 2^k planes of 256 x 256 / 512 x 512 inputs: no ragged
// column; bfloat16 only).  Tiles are 14 rows high either way.
#include "rcx_cpt_kernel.h"
#include "rcx_opts.h"

namespace rcx {
namespace upcpt {

using namespace cpt;

// tile width of a plane: 14; 16 where that divides the width and 14 does not (bfloat16 activations only: float16's extra conversion
// registers do not fit beside the 16-wide tile's accumulators, float32 has no 20-column row statement).  Any other width: 14-wide tiles with
// a ragged last column (rows are addressed through per-row descriptors: what lies past the row's end reads 0 and is not stored).
static inline int tile_width(int W, int x_dt) { return (W % 14 != 0 && W % 16 == 0 && x_dt == 1) ? 16 : 14; }

constexpr int NR = 18;            // input rows of a tile: -2 .. 15

// the order in which a wave issues its vector-memory instructions (identical for every wave), replayed at compile time
// AHEAD = x rows in flight in front of the row being used
template <int MODE, int TW, int AHEAD> struct Sched {
    static constexpr int NCOL = TW + 4, NCR = TW / 2 + 4, NSTORE = TW;
    static constexpr int c_last = MODE == 0 ? 8 : 7;                                     // last coarse row a tile needs
    static constexpr bool is_build(int ri) { return MODE == 0 ? (ri & 1) == 1 : ((ri & 1) == 0 && ri >= 2); }
    static constexpr int build_row(int ri) { return MODE == 0 ? (ri - 1) / 2 : (ri - 2) / 2; }
    // kind 0: coarse rows -2, -1 (prologue); 1: the coarse row built in iteration ri; 2: x row ri.  Result: memory instructions issued
    // after the awaited one and before the wait = the largest count the wait may leave outstanding.
    static constexpr int pending(int kind, int ri_target)
    {
        int seq = 0, xend[NR + AHEAD + 1] = {}, cend[16] = {};
        for (int i = -2; i <= 0; ++i) { seq += NCR; cend[i + 2] = seq; }
        for (int r = 0; r < AHEAD; ++r) { seq += NCOL; xend[r] = seq; }
        if (kind == 0) return seq - cend[1];
        for (int ri = 0; ri < NR; ++ri) {
            if (ri + AHEAD < NR) { seq += NCOL; xend[ri + AHEAD] = seq; }
            if (is_build(ri)) {
                const int ib = build_row(ri);
                if (kind == 1 && ri == ri_target) return seq - cend[ib + 2];
                if (ib + 1 <= c_last) { seq += NCR; cend[ib + 3] = seq; }                // requested right after row ib was consumed
            }
            if (kind == 2 && ri == ri_target) return seq - xend[ri];
            if (ri - 4 >= 0 && ri - 4 <= 13) seq += NSTORE;                              // output row t - 2 leaves at the end of the iteration
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};

// one row of the coarse plane: columns -2 .. TW/2 + 1 of the tile, each at its own scalar byte offset (column index clamped into the plane
// by the caller: ATen's border rule; a ragged last tile clamps in its middle)
#define UPC_OUT11(v) "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10])
#define UPC_C(OP, d) "s_add_i32 %[t], %[rb], %[c" #d "]\n\t" OP " %" #d ", %[vo], %[rs], %[t] offen\n\t"
#define UPC_ROW11(OP) UPC_C(OP, 0) UPC_C(OP, 1) UPC_C(OP, 2) UPC_C(OP, 3) UPC_C(OP, 4) UPC_C(OP, 5) UPC_C(OP, 6) UPC_C(OP, 7) UPC_C(OP, 8) UPC_C(OP, 9) UPC_C(OP, 10)
#define UPC_ROW12(OP) UPC_ROW11(OP) UPC_C(OP, 11)
#define UPC_IN11 [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [c0] "s"(ck[0]), [c1] "s"(ck[1]), [c2] "s"(ck[2]), [c3] "s"(ck[3]), [c4] "s"(ck[4]), [c5] "s"(ck[5]), \
                 [c6] "s"(ck[6]), [c7] "s"(ck[7]), [c8] "s"(ck[8]), [c9] "s"(ck[9]), [c10] "s"(ck[10])
template <typename TC>
__device__ __forceinline__ void coarse_row_load(uint32_t (&v)[11], unsigned vo, i32x4 rs, int rb, const int (&ck)[11])
{
    int t;
    if constexpr (std::is_same<TC, f16_t>::value) asm volatile(UPC_ROW11(CPT_LDH) : UPC_OUT11(v), [t] "=&s"(t) : UPC_IN11 : "scc");
    else if constexpr (sizeof(TC) == 2) asm volatile(UPC_ROW11(CPT_LD16) : UPC_OUT11(v), [t] "=&s"(t) : UPC_IN11 : "scc");
    else asm volatile(UPC_ROW11(CPT_LD32) : UPC_OUT11(v), [t] "=&s"(t) : UPC_IN11 : "scc");
}
template <typename TC>
__device__ __forceinline__ void coarse_row_load(uint32_t (&v)[12], unsigned vo, i32x4 rs, int rb, const int (&ck)[12])
{
    int t;
    if constexpr (std::is_same<TC, f16_t>::value) asm volatile(UPC_ROW12(CPT_LDH) : UPC_OUT11(v), "=&v"(v[11]), [t] "=&s"(t) : UPC_IN11, [c11] "s"(ck[11]) : "scc");
    else if constexpr (sizeof(TC) == 2) asm volatile(UPC_ROW12(CPT_LD16) : UPC_OUT11(v), "=&v"(v[11]), [t] "=&s"(t) : UPC_IN11, [c11] "s"(ck[11]) : "scc");
    else asm volatile(UPC_ROW12(CPT_LD32) : UPC_OUT11(v), "=&v"(v[11]), [t] "=&s"(t) : UPC_IN11, [c11] "s"(ck[11]) : "scc");
}
template <int PENDING> __device__ __forceinline__ void pin_coarse(uint32_t (&v)[11])
{
    asm volatile("s_waitcnt vmcnt(%11)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]) : "n"(PENDING));
}
template <int PENDING> __device__ __forceinline__ void pin_coarse(uint32_t (&v)[12])
{
    asm volatile("s_waitcnt vmcnt(%12)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]) : "n"(PENDING));
}

// ---- 16-wide tiles (16-bit activations): a row of 20 columns = -2, -1 (vl), 0 .. 15 (vm), 16, 17 (vr), and a row of 16 outputs
#define UPW_ROW_IMM(OP)                                                                                                              \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                  \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1)                                                                          \
    CPT_LI(OP, 2, "vm", "t", 0) CPT_LI(OP, 3, "vm", "t", 1) CPT_LI(OP, 4, "vm", "t", 2) CPT_LI(OP, 5, "vm", "t", 3)                  \
    CPT_LI(OP, 6, "vm", "t", 4) CPT_LI(OP, 7, "vm", "t", 5) CPT_LI(OP, 8, "vm", "t", 6) CPT_LI(OP, 9, "vm", "t", 7)                  \
    CPT_LI(OP, 10, "vm", "t", 8) CPT_LI(OP, 11, "vm", "t", 9) CPT_LI(OP, 12, "vm", "t", 10) CPT_LI(OP, 13, "vm", "t", 11)            \
    CPT_LI(OP, 14, "vm", "t", 12) CPT_LI(OP, 15, "vm", "t", 13) CPT_LI(OP, 16, "vm", "t", 14) CPT_LI(OP, 17, "vm", "t", 15)          \
    CPT_LI(OP, 18, "vr", "t", 0) CPT_LI(OP, 19, "vr", "t", 1)
#define UPW_ROW_GEN(OP)                                                                                                              \
    "s_add_i32 %[t], %[rb], 0\n\ts_add_i32 %[t2], %[rb], %[pix]\n\t"                                                                \
    CPT_LG(OP, 0, "vl", "t") CPT_LG(OP, 1, "vl", "t2") CPT_LG(OP, 18, "vr", "t") CPT_LG(OP, 19, "vr", "t2")                          \
    CPT_LG(OP, 2, "vm", "t") CPT_LG(OP, 3, "vm", "t2")                                                                               \
    CPT_LGN(OP, 4) CPT_LGN(OP, 5) CPT_LGN(OP, 6) CPT_LGN(OP, 7) CPT_LGN(OP, 8) CPT_LGN(OP, 9) CPT_LGN(OP, 10) CPT_LGN(OP, 11)         \
    CPT_LGN(OP, 12) CPT_LGN(OP, 13) CPT_LGN(OP, 14) CPT_LGN(OP, 15) CPT_LGN(OP, 16) CPT_LGN(OP, 17)
template <typename TIO, int PIXB>
__device__ __forceinline__ void row_load16w(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    static_assert(sizeof(TIO) == 2, "16-wide tiles take 16-bit activations");
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 15 <= 4095) {
        (void)pix; (void)t2;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(UPW_ROW_IMM(CPT_LDH) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(UPW_ROW_IMM(CPT_LD16) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else {
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(UPW_ROW_GEN(CPT_LDH) : CPT_OUT20(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else
            asm volatile(UPW_ROW_GEN(CPT_LD16) : CPT_OUT20(v), [t] "=&s"(t), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    }
}
template <typename T16, int PIXB>
__device__ __forceinline__ void row_store16w(const f32x2 (&a)[8], unsigned vo, i32x4 rs, int rb, int pix)
{
    uint32_t p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if constexpr (std::is_same<T16, f16_t>::value) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a[j].x), "v"(a[j].y));
        else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a[j].x), "v"(a[j].y));
    }
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 15 <= 4095) {
        (void)pix; (void)t2;
        asm volatile("s_add_i32 %[t], %[rb], 0\n\t"
                     CPT_SI("buffer_store_short", 0, "t", 0) CPT_SI("buffer_store_short_d16_hi", 0, "t", 1) CPT_SI("buffer_store_short", 1, "t", 2)
                     CPT_SI("buffer_store_short_d16_hi", 1, "t", 3) CPT_SI("buffer_store_short", 2, "t", 4) CPT_SI("buffer_store_short_d16_hi", 2, "t", 5)
                     CPT_SI("buffer_store_short", 3, "t", 6) CPT_SI("buffer_store_short_d16_hi", 3, "t", 7) CPT_SI("buffer_store_short", 4, "t", 8)
                     CPT_SI("buffer_store_short_d16_hi", 4, "t", 9) CPT_SI("buffer_store_short", 5, "t", 10) CPT_SI("buffer_store_short_d16_hi", 5, "t", 11)
                     CPT_SI("buffer_store_short", 6, "t", 12) CPT_SI("buffer_store_short_d16_hi", 6, "t", 13) CPT_SI("buffer_store_short", 7, "t", 14)
                     CPT_SI("buffer_store_short_d16_hi", 7, "t", 15)
                     : [t] "=&s"(t)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc", "memory");
    } else {
        asm volatile("s_add_i32 %[t2], %[rb], 0\n\t"
                     CPT_SG("buffer_store_short", 0, "t2") CPT_SGN("buffer_store_short_d16_hi", 0) CPT_SGN("buffer_store_short", 1) CPT_SGN("buffer_store_short_d16_hi", 1)
                     CPT_SGN("buffer_store_short", 2) CPT_SGN("buffer_store_short_d16_hi", 2) CPT_SGN("buffer_store_short", 3) CPT_SGN("buffer_store_short_d16_hi", 3)
                     CPT_SGN("buffer_store_short", 4) CPT_SGN("buffer_store_short_d16_hi", 4) CPT_SGN("buffer_store_short", 5) CPT_SGN("buffer_store_short_d16_hi", 5)
                     CPT_SGN("buffer_store_short", 6) CPT_SGN("buffer_store_short_d16_hi", 6) CPT_SGN("buffer_store_short", 7) CPT_SGN("buffer_store_short_d16_hi", 7)
                     : [t2] "=&s"(t2)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
        (void)t;
    }
}
// the tile-width-generic faces of the row statements
template <typename TIO, int PIXB> __device__ __forceinline__ void xrow_load(uint32_t (&v)[18], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix) { row_load<TIO, PIXB>(v, vl, vm, vr, rs, rb, pix); }
template <typename TIO, int PIXB> __device__ __forceinline__ void xrow_load(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix) { row_load16w<TIO, PIXB>(v, vl, vm, vr, rs, rb, pix); }
template <int PENDING> __device__ __forceinline__ void xrow_pin(uint32_t (&v)[18]) { pin_row<PENDING>(v); }
template <int PENDING> __device__ __forceinline__ void xrow_pin(uint32_t (&v)[20]) { pin_row20<PENDING>(v); }
template <typename TIO, int PIXB> __device__ __forceinline__ void yrow_store(const f32x2 (&a)[7], unsigned vo, i32x4 rs, int rb, int pix) { RowSt<TIO, PIXB>::st(a, vo, rs, rb, pix); }
template <typename TIO, int PIXB> __device__ __forceinline__ void yrow_store(const f32x2 (&a)[8], unsigned vo, i32x4 rs, int rb, int pix) { row_store16w<TIO, PIXB>(a, vo, rs, rb, pix); }

// A row of a plane as a buffer of its own (base = the row, num_records = its bytes): everything left and right of it -- and, with zero
// records, an absent row -- is out of range: loads return 0, stores are dropped.  No edge flags, and the last tile column may be ragged.
// Scalar arithmetic on uniform values only: a v_readfirstlane_b32 here would write the descriptor's SGPRs from the vector pipe right in
// front of the loads that read them (5 wait states the compiler does not insert inside an asm statement; tools/check_asm_hazards.py).
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

// MODE 0 bilinear (exact 2x: weights 1/4, 3/4, clamped borders = ATen's align_corners=False arithmetic), 1 nearest.
// PIXB = bytes per pixel of x and y when known at compile time (64 or 128 channels of a 16-bit type), 0 = run time.
template <int MODE, int PIXB, typename TIO, typename TC, int TW = 14>
__global__ __launch_bounds__(256, 2) void k_upadd_cpt(const TIO* __restrict__ x, const TC* __restrict__ coarse, TIO* __restrict__ y,
                                                      const float* __restrict__ w, const float* __restrict__ bias, int N, int C, int H, int W, int has_bias)
{
    constexpr int ESZ = (int)sizeof(TIO), CSZ = (int)sizeof(TC), NP = TW / 2, NCOL = TW + 4, NCR = TW / 2 + 4, NHP = NCOL / 2;
    static_assert(TW == 14 || (TW == 16 && sizeof(TIO) == 2), "tile width");
    // float16 converts every element into a second register: one row less in flight keeps the kernel inside 256 registers (a spilled
    // row register would be stored before its load has landed)
    constexpr int AHEAD = (std::is_same<TIO, f16_t>::value || TW == 16) ? 1 : 2;
    using S = Sched<MODE, TW, AHEAD>;
    const int nb = (C + 63) / 64, TR = (H + 13) / 14, TCn = (W + TW - 1) / TW, Hc = H / 2, Wc = W / 2;
    const int pix = PIXB ? PIXB : C * ESZ, pixc = C * CSZ;
    const unsigned total = (unsigned)N * nb * TR * TCn;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned upp = (unsigned)(TR * TCn);                    // tile-waves per (image, channel block) plane
    const unsigned wg = (upp & 3u) == 0 ? xcd_workgroup(blockIdx.x, upp >> 2, total / upp) : blockIdx.x;
    const unsigned unit = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (unit >= total) return;
    // tile column fastest: the four waves of a workgroup are horizontal neighbours (their halo columns are each other's interiors)
    const int tc = (int)(unit % (unsigned)TCn);
    unsigned q = unit / (unsigned)TCn;
    const int tr = (int)(q % (unsigned)TR);
    q /= (unsigned)TR;
    const int cb = (int)(q % (unsigned)nb), n = (int)(q / (unsigned)nb);
    const int c = cb * 64 + lane;
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;
    const unsigned OOB = 0x80000000u;

    // the coarse plane of this image as one buffer (its column and row indices are clamped, never out of range); x and y row by row
    i32x4 csrc;
    {
        const unsigned long long a = (unsigned long long)(reinterpret_cast<const char*>(coarse) + (size_t)n * Hc * Wc * pixc);
        csrc.x = (int)(unsigned)a;
        csrc.y = (int)(unsigned)(a >> 32) & 0xffff;
        csrc.z = Hc * Wc * pixc;
        csrc.w = 0x00020000;
    }
    const unsigned long long xbase = (unsigned long long)(reinterpret_cast<const char*>(x) + (size_t)n * H * W * pix);
    const unsigned long long ybase = (unsigned long long)(reinterpret_cast<char*>(y) + (size_t)n * H * W * pix);
    const unsigned cvo = (unsigned)(cc * CSZ);
    const int cb0 = NP * tc;
    int ck[NCR];                                                 // byte offsets of the coarse columns -2 .. NP + 1 of the tile, clamped into the plane
#pragma unroll
    for (int k = 0; k < NCR; ++k) {
        int col = cb0 - 2 + k;
        col = col < 0 ? 0 : (col > Wc - 1 ? Wc - 1 : col);
        ck[k] = col * pixc;
    }
    auto load_coarse = [&](uint32_t (&dst)[NCR], int i) {        // tile-local coarse row i, clamped into the plane (ATen's border rule)
        int ar = 7 * tr + i;
        ar = ar < 0 ? 0 : (ar > Hc - 1 ? Hc - 1 : ar);
        coarse_row_load<TC>(dst, cvo, csrc, ar * Wc * pixc, ck);
    };
    // the resized coarse row is zero outside the plane (the conv pads x + resize(coarse) with zeros): one mask per column pair
    float hm[NHP];
#pragma unroll
    for (int j2 = 0; j2 < NHP; ++j2) { const int col = TW * tc - 2 + 2 * j2; hm[j2] = (col >= 0 && col < W) ? 1.f : 0.f; }
    const f32x2 wq = MODE == 1 ? splat(0.f) : splat(0.25f), wt = MODE == 1 ? splat(1.f) : splat(0.75f);
    // H row: a coarse row resized horizontally to the columns -2 .. TW + 1 (pairs outside the image zeroed)
    auto build_H = [&](f32x2 (&Hs)[NHP], const uint32_t (&cr)[NCR]) {
        float cv[NCR];
#pragma unroll
        for (int k = 0; k < NCR; ++k) cv[k] = raw_f32<TC>(cr[k]);
        f32x2 P[6], Pq[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) { P[m] = f32x2{cv[2 * m], 2 * m + 1 < NCR ? cv[2 * m + 1 < NCR ? 2 * m + 1 : 0] : 0.f}; Pq[m] = P[m] * wq; }
        sfor<NHP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const f32x2 e = f32x2{(j & 1) ? Pq[j >> 1].y : Pq[j >> 1].x, (j & 1) ? Pq[(j >> 1) + 1].y : Pq[(j >> 1) + 1].x};
            const float mid = ((j + 1) & 1) ? P[(j + 1) >> 1].y : P[(j + 1) >> 1].x;
            Hs[j] = pfma(splat(mid), wt, e);
        });
#pragma unroll
        for (int j2 = 0; j2 < NHP; ++j2) Hs[j2] = Hs[j2] * splat(hm[j2]);
    };

    const unsigned voffM = (unsigned)((TW * tc) * pix + cc * ESZ);
    const unsigned voffL = voffM - 2u * (unsigned)pix;            // tile column 0: negative = out of range
    const unsigned voffR = voffM + (unsigned)TW * (unsigned)pix;
    auto load_row = [&](uint32_t (&raw)[NCOL], int r) {          // rows outside the plane: a valid row is loaded and not used
        int ar = 14 * tr + r;
        ar = ar < 0 ? 0 : (ar > H - 1 ? H - 1 : ar);
        xrow_load<TIO, PIXB>(raw, voffL, voffM, voffR, row_desc(xbase, ar, H, W * pix), 0, pix);
    };
    auto row_valid = [&](int r) -> bool { const int ar = 14 * tr + r; return ar >= 0 && ar < H; };       // uniform

    uint32_t raw[NR][NCOL];
    f32x2 Hh[2][NHP];
    f32x2 acc[5][NP];
    uint32_t craw[NCR];
    {
        // prologue, in the order Sched replays: coarse rows -2, -1, 0, then x rows -2, -1, then (compiler-counted) the taps
        uint32_t cm2[NCR], cm1[NCR];
        load_coarse(cm2, -2);
        load_coarse(cm1, -1);
        load_coarse(craw, 0);
        sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], -2 + decltype(rc)::value); });
        pin_coarse<S::cap(S::pending(0, 0))>(cm1);                // rows -2 and -1 have landed (memory operations complete in order)
        pin_coarse<S::cap(S::pending(0, 0))>(cm2);
        build_H(Hh[0], cm2);
        build_H(Hh[1], cm1);
    }
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 25 * C * 4, 0x00020000);
    Taps tf;
    load_taps(tf, wsrc, __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, has_bias ? C * 4 : 0, 0x00020000), 0, C, cc);      // zero records without a bias: the load returns 0
    // the taps land HERE, on every path: left to the compiler, their loads sink to the first use inside the conditional row bodies, and
    // its own count of outstanding loads (which knows nothing of the hand-issued ones) then drains the row prefetch in every iteration
#pragma unroll
    for (int u = 0; u < 5; ++u) { pin(tf.p[u][0]); pin(tf.p[u][1]); pin(tf.p[u][2]); }
    f32x2 bf = splat(tf.bias);
    pin(bf);                                                     // a real register pair: no half of it is ever borrowed from a row in flight
    const unsigned yoff = cvalid ? (unsigned)((TW * tc) * pix + c * ESZ) : OOB;

    sfor<NR>([&](auto rc) {
        constexpr int ri = decltype(rc)::value, t = ri - 2;
        if constexpr (ri + AHEAD < NR) load_row(raw[ri + AHEAD], t + AHEAD);
        // vertical source rows (tile origin is even): t even -> (t/2 - 1, t/2) weight 0.75 on the second; t odd -> ((t-1)/2, (t+1)/2), 0.25
        constexpr int te = (t + 2) & 1;
        constexpr int i0 = MODE == 1 ? ((t + 2) >> 1) - 1 : (te ? (t - 1) / 2 : t / 2 - 1);
        constexpr int i1 = MODE == 1 ? i0 : i0 + 1;
        constexpr float lam = MODE == 1 ? 0.f : (te ? 0.25f : 0.75f);
        if constexpr (S::is_build(ri)) {
            constexpr int ib = S::build_row(ri);
            static_assert(ib == (MODE == 0 ? i1 : i0), "the row built here is the one this iteration is the first to use");
            pin_coarse<S::cap(S::pending(1, ri))>(craw);
            build_H(Hh[(ib + 2) & 1], craw);
            if constexpr (ib + 1 <= S::c_last) load_coarse(craw, ib + 1);
        }
        xrow_pin<S::cap(S::pending(2, ri))>(raw[ri]);
        if (row_valid(t)) {
            f32x2 row[NHP], odd[NHP - 1];
#pragma unroll
            for (int k = 0; k < NHP; ++k) {
                const f32x2 xv = f32x2{raw_f32<TIO>(raw[ri][2 * k]), raw_f32<TIO>(raw[ri][2 * k + 1])};
                if (MODE == 1) row[k] = xv + Hh[(i0 + 2) & 1][k];
                else row[k] = pfma(splat(lam), Hh[(i1 + 2) & 1][k], pfma(splat(1.f - lam), Hh[(i0 + 2) & 1][k], xv));
            }
            // the two columns left of the image and right of it are zero padding of the conv INPUT: x read 0 there and H was zeroed
#pragma unroll
            for (int j = 0; j < NHP - 1; ++j) odd[j] = shift1(row[j], row[j + 1]);
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int o = t - u + 2;
                if (o < 0 || o > 13) continue;
                f32x2(&a)[NP] = acc[o % 5];
#pragma unroll
                for (int j = 0; j < NP; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), u == 0 ? bf : a[j]);
#pragma unroll
                for (int j = 0; j < NP; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
                for (int j = 0; j < NP; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
                for (int j = 0; j < NP; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
                for (int j = 0; j < NP; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
            }
        } else if constexpr (t + 2 >= 0 && t + 2 <= 13) {
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[(t + 2) % 5][j] = bf;
        }
        if constexpr (t - 2 >= 0 && t - 2 <= 13) {
            constexpr int o = t - 2;
            yrow_store<TIO, PIXB>(acc[o % 5], yoff, row_desc(ybase, 14 * tr + o, H, W * pix), 0, pix);   // rows past the plane, columns past the row: dropped, but issued (Sched counts them)
        }
#pragma unroll
        for (int o = 0; o < 14; ++o) if (o > t - 2 && o <= t + 2) pin(acc[o % 5]);
        pin(Hh[0]);
        pin(Hh[1]);
        CPT_FENCE;
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
//   k_down5_cpt   y = conv5, stride 2 (x)   -- one step of the down ladder (model/recnext.py:27-29) / RecAttn2d's `down` conv
//                                              (model/recattn.py:61) on any even plane whose width is a multiple of 14
// Pass 1 of k_recconv_cpt as a kernel of its own: a wave = 64 channels of one 14 x 14 input tile = 7 x 7 outputs, input rows -2 .. 14
// stream through registers, input-row stationary with the taps paired ((w0, w1), (w2, w3), (w4, 0): the even / odd partial sums of an
// output share a register pair and are added at the end), an output row of 7 pixels leaves as one statement.
constexpr int NR1 = 17;           // input rows of a tile: -2 .. 14

template <int AHEAD, int TW> struct SchedDown {
    static constexpr int pending(int ri_target)
    {
        int seq = 0, xend[NR1 + AHEAD + 1] = {};
        for (int r = 0; r < AHEAD; ++r) { seq += TW + 4; xend[r] = seq; }
        for (int ri = 0; ri < NR1; ++ri) {
            if (ri + AHEAD < NR1) { seq += TW + 4; xend[ri + AHEAD] = seq; }
            if (ri == ri_target) return seq - xend[ri];
            if (ri >= 4 && (ri & 1) == 0) seq += TW / 2;          // output row (ri - 4) / 2 leaves at the end of the iteration
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};

// one output row: 7 (8: 16-wide tiles) pixels, pitch `pix` bytes; vo out of range -> dropped (but issued)
#define DN_S(OP, d) OP " %[p" #d "], %[vo], %[rs], %[t] offen\n\t"
#define DN_N "s_add_i32 %[t], %[t], %[pix]\n\t"
template <typename TO> struct DownSt;
template <> struct DownSt<float> {
    static __device__ __forceinline__ void st(const float (&v)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        int t;
        asm volatile("s_add_i32 %[t], %[rb], 0\n\t" DN_S("buffer_store_dword", 0) DN_N DN_S("buffer_store_dword", 1) DN_N DN_S("buffer_store_dword", 2) DN_N
                     DN_S("buffer_store_dword", 3) DN_N DN_S("buffer_store_dword", 4) DN_N DN_S("buffer_store_dword", 5) DN_N DN_S("buffer_store_dword", 6)
                     : [t] "=&s"(t)
                     : [p0] "v"(v[0]), [p1] "v"(v[1]), [p2] "v"(v[2]), [p3] "v"(v[3]), [p4] "v"(v[4]), [p5] "v"(v[5]), [p6] "v"(v[6]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    }
    static __device__ __forceinline__ void st(const float (&v)[8], unsigned vo, i32x4 rs, int rb, int pix)
    {
        int t;
        asm volatile("s_add_i32 %[t], %[rb], 0\n\t" DN_S("buffer_store_dword", 0) DN_N DN_S("buffer_store_dword", 1) DN_N DN_S("buffer_store_dword", 2) DN_N
                     DN_S("buffer_store_dword", 3) DN_N DN_S("buffer_store_dword", 4) DN_N DN_S("buffer_store_dword", 5) DN_N DN_S("buffer_store_dword", 6) DN_N
                     DN_S("buffer_store_dword", 7)
                     : [t] "=&s"(t)
                     : [p0] "v"(v[0]), [p1] "v"(v[1]), [p2] "v"(v[2]), [p3] "v"(v[3]), [p4] "v"(v[4]), [p5] "v"(v[5]), [p6] "v"(v[6]), [p7] "v"(v[7]),
                       [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    }
};
template <typename T16> struct DownSt16 {
    static __device__ __forceinline__ void st(const float (&v)[7], unsigned vo, i32x4 rs, int rb, int pix)
    {
        uint32_t p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = pk16<T16>(v[2 * j], j < 3 ? v[2 * j + 1] : 0.f);
        int t;
        asm volatile("s_add_i32 %[t], %[rb], 0\n\t" DN_S("buffer_store_short", 0) DN_N DN_S("buffer_store_short_d16_hi", 0) DN_N DN_S("buffer_store_short", 1) DN_N
                     DN_S("buffer_store_short_d16_hi", 1) DN_N DN_S("buffer_store_short", 2) DN_N DN_S("buffer_store_short_d16_hi", 2) DN_N DN_S("buffer_store_short", 3)
                     : [t] "=&s"(t)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    }
    static __device__ __forceinline__ void st(const float (&v)[8], unsigned vo, i32x4 rs, int rb, int pix)
    {
        uint32_t p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = pk16<T16>(v[2 * j], v[2 * j + 1]);
        int t;
        asm volatile("s_add_i32 %[t], %[rb], 0\n\t" DN_S("buffer_store_short", 0) DN_N DN_S("buffer_store_short_d16_hi", 0) DN_N DN_S("buffer_store_short", 1) DN_N
                     DN_S("buffer_store_short_d16_hi", 1) DN_N DN_S("buffer_store_short", 2) DN_N DN_S("buffer_store_short_d16_hi", 2) DN_N DN_S("buffer_store_short", 3) DN_N
                     DN_S("buffer_store_short_d16_hi", 3)
                     : [t] "=&s"(t)
                     : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [vo] "v"(vo), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc", "memory");
    }
};
template <> struct DownSt<bf16_t> : DownSt16<bf16_t> {};
template <> struct DownSt<f16_t> : DownSt16<f16_t> {};

template <int PIXB, typename TIO, typename TO, int TW = 14>
__global__ __launch_bounds__(256, 2) void k_down5_cpt(const TIO* __restrict__ x, TO* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                                                      int N, int C, int H, int W, int has_bias)
{
    constexpr int ESZ = (int)sizeof(TIO), OSZ = (int)sizeof(TO), AHEAD = 3, NPO = TW / 2, NCOL = TW + 4, NHP = NCOL / 2;
    static_assert(TW == 14 || (TW == 16 && sizeof(TIO) == 2), "tile width");
    using S = SchedDown<AHEAD, TW>;
    const int nb = (C + 63) / 64, TR = (H + 13) / 14, TCn = (W + TW - 1) / TW, Ho = H / 2, Wo = W / 2;
    const int pix = PIXB ? PIXB : C * ESZ, pixo = C * OSZ;
    const unsigned total = (unsigned)N * nb * TR * TCn;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned upp = (unsigned)(TR * TCn);                    // tile-waves per (image, channel block) plane
    const unsigned wg = (upp & 3u) == 0 ? xcd_workgroup(blockIdx.x, upp >> 2, total / upp) : blockIdx.x;
    const unsigned unit = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (unit >= total) return;
    const int tc = (int)(unit % (unsigned)TCn);
    unsigned q = unit / (unsigned)TCn;
    const int tr = (int)(q % (unsigned)TR);
    q /= (unsigned)TR;
    const int cb = (int)(q % (unsigned)nb), n = (int)(q / (unsigned)nb);
    const int c = cb * 64 + lane;
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;
    const unsigned OOB = 0x80000000u;
    const unsigned long long xbase = (unsigned long long)(reinterpret_cast<const char*>(x) + (size_t)n * H * W * pix);
    const unsigned long long ybase = (unsigned long long)(reinterpret_cast<char*>(y) + (size_t)n * Ho * Wo * pixo);
    const unsigned voffM = (unsigned)((TW * tc) * pix + cc * ESZ);
    const unsigned voffL = voffM - 2u * (unsigned)pix;            // tile column 0: negative = out of range
    const unsigned voffR = voffM + (unsigned)TW * (unsigned)pix;
    auto load_row = [&](uint32_t (&raw)[NCOL], int r) {
        int ar = 14 * tr + r;
        ar = ar < 0 ? 0 : (ar > H - 1 ? H - 1 : ar);
        xrow_load<TIO, PIXB>(raw, voffL, voffM, voffR, row_desc(xbase, ar, H, W * pix), 0, pix);
    };
    auto row_valid = [&](int r) -> bool { const int ar = 14 * tr + r; return ar >= 0 && ar < H; };
    uint32_t raw[NR1][NCOL];
    sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], -2 + decltype(rc)::value); });
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 25 * C * 4, 0x00020000);
    Taps td;
    load_taps(td, wsrc, __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, has_bias ? C * 4 : 0, 0x00020000), 0, C, cc);      // zero records without a bias: the load returns 0
#pragma unroll
    for (int u = 0; u < 5; ++u) { pin(td.p[u][0]); pin(td.p[u][1]); pin(td.p[u][2]); }       // landed here, on every path (see k_upadd_cpt)
    f32x2 b0 = f32x2{td.bias, 0.f};
    pin(b0);
    const unsigned yoff = cvalid ? (unsigned)((NPO * tc) * pixo + c * OSZ) : OOB;
    f32x2 facc[3][NPO];
    sfor<NR1>([&](auto rc) {
        constexpr int ri = decltype(rc)::value, r = ri - 2;
        if constexpr (ri + AHEAD < NR1) load_row(raw[ri + AHEAD], r + AHEAD);
        xrow_pin<S::cap(S::pending(ri))>(raw[ri]);
        f32x2 xr[NHP];
#pragma unroll
        for (int k = 0; k < NHP; ++k) xr[k] = f32x2{raw_f32<TIO>(raw[ri][2 * k]), raw_f32<TIO>(raw[ri][2 * k + 1])};
        const bool rv = row_valid(r);
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            const int u = r - 2 * o + 2;
            if (u < 0 || u > 4) continue;
            f32x2(&a)[NPO] = facc[o % 3];
            if (rv) {
#pragma unroll
                for (int i = 0; i < NPO; ++i) a[i] = pfma(xr[i], td.p[u][0], u == 0 ? b0 : a[i]);
#pragma unroll
                for (int i = 0; i < NPO; ++i) a[i] = pfma(xr[i + 1], td.p[u][1], a[i]);
#pragma unroll
                for (int i = 0; i < NPO; ++i) a[i].x = fmaf(xr[i + 2].x, td.p[u][2].x, a[i].x);
            } else if (u == 0) {
#pragma unroll
                for (int i = 0; i < NPO; ++i) a[i] = b0;
            }
            if (u == 4) {
                float out[NPO];
#pragma unroll
                for (int i = 0; i < NPO; ++i) out[i] = a[i].x + a[i].y;
                DownSt<TO>::st(out, yoff, row_desc(ybase, 7 * tr + o, Ho, Wo * pixo), 0, pixo);
            }
        }
#pragma unroll
        for (int o = 0; o < 7; ++o) if (r - 2 * o + 2 >= 0 && r - 2 * o + 2 < 4) pin(facc[o % 3]);
        CPT_FENCE;
    });
}

template <typename TIO, typename TO>
static hipError_t launch_down(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, hipStream_t s)
{
    const int tw = tile_width(W, std::is_same<TIO, bf16_t>::value ? 1 : 0);
    const long long units = (long long)N * ((C + 63) / 64) * ((H + 13) / 14) * ((W + tw - 1) / tw);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    const int hb = b != nullptr;
    const int pixb = C * (int)sizeof(TIO);
#define RCX_GO(PB, TW) hipLaunchKernelGGL((k_down5_cpt<PB, TIO, TO, TW>), grid, block, 0, s, (const TIO*)x, (TO*)y, w, b, N, C, H, W, hb)
    if constexpr (std::is_same<TIO, bf16_t>::value) {
        if (tw == 16) {
            if (pixb == 128) RCX_GO(128, 16);
            else if (pixb == 256) RCX_GO(256, 16);
            else RCX_GO(0, 16);
            return hipGetLastError();
        }
    }
    if constexpr (sizeof(TIO) == 2) {
        if (pixb == 128) { RCX_GO(128, 14); return hipGetLastError(); }
        if (pixb == 256) { RCX_GO(256, 14); return hipGetLastError(); }
    }
    RCX_GO(0, 14);
#undef RCX_GO
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
//   k_down7m2_cpt   y = conv 7x7, stride 2, channel multiplier 2 (x) + bias   -- Downsample.token_mixer (+ its folded BatchNorm),
//                                                                                 model/recnext.py:165-166, on ANY even plane
// A lane owns one OUTPUT channel o (input channel o / 2: the two lanes of a pair load the same element, a wave reads 64 contiguous
// bytes per pixel and writes 128), a wave = 64 output channels of one 14 x 14 input tile = 7 x 7 outputs; input rows -3 .. 15 stream
// through registers, taps paired ((w0,w1), (w2,w3), (w4,w5), (w6,0)), four output rows in flight.
// Rows are addressed through PER-ROW buffer descriptors (base = the row, num_records = its bytes): a column left of the plane or right
// of it is out of range and reads 0, a store past the row's end is dropped -- no edge flags, and the last tile column may be ragged, so
// any even width works (the two kernels above carry image descriptors and edge flags; this is the form they should take next).
#define DN7_ROW_IMM(OP)                                                                                                              \
    "s_add_i32 %[t], %[rb], 0\n\t"                                                                                                  \
    CPT_LI(OP, 0, "vl", "t", 0) CPT_LI(OP, 1, "vl", "t", 1) CPT_LI(OP, 2, "vl", "t", 2)                                              \
    CPT_LI(OP, 3, "vm", "t", 0) CPT_LI(OP, 4, "vm", "t", 1) CPT_LI(OP, 5, "vm", "t", 2) CPT_LI(OP, 6, "vm", "t", 3)                  \
    CPT_LI(OP, 7, "vm", "t", 4) CPT_LI(OP, 8, "vm", "t", 5) CPT_LI(OP, 9, "vm", "t", 6) CPT_LI(OP, 10, "vm", "t", 7)                 \
    CPT_LI(OP, 11, "vm", "t", 8) CPT_LI(OP, 12, "vm", "t", 9) CPT_LI(OP, 13, "vm", "t", 10) CPT_LI(OP, 14, "vm", "t", 11)            \
    CPT_LI(OP, 15, "vm", "t", 12) CPT_LI(OP, 16, "vm", "t", 13)                                                                      \
    CPT_LI(OP, 17, "vr", "t", 0) CPT_LI(OP, 18, "vr", "t", 1) CPT_LI(OP, 19, "vr", "t", 2)
#define DN7_N(OP, d, V) "s_add_i32 %[t2], %[t2], %[pix]\n\t" CPT_LG(OP, d, V, "t2")
#define DN7_ROW_GEN(OP)                                                                                                              \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 0, "vl", "t2") DN7_N(OP, 1, "vl") DN7_N(OP, 2, "vl")                                    \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 17, "vr", "t2") DN7_N(OP, 18, "vr") DN7_N(OP, 19, "vr")                                 \
    "s_add_i32 %[t2], %[rb], 0\n\t" CPT_LG(OP, 3, "vm", "t2") DN7_N(OP, 4, "vm") DN7_N(OP, 5, "vm") DN7_N(OP, 6, "vm") DN7_N(OP, 7, "vm")   \
    DN7_N(OP, 8, "vm") DN7_N(OP, 9, "vm") DN7_N(OP, 10, "vm") DN7_N(OP, 11, "vm") DN7_N(OP, 12, "vm") DN7_N(OP, 13, "vm") DN7_N(OP, 14, "vm") \
    DN7_N(OP, 15, "vm") DN7_N(OP, 16, "vm")
// columns -3, -2, -1 (vl), 0 .. 13 (vm), 14, 15, 16 (vr)
template <typename TIO, int PIXB>
__device__ __forceinline__ void row_load7(uint32_t (&v)[20], unsigned vl, unsigned vm, unsigned vr, i32x4 rs, int rb, int pix)
{
    int t, t2;
    if constexpr (PIXB > 0 && PIXB * 13 <= 4095) {
        (void)pix; (void)t2;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(DN7_ROW_IMM(CPT_LDH) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(DN7_ROW_IMM(CPT_LD16) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
        else
            asm volatile(DN7_ROW_IMM(CPT_LD32) : CPT_OUT20(v), [t] "=&s"(t) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pb] "n"(PIXB) : "scc");
    } else {
        (void)t;
        if constexpr (std::is_same<TIO, f16_t>::value)
            asm volatile(DN7_ROW_GEN(CPT_LDH) : CPT_OUT20(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else if constexpr (sizeof(TIO) == 2)
            asm volatile(DN7_ROW_GEN(CPT_LD16) : CPT_OUT20(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
        else
            asm volatile(DN7_ROW_GEN(CPT_LD32) : CPT_OUT20(v), [t2] "=&s"(t2) : [vl] "v"(vl), [vm] "v"(vm), [vr] "v"(vr), [rs] "s"(rs), [rb] "s"(rb), [pix] "s"(pix) : "scc");
    }
}

constexpr int NR7 = 19;           // input rows of a tile: -3 .. 15
template <int AHEAD> struct SchedDown7 {
    static constexpr int pending(int ri_target)
    {
        int seq = 0, xend[NR7 + AHEAD + 1] = {};
        for (int r = 0; r < AHEAD; ++r) { seq += 20; xend[r] = seq; }
        for (int ri = 0; ri < NR7; ++ri) {
            if (ri + AHEAD < NR7) { seq += 20; xend[ri + AHEAD] = seq; }
            if (ri == ri_target) return seq - xend[ri];
            if (ri >= 6 && (ri & 1) == 0) seq += 7;               // output row (ri - 6) / 2 leaves at the end of the iteration
        }
        return 0;
    }
    static constexpr int cap(int v) { return v > 63 ? 63 : v; }
};

template <int PIXB, typename TIO>
__global__ __launch_bounds__(256, 2) void k_down7m2_cpt(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                                                        int N, int Cin, int H, int W, int has_bias)
{
    constexpr int ESZ = (int)sizeof(TIO), AHEAD = 2;
    using S = SchedDown7<AHEAD>;
    const int Co = 2 * Cin, nb = (Co + 63) / 64, TR = (H + 13) / 14, TCn = (W + 13) / 14, Ho = H / 2, Wo = W / 2;
    const int pix = PIXB ? PIXB : Cin * ESZ, pixo = Co * ESZ;
    const unsigned total = (unsigned)N * nb * TR * TCn;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned upp = (unsigned)(TR * TCn);                    // tile-waves per (image, channel block) plane
    const unsigned wg = (upp & 3u) == 0 ? xcd_workgroup(blockIdx.x, upp >> 2, total / upp) : blockIdx.x;
    const unsigned unit = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (unit >= total) return;
    const int tc = (int)(unit % (unsigned)TCn);
    unsigned q = unit / (unsigned)TCn;
    const int tr = (int)(q % (unsigned)TR);
    q /= (unsigned)TR;
    const int ob = (int)(q % (unsigned)nb), n = (int)(q / (unsigned)nb);
    const int o = ob * 64 + lane;
    const bool ovalid = o < Co;
    const int oo = ovalid ? o : Co - 1;
    const unsigned OOB = 0x80000000u;
    const unsigned long long xbase = (unsigned long long)(reinterpret_cast<const char*>(x) + (size_t)n * H * W * pix);
    const unsigned long long ybase = (unsigned long long)(reinterpret_cast<char*>(y) + (size_t)n * Ho * Wo * pixo);
    const unsigned voffM = (unsigned)((14 * tc) * pix + (oo >> 1) * ESZ);
    const unsigned voffL = voffM - 3u * (unsigned)pix;            // tile column 0: negative = out of range
    const unsigned voffR = voffM + 14u * (unsigned)pix;
    auto load_row = [&](uint32_t (&raw)[20], int r) { row_load7<TIO, PIXB>(raw, voffL, voffM, voffR, row_desc(xbase, 14 * tr + r, H, W * pix), 0, pix); };
    uint32_t raw[NR7][20];
    sfor<AHEAD>([&](auto rc) { load_row(raw[decltype(rc)::value], -3 + decltype(rc)::value); });
    // the 49 taps of this output channel as four register pairs per tap row, landed before the loop on every path (see k_upadd_cpt)
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 49 * Co * 4, 0x00020000);
    f32x2 tp[7][4];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
#pragma unroll
        for (int v = 0; v < 7; ++v) {
            const float wv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wsrc, oo * 4, (u * 7 + v) * Co * 4, 0));
            if (v & 1) tp[u][v >> 1].y = wv;
            else tp[u][v >> 1].x = wv;
        }
        tp[u][3].y = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) { pin(tp[u][0]); pin(tp[u][1]); pin(tp[u][2]); pin(tp[u][3]); }
    f32x2 b0 = f32x2{has_bias ? bias[oo] : 0.f, 0.f};
    pin(b0);
    const unsigned yoff = ovalid ? (unsigned)((7 * tc) * pixo + o * ESZ) : OOB;
    f32x2 facc[4][7];
    sfor<NR7>([&](auto rc) {
        constexpr int ri = decltype(rc)::value, r = ri - 3;
        if constexpr (ri + AHEAD < NR7) load_row(raw[ri + AHEAD], r + AHEAD);
        pin_row20<S::cap(S::pending(ri))>(raw[ri]);
        f32x2 xr[10];                                             // (column 2k - 3, column 2k - 2)
#pragma unroll
        for (int k = 0; k < 10; ++k) xr[k] = f32x2{raw_f32<TIO>(raw[ri][2 * k]), raw_f32<TIO>(raw[ri][2 * k + 1])};
        // a row outside the plane was read through an empty descriptor: zeros, so it adds nothing and needs no branch
#pragma unroll
        for (int ot = 0; ot < 7; ++ot) {
            const int u = r - 2 * ot + 3;
            if (u < 0 || u > 6) continue;
            f32x2(&a)[7] = facc[ot % 4];
#pragma unroll
            for (int i = 0; i < 7; ++i) a[i] = pfma(xr[i], tp[u][0], u == 0 ? b0 : a[i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) a[i] = pfma(xr[i + 1], tp[u][1], a[i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) a[i] = pfma(xr[i + 2], tp[u][2], a[i]);
#pragma unroll
            for (int i = 0; i < 7; ++i) a[i].x = fmaf(xr[i + 3].x, tp[u][3].x, a[i].x);
            if (u == 6) {
                float out[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) out[i] = a[i].x + a[i].y;
                DownSt<TIO>::st(out, yoff, row_desc(ybase, 7 * tr + ot, Ho, Wo * pixo), 0, pixo);
            }
        }
#pragma unroll
        for (int ot = 0; ot < 7; ++ot) if (r - 2 * ot + 3 >= 0 && r - 2 * ot + 3 < 6) pin(facc[ot % 4]);
        CPT_FENCE;
    });
}

template <typename TIO>
static hipError_t launch_down7(const void* x, void* y, const float* w, const float* b, int N, int Cin, int H, int W, hipStream_t s)
{
    const long long units = (long long)N * ((2 * Cin + 63) / 64) * ((H + 13) / 14) * ((W + 13) / 14);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    const int hb = b != nullptr;
    const int pixb = Cin * (int)sizeof(TIO);
#define RCX_GO(PB) hipLaunchKernelGGL((k_down7m2_cpt<PB, TIO>), grid, block, 0, s, (const TIO*)x, (TIO*)y, w, b, N, Cin, H, W, hb)
    if constexpr (sizeof(TIO) == 2) {
        if (pixb == 128) { RCX_GO(128); return hipGetLastError(); }
        if (pixb == 256) { RCX_GO(256); return hipGetLastError(); }
    }
    RCX_GO(0);
#undef RCX_GO
    return hipGetLastError();
}

static inline bool enabled()
{
    const char* v = rcx::opt::value(rcx::opt::UPADD_CPT);
    const char* l = rcx::opt::value(rcx::opt::LANES);
    return !(v && *v == '0') && !(l && *l == '0');
}

template <int MODE, typename TIO, typename TC>
static hipError_t launch(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, int H, int W, hipStream_t s)
{
    const int tw = tile_width(W, std::is_same<TIO, bf16_t>::value ? 1 : 0);
    const long long units = (long long)N * ((C + 63) / 64) * ((H + 13) / 14) * ((W + tw - 1) / tw);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    const int hb = b != nullptr;
    const int pixb = C * (int)sizeof(TIO);
#define RCX_GO(PB, TW) hipLaunchKernelGGL((k_upadd_cpt<MODE, PB, TIO, TC, TW>), grid, block, 0, s, (const TIO*)x, (const TC*)coarse, (TIO*)y, w, b, N, C, H, W, hb)
    if constexpr (std::is_same<TIO, bf16_t>::value) {
        if (tw == 16) {
            if (pixb == 128) RCX_GO(128, 16);
            else if (pixb == 256) RCX_GO(256, 16);
            else RCX_GO(0, 16);
            return hipGetLastError();
        }
    }
    if constexpr (sizeof(TIO) == 2) {
        if (pixb == 128) { RCX_GO(128, 14); return hipGetLastError(); }
        if (pixb == 256) { RCX_GO(256, 14); return hipGetLastError(); }
    }
    RCX_GO(0, 14);
#undef RCX_GO
    return hipGetLastError();
}

}  // namespace upcpt

// y = conv5(x + resize2x(coarse)): any plane that is exactly twice its coarse plane, at least 28 x 28; the 14 x 14 plane has its own
// whole-plane kernel (rcx_cpl14.hip)
bool upadd_cpt_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt)
{
    if (!upcpt::enabled() || k != 5 || out_dt != x_dt || x_dt < 0 || x_dt > 2 || !(c_dt == x_dt || c_dt == 0)) return false;
    const int tw = upcpt::tile_width(W, x_dt);
    if (N < 1 || C < 1 || Hc * 2 != H || Wc * 2 != W || H < 28 || W < 28) return false;
    const long long img = (long long)H * W * C * 4;
    const long long units = (long long)N * ((C + 63) / 64) * ((H + 13) / 14) * ((W + tw - 1) / tw);
    return img < (1ll << 31) && units < (1ll << 31);
}

int upadd_cpt_describe(int N, int C, int H, int W, int mode, int x_dt, char* buf, int len)
{
    const int pixb = x_dt != 0 && (C == 64 || C == 128) ? C * 2 : 0, tw = upcpt::tile_width(W, x_dt);
    return snprintf(buf, len, "upadd_cpt(k_upadd_cpt<%d, %d>,tw=%d,cb=64,nt=256,tiles=%lld)", mode, pixb, tw,
                    (long long)N * ((C + 63) / 64) * ((H + 13) / 14) * ((W + tw - 1) / tw));
}

hipError_t upadd_cpt(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, int H, int W, int mode,
                     int x_dt, int c_dt, hipStream_t s)
{
#define RCX_UP(TX, TC) (mode == 1 ? upcpt::launch<1, TX, TC>(x, coarse, y, w, b, N, C, H, W, s) : upcpt::launch<0, TX, TC>(x, coarse, y, w, b, N, C, H, W, s))
    if (x_dt == 1) return c_dt == 1 ? RCX_UP(bf16_t, bf16_t) : RCX_UP(bf16_t, float);
    if (x_dt == 2) return c_dt == 2 ? RCX_UP(f16_t, f16_t) : RCX_UP(f16_t, float);
    return RCX_UP(float, float);
#undef RCX_UP
}

// y = conv5 stride 2 (x): any even plane of at least 28 x 28
bool down5_cpt_applicable(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt)
{
    if (!upcpt::enabled() || k != 5 || stride != 2 || in_dt < 0 || in_dt > 2 || !(out_dt == in_dt || out_dt == 0)) return false;
    const int tw = upcpt::tile_width(W, in_dt);
    if (N < 1 || C < 1 || (H & 1) || (W & 1) || H < 28 || W < 28) return false;
    const long long img = (long long)H * W * C * 4;
    const long long units = (long long)N * ((C + 63) / 64) * ((H + 13) / 14) * ((W + tw - 1) / tw);
    return img < (1ll << 31) && units < (1ll << 31);
}

hipError_t down5_cpt(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int in_dt, int out_dt, hipStream_t s)
{
    if (in_dt == 1) return out_dt == 1 ? upcpt::launch_down<bf16_t, bf16_t>(x, y, w, b, N, C, H, W, s) : upcpt::launch_down<bf16_t, float>(x, y, w, b, N, C, H, W, s);
    if (in_dt == 2) return out_dt == 2 ? upcpt::launch_down<f16_t, f16_t>(x, y, w, b, N, C, H, W, s) : upcpt::launch_down<f16_t, float>(x, y, w, b, N, C, H, W, s);
    return upcpt::launch_down<float, float>(x, y, w, b, N, C, H, W, s);
}

// y = conv7 stride 2, channel multiplier 2 (Downsample.token_mixer): any even plane of at least 14 x 14
bool down7m2_cpt_applicable(int N, int Cin, int H, int W, int k, int stride, int dtype)
{
    if (!upcpt::enabled() || k != 7 || stride != 2 || dtype < 0 || dtype > 2) return false;
    if (N < 1 || Cin < 1 || (H & 1) || (W & 1) || H < 14 || W < 14) return false;
    const long long img = (long long)H * W * Cin * 4;
    const long long units = (long long)N * ((2 * Cin + 63) / 64) * ((H + 13) / 14) * ((W + 13) / 14);
    return img < (1ll << 31) && units < (1ll << 31);
}

hipError_t down7m2_cpt(const void* x, void* y, const float* w, const float* b, int N, int Cin, int H, int W, int dtype, hipStream_t s)
{
    if (dtype == 1) return upcpt::launch_down7<bf16_t>(x, y, w, b, N, Cin, H, W, s);
    if (dtype == 2) return upcpt::launch_down7<f16_t>(x, y, w, b, N, Cin, H, W, s);
    return upcpt::launch_down7<float>(x, y, w, b, N, Cin, H, W, s);
}

}  // namespace rcx
