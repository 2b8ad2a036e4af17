// RecAttn2d's coarse level in ONE launch for the short sequences (14x14 and 7x7 planes: 49 / 16 tokens; 15 of RecNeXt-A3's 21 units): the
// grouped 1x1 `qk` projection, the activation, k v^T, the normaliser, q kv and + pe (model/recattn.py:21-28 LinearAttention1, :44-51
// LinearAttention2 -- the same function) -- everything between the stride-2 conv and the final conv(x + resize(.)) of RecAttn2d.forward
// (:61-67).  It replaces two library GEMMs, their bias adds and rcx_linear_attention_pe_fwd, and the q / k tensors never exist in memory.
//
// ONE WORKGROUP = one image, ONE WAVE = one head: the image's d is staged in LDS once (one barrier), after which the waves share nothing.
// Head dimension 32 (RecNeXt-A3 / A4) or, padded to 32 inside the kernel, 4 .. 28 in steps of 4 (A0 / A1 / A2: 20 / 24 / 28; see `wfrag`); at most 64 tokens, at most 16 heads.
//   projection  q[t][c] = sum_ci d[t][ci] Wq[c][ci] + bq[c] over the FIRST half of d's channels, k likewise over the second half (groups = 2):
//               v_mfma_f32_32x32x16_bf16, A = d rows (float32 from the LDS image, rounded to bf16 in registers), B = the head's 32 rows of the bf16
//               weight pack, float32 accumulation -- C / 32 k-steps per 32-token tile.  The accumulator tile has the CHANNEL on the lane and
//               the tokens in its registers, which is exactly the A operand k^T of the next product;
//   kv          kv[c1][c2] = (1/n) sum_t k[t][c1] v[t][c2]: A = k straight from the accumulator registers (registers 8s .. 8s+7 = k-step s: the
//               token order they imply is the order the v fragment is gathered in), B = v = the head's 32 channels of d (LDS, float32 -> bf16);
//   out         out[t][c2] = sum_c1 q[t][c1] kv[c1][c2]: B = kv straight from ITS accumulator registers, A = q through a wave-private
//               float32 image in LDS (the one transpose), read in the permuted c1 order the kv registers imply; the normaliser
//               q . kbar + 1e-6 in float32 from the same image;
//   + pe        the depthwise 3x3 of d (float32, from the LDS image) where it is added; a is stored as float32.
// Numerics: the MFMA operands are bf16 (d, the weights, k, v, q, kv), every accumulation, the activation, the normaliser and pe are float32.
// That form holds north_star's flat 1e-2 for 16-bit activations (tests/test_recconv_gpu.py: the recattn goldens and A3 stages 2 / 3 run it;
// profiles/archive/r04_recattn_qk_gemm_operands.txt measured the projection with bf16 operands alone); float32 activations keep the float32 GEMMs and
// rcx_linear_attention_pe_fwd (their bar is 1e-3).
#include "rcx_common.h"
#include "rcx_launch.h"
#include "rcx_opts.h"
#include "rcx_cpl14_pieces.h"          // the packed-FMA row pieces of the channel-per-lane kernels: the whole unit's final conv (round 5)

namespace rcx {
namespace qkc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// eight float32 (two 16-byte LDS reads kept WHOLE: converted element by element, the compiler re-splits the reads into 12 + 4 + 4-byte pieces whose
// row-strided 4-byte parts hit the same bank four ways: SQ_LDS_BANK_CONFLICT 0.40 of the LDS cycles, profiles/archive/r04_sq_wave_states.txt) -> a bf16 operand
__device__ __forceinline__ bf16x8 to_bf16x8(f32x4q lo, f32x4q hi)
{
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- head dimensions below 32 (round 5: RecNeXt-A0 / A1 / A2, 20 / 24 / 28 channels per head; model/recattn.py:382-396).  Everything INSIDE the kernels stays
// 32 wide per head: the LDS images of d, a, the pe taps and the biases are PADDED (head hd at columns 32 hd .. 32 hd + D - 1, the rest zeros -- or -100 for the
// q / k biases: elu1 of it is 0, so a padded channel neither projects nor normalises), the weights of a padded output row or input column are zeros.  Only the
// faces to memory know the compact layout (C = D heads channels per token): the staging of d, x, the small arrays, the weight fragments and the stores of the
// attention output.  D is a multiple of 4 (16-byte groups never straddle a head) and a q / k half is whole heads (heads even).
typedef unsigned u32x2q __attribute__((ext_vector_type(2)));
constexpr float QK_PAD_BIAS = -100.f;
// the 8 bf16 weights of one output row (row = its first compact input) for the padded input columns pcol .. pcol + 7 of the row's half (pcol % 8 == 0)
template <bool PAD>
__device__ __forceinline__ u32x4q wfrag(const bf16_t* row, bool rvalid, int pcol, int D)
{
    if constexpr (!PAD) return rvalid ? *reinterpret_cast<const u32x4q*>(row + pcol) : u32x4q{0u, 0u, 0u, 0u};
    const int off = pcol & 31, ci = (pcol >> 5) * D + off;
    // (unconditional loads from an address that always exists, then a select: a branch per fragment otherwise)
    const bool vl = rvalid && off < D, vh = rvalid && off + 4 < D;
    const u32x2q lo = *reinterpret_cast<const u32x2q*>(row + (vl ? ci : 0)), hi = *reinterpret_cast<const u32x2q*>(row + (vh ? ci + 4 : 0));
    return u32x4q{vl ? lo.x : 0u, vl ? lo.y : 0u, vh ? hi.x : 0u, vh ? hi.y : 0u};
}
constexpr int DPAD = 4;           // floats added to a row of the shared d image (rows of C + 4: the 32 token rows of a fragment read fall on different banks)

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : __expf(x); }
__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wave execute in order; this only keeps the compiler from moving accesses across the hand-off
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// token of accumulator register i (0..15) of lane half h in a 32-row tile
__device__ __forceinline__ constexpr int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }


// rows [row0, row0 + nrows) of an image's d (float32, C channels) -> LDS rows of DROW floats, 16 bytes per lane and UN requests in flight per lane
// (one request per iteration costs a memory round trip per 16 bytes: 8 round trips for a 49 x 256 image).  Rows outside the image read 0.
template <int UN = 8>
__device__ __forceinline__ void stage_rows(float* Ld, __amdgpu_buffer_rsrc_t dsrc, int row0, int nrows, int Cg, int D, int heads, int DROW, int nthr)
{
    // Cg = D heads channels per token in memory; the LDS row has head hd at columns 32 hd .. (D == 32: the same thing)
    const int cq = Cg / 4, chunks = nrows * cq;
    for (int i0 = threadIdx.x; i0 < chunks; i0 += UN * nthr) {
        u32x4q v[UN];
        int dst[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * nthr, t = i / cq, q4 = i - t * cq;
            const int hdq = D == 32 ? 0 : (4 * q4) / D, col = D == 32 ? 4 * q4 : 32 * hdq + (4 * q4 - hdq * D);
            dst[u] = i < chunks ? t * DROW + col : -1;
            v[u] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(dsrc, i < chunks ? ((row0 + t) * Cg + q4 * 4) * 4 : -16, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (dst[u] >= 0) *reinterpret_cast<u32x4q*>(Ld + dst[u]) = v[u];
    }
    if (D != 32) {                                            // the padding columns of every row: zeros
        const int pq = (32 - D) / 4, per = heads * pq;
        for (int i = threadIdx.x; i < nrows * per; i += nthr) {
            const int t = i / per, e = i - t * per, hdq = e / pq, q = e - hdq * pq;
            *reinterpret_cast<float4*>(Ld + t * DROW + 32 * hdq + D + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// LONG sequences (the 56x56 and 28x28 stages: 784 / 196 tokens, 2 / 4 heads; any token count): the same function in TWO launches, because k v^T sums
// over every token of the image before the first output can be formed and an image's d (200 KB of float32 at 784 x 64) is not LDS-sized:
//   k_recattn_kv   grid (G, B): a workgroup projects k for its range of 32-token tiles (a wave = (head, split)), accumulates k^T v on the matrix cores
//                  and leaves ONE partial 32 x 32 sum + column sums per (image, range, head) in the workspace (fixed order: deterministic);
//   k_recattn_out  grid (G, B): stages the range's rows of d + one plane row and one token either side in LDS (the 3x3 of pe), sums the G partials,
//                  projects q^T = Wq d^T (the OPERANDS SWAPPED against the short kernel: the accumulator then has the TOKEN on the lane and c1 in the
//                  registers, which is the B operand of out^T = kv^T q^T as it stands, kv^T being the kv accumulator as it stands: no transpose
//                  through LDS), the normaliser (lane partial + its partner lane), pe as float4s of the lane's token, 16-byte stores.
// A lane of out^T holds channels 8 g + 4 h + (0..3), g = 0..3, of its token: registers 4 g .. 4 g + 3.

// the pe taps (9 x C), the pe bias and the q biases -> Lw[0, 11 C): 16 bytes per lane, every request issued before the first store (the plain loops
// were four to five dependent L2 round trips).  9 C / 4 <= 2 NTHR and C / 4 <= NTHR for every instantiation (C = 32 KS, NTHR >= 64 KS or 512).
template <int C, int NTHR, bool PAD>
__device__ __forceinline__ void stage_small(float* Lw, const float* __restrict__ wpe, const float* __restrict__ bpe, const float* __restrict__ bqk, int D)
{
    if constexpr (!PAD) {
        static_assert(9 * C / 4 <= 3 * NTHR && C / 4 <= NTHR, "stage_small: at most three requests per lane");
        const int i = threadIdx.x;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* w4 = reinterpret_cast<const float4*>(wpe);
        float4 a0 = z, a1 = z, a2 = z, b0 = z, c0 = z;      // (`cond ? p[i] : z` selects between ADDRESSES and puts z in scratch)
        if (i < 9 * C / 4) a0 = w4[i];
        if (i + NTHR < 9 * C / 4) a1 = w4[i + NTHR];
        if constexpr (9 * C / 4 > 2 * NTHR) { if (i + 2 * NTHR < 9 * C / 4) a2 = w4[i + 2 * NTHR]; }          // (two heads per wave: NTHR = C)
        if (bpe && i < C / 4) b0 = reinterpret_cast<const float4*>(bpe)[i];
        if (i < C / 4) c0 = reinterpret_cast<const float4*>(bqk)[i];
        float4* L4 = reinterpret_cast<float4*>(Lw);
        if (i < 9 * C / 4) L4[i] = a0;
        if (i + NTHR < 9 * C / 4) L4[i + NTHR] = a1;
        if constexpr (9 * C / 4 > 2 * NTHR) { if (i + 2 * NTHR < 9 * C / 4) L4[i + 2 * NTHR] = a2; }
        if (i < C / 4) { L4[9 * C / 4 + i] = b0; L4[10 * C / 4 + i] = c0; }
    } else {
        // C = 32 heads (padded); memory holds D heads channels per row: group i of the 11 C / 4 is row i / (C / 4) (9 tap rows, the pe bias, the q biases),
        // padded column 4 (i % (C / 4)).  One or two groups per lane, both requests issued before the first store; padding = 0 (-100 for the q biases).
        static_assert(11 * C / 4 <= 3 * NTHR, "stage_small: at most three requests per lane");
        const int Cg = (C / 32) * D;
        f32x4q v0, v1, v2;
        auto fetch = [&](int i, f32x4q& v) {
            const int row = i / (C / 4), cp = 4 * (i - row * (C / 4)), c = (cp >> 5) * D + (cp & 31);
            const float pad = row == 10 ? QK_PAD_BIAS : 0.f;
            v = f32x4q{pad, pad, pad, pad};
            const float* src = row < 9 ? wpe + (size_t)row * Cg + c : (row == 9 ? bpe + c : bqk + c);
            if (i < 11 * C / 4 && (cp & 31) < D && (row != 9 || bpe != nullptr)) v = *reinterpret_cast<const f32x4q*>(src);
        };
        fetch(threadIdx.x, v0);
        fetch(threadIdx.x + NTHR, v1);
        if constexpr (11 * C / 4 > 2 * NTHR) fetch(threadIdx.x + 2 * NTHR, v2);
        f32x4q* L4 = reinterpret_cast<f32x4q*>(Lw);
        if (threadIdx.x < 11 * C / 4) L4[threadIdx.x] = v0;
        if (threadIdx.x + NTHR < 11 * C / 4) L4[threadIdx.x + NTHR] = v1;
        if constexpr (11 * C / 4 > 2 * NTHR) { if (threadIdx.x + 2 * NTHR < 11 * C / 4) L4[threadIdx.x + 2 * NTHR] = v2; }
    }
}

// The end of a 32-token tile once q^T is accumulated (aq: rows c1 in the registers, column = the lane's token t): bias + activation, the normaliser
// (the lane's 16 channels + its partner lane's), out^T = kv^T q^T, pe as float4s of the lane's token and 16-byte stores.  drow = the token's row of
// the LDS image, at the lane's first channel (rows above / below the plane hold zeros: they are outside the buffer the image was staged from), zrow = a row of zeros for the taps
// that would wrap to the neighbouring plane row, Lwc / outp already point at the lane's first channel (head * 32 + 4 h).
// ostr = floats between tokens at outp (the compact C in memory; the padded row when the caller keeps a in LDS), dvalid = channels of this head that exist
// past the lane's first one (D - 4 h; 32: all four groups): a group of four is stored when its first channel exists
template <int C, bool MASK>
__device__ __forceinline__ void out_epilogue(const f32x16& aq, bf16x8 kv0, bf16x8 kv1, const float* Lbq, const float* Lkb, const float4* drow,
                                             const float4* zrow, const float4* Lwc, int t, int n, int Wp, int h, float* outp, int ooff, int ostr, int dvalid)
{
    // MASK (head dimension < 32): outp + ostr n floats is the image's output; a group past the head's D channels goes to an out-of-range buffer offset, which the
    // hardware drops (a branch around the store cost the tile loop 35 registers and spilled)
    __amdgpu_buffer_rsrc_t osrc;
    if constexpr (MASK) osrc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, n * ostr * 4, 0x00020000);
    constexpr int DROW4 = (C + DPAD) / 4;
    float dpart = 0.f;
    bf16x8 q0, q1;
    int hop = 4 * h;                               // opaque: or the 32 biases / kbar values of a lane are hoisted out of the caller's tile loop into registers
    asm volatile("" : "+v"(hop));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float qq = elu1(aq[i] + Lbq[acc_row(i, 0) + hop]);
        dpart = fmaf(qq, Lkb[acc_row(i, 0) + hop], dpart);
        if (i < 8) q0[i] = (__bf16)qq; else q1[i - 8] = (__bf16)qq;
    }
    const float dn = dpart + __shfl_xor(dpart, 32) + 1e-6f;
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv0, q0, o, 0, 0, 0);                  // out^T[c2][t] = sum_c1 kv[c1][c2] q^T[c1][t]
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kv1, q1, o, 0, 0, 0);
    if (t < n) {
        const float rdn = 1.f / dn;
        const int y = t / Wp, x = t - y * Wp;
        const float4* nb[9];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const bool wrap = (dx < 0 && x == 0) || (dx > 0 && x == Wp - 1);
                nb[(dy + 1) * 3 + dx + 1] = wrap ? zrow : drow + (dy * Wp + dx) * DROW4;
            }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4q* Lwv = reinterpret_cast<const f32x4q*>(Lwc);
            f32x4q pe = Lwv[(9 * C) / 4 + 2 * gq];
#pragma unroll
            for (int j = 0; j < 9; ++j) pe = __builtin_elementwise_fma(Lwv[(j * C) / 4 + 2 * gq], reinterpret_cast<const f32x4q*>(nb[j])[2 * gq], pe);
            const f32x4q ov = {o[4 * gq + 0], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]};
            const f32x4q res = __builtin_elementwise_fma(ov, f32x4q{rdn, rdn, rdn, rdn}, pe);
            if constexpr (MASK) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, res), osrc, 8 * gq < dvalid ? (t * ostr + ooff + 8 * gq) * 4 : -16, 0, 0);
            else *reinterpret_cast<f32x4q*>(outp + (size_t)t * ostr + ooff + 8 * gq) = res;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// SHORT sequences (at most 64 tokens: the 14x14 and 7x7 stages), ONE launch: a WORKGROUP = one image, a WAVE = one head.  The image's d is staged
// in LDS once (with a plane row + a token of zeros either side for the 3x3 of pe; one barrier), after which the waves share nothing:
//   1. k = elu1(Wk d + bk) tile by tile (lane = channel, tokens in the registers), its column sums, kv += k^T v (A = the k accumulators as they
//      stand, B = v gathered from the LDS image in the token order those registers imply);
//   2. q^T = Wq d^T (lane = token), then the shared tile epilogue (out_epilogue: normaliser, out^T = kv^T q^T, pe, 16-byte stores).
// The first version kept q in the lane = channel form and took it through a wave-private LDS image for the second product, computed pe with 9
// bounds-checked 4-byte LDS reads per output and divided per output: 26-28 us at 256 x 49 x 256, of which pe 8.5 (profiles/archive/r04_recattn_one_launch.txt).
// Weight fragments come from global memory (L2) through a ring PF k-steps deep; NT = 32-token tiles, KS = C / 32 = heads.
// XW > 0: d is not read from memory but computed here, d = conv5 stride 2 (x) + bias of RecAttn2d.forward (model/recattn.py:61), from the image's XW x XW
// plane of 16-bit activations (XW = 14 / 7: Hp = Wp = 7 / 4): two lanes per channel, each the upper / lower output rows, every x row loaded once and
// scattered into the output rows it touches (float32).  Saves the stand-alone step's launch and d's trip through memory.
// FULL (with XW > 0): the unit's last step too, y = conv5(x + nearest-resize(a)) + bias (model/recattn.py:67): a stays in LDS (one more barrier), the
// same two lanes per channel form the upper / lower output rows from x rows loaded once each -- RecAttn2d.forward in one launch.
// HPW = heads per wave (2: the 7 x 7 plane with 16 heads as ONE launch -- 8 waves with 256 registers hold the final conv, 16 waves with 128 do not; the heads of a
// wave run one after the other through the same code, the next head's first weight fragments requested while the current one's epilogue runs)
template <int NT, int KS, int XW = 0, typename TX = bf16_t, bool FULL = false, bool PAD = false, int HPW = 1>      // PAD: head dimension < 32 (Dr); else every mask below folds away
__global__ void __launch_bounds__(64 * KS / HPW)
k_recattn_short(const float* __restrict__ d, const bf16_t* __restrict__ wqk, const float* __restrict__ bqk, const float* __restrict__ wpe,
                const float* __restrict__ bpe, float* __restrict__ out, int Hp, int Wp,
                const TX* __restrict__ x, const float* __restrict__ wdn, const float* __restrict__ bdn,
                const float* __restrict__ wcv, const float* __restrict__ bcv, TX* __restrict__ yout, int Dr)
{
    static_assert(!FULL || XW > 0, "the whole unit starts from x");
    static_assert(HPW == 1 || (HPW == 2 && FULL && KS % 2 == 0 && (2 * KS) % (KS < 8 ? KS : 8) == 0), "two heads per wave: the whole-unit form only");
    const int D = PAD ? Dr : 32;
    extern __shared__ __attribute__((aligned(16))) float lds_s[];
    constexpr int C = 32 * KS, K = C / 2, DROW = C + DPAD, DROW4 = DROW / 4, NTHR = 64 * KS / HPW, PF = KS < 8 ? KS : 8;        // weight fragments in flight: a ring 4 deep left ~0.4 us of L2 latency exposed at every refill
    const int hd0 = (threadIdx.x >> 6) * HPW, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x, n = Hp * Wp, rows = NT * 32 + 2 * Wp + 2, R0 = -Wp - 1;
    float* const Ld = lds_s;                                  // [rows + 1][DROW]: tokens R0 .. of the image (zeros outside it), then a row of zeros
    const float4* const Ld4 = reinterpret_cast<const float4*>(lds_s);
    const f32x4q* const Ldv = reinterpret_cast<const f32x4q*>(lds_s);
    float* const Lw = Ld + (size_t)(rows + 1) * DROW;         // [9][C] pe taps, [C] pe bias, [C] q biases, [C] kbar
    // C, K = the PADDED channel count and half (32 per head); Cg, Kg = what memory holds (D per head)
    const int Cg = KS * D, Kg = Cg / 2;
    const float* dimg = d + (size_t)b * n * Cg;
    const __amdgpu_buffer_rsrc_t dsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dimg, 0, n * Cg * 4, 0x00020000);

    const bool rvalid = r < D;                                                 // this lane's output channel of the head exists
    const bf16_t* wq_row = wqk + (size_t)(hd0 * D + (rvalid ? r : 0)) * Kg;     // its weight row (compact); wfrag picks the 8 inputs of a k-step
    const bf16_t* wk_row = wqk + (size_t)(Cg + hd0 * D + (rvalid ? r : 0)) * Kg;
    u32x4q wf[PF];
    if constexpr (XW == 0) {                                  // (with the conv inside: requested after it -- its 70 live registers leave no room at 16 waves)
#pragma unroll
        for (int j = 0; j < PF; ++j) wf[j] = wfrag<PAD>(wk_row, rvalid, 16 * j + 8 * h, D);
    }

    if constexpr (XW == 0) {
        stage_rows<NT == 2 ? 12 : 6>(Ld, dsrc, R0, rows, Cg, D, KS, DROW, NTHR);      // rows / 8 requests per lane: the whole image in one round trip up to 14 x 14 / 7 x 7 planes
        for (int i = threadIdx.x; i < DROW; i += NTHR) Ld[(size_t)rows * DROW + i] = 0.f;
    } else {
        constexpr int WO = (XW + 1) / 2, RPP = NTHR == C ? WO : (WO + 1) / 2, NXR = 2 * RPP + 3;          // output plane, output rows per lane (all of them with one lane per channel), x rows they touch
        static_assert(NTHR == 2 * C || NTHR == C, "two lanes per channel, or one (two heads per wave)");
        const int cp = threadIdx.x < C ? threadIdx.x : threadIdx.x - C, o0 = threadIdx.x < C ? 0 : WO - RPP;      // padded channel (per wave for C >= 64; C = 32 has both halves in its one wave)
        const bool cvalid = (cp & 31) < D;                                     // a padding lane runs on channel 0 and leaves zeros
        const int c = cvalid ? (cp >> 5) * D + (cp & 31) : 0;
        const TX* xc = x + (size_t)b * XW * XW * Cg + c;
        float wt[25];
#pragma unroll
        for (int j = 0; j < 25; ++j) wt[j] = wdn[j * Cg + c];
        const float bias = bdn ? bdn[c] : 0.f;
        // rows of the LDS image that are not tokens: zeros (the halo of pe, the rows of tokens past the plane, the zero row)
        {
            float4* L4 = reinterpret_cast<float4*>(lds_s);
            const int tok0 = (Wp + 1) * DROW4, tok1 = (Wp + 1 + n) * DROW4, total = (rows + 1) * DROW4;
            for (int i = threadIdx.x; i < total; i += NTHR)
                if (i < tok0 || i >= tok1) L4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float a[RPP][WO];
#pragma unroll
        for (int o = 0; o < RPP; ++o)
#pragma unroll
            for (int j = 0; j < WO; ++j) a[o][j] = bias;
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int y = 2 * o0 - 2 + i;
            if (y >= 0 && y < XW) {
                float xr[XW];
#pragma unroll
                for (int xx = 0; xx < XW; ++xx) xr[xx] = elem_to_f32(xc[(size_t)(y * XW + xx) * Cg]);
#pragma unroll
                for (int o = 0; o < RPP; ++o) {
                    const int dy = i - 2 * o;                  // compile time after unrolling
                    if (dy >= 0 && dy < 5) {
#pragma unroll
                        for (int j = 0; j < WO; ++j)
#pragma unroll
                            for (int dx = 0; dx < 5; ++dx) {
                                const int xx = 2 * j - 2 + dx;
                                if (xx >= 0 && xx < XW) a[o][j] = fmaf(wt[dy * 5 + dx], xr[xx], a[o][j]);
                            }
                    }
                }
            }
        }
        // (an odd output plane with two lanes per channel -- 7 rows as 0 .. 3 and 3 .. 6 -- has its middle row formed twice: ONE lane writes it, see the final conv below)
#pragma unroll
        for (int o = 0; o < RPP; ++o) {
            if (NTHR != C && 2 * RPP > WO && o == 0 && threadIdx.x >= C) continue;
#pragma unroll
            for (int j = 0; j < WO; ++j) Ld[(size_t)(Wp + 1 + (o0 + o) * Wp + j) * DROW + cp] = cvalid ? a[o][j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) wf[j] = wfrag<PAD>(wk_row, rvalid, 16 * j + 8 * h, D);
    }
    stage_small<C, NTHR, PAD>(Lw, wpe, bpe, bqk, D);
    __syncthreads();

#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {                        // the heads of this wave, one after the other
        const int hd = hd0 + hh;
        const bf16_t* const wq_h = wq_row + (size_t)hh * D * Kg;
        const bf16_t* const wk_h = wk_row + (size_t)hh * D * Kg;
        const float bk = rvalid ? bqk[Cg + hd * D + r] : QK_PAD_BIAS;
        // ---- 1. k tiles (lane = channel hd * 32 + r, tokens in the registers), kv
        f32x16 acc[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tt][i] = 0.f;
        // the d fragments of k-step s + 1 are read from LDS before the products of k-step s are issued (the compiler waits for every read in
        // front of the product that uses it: ~190 cycles a step instead of the product's 64)
        f32x4q xf[2][NT][2];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const f32x4q* pa = Ldv + (32 * tt + r - R0) * DROW4 + K / 4 + 2 * h;
            xf[0][tt][0] = pa[0]; xf[0][tt][1] = pa[1];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 bk8 = __builtin_bit_cast(bf16x8, wf[s % PF]);
            if (s + PF < KS) wf[s % PF] = wfrag<PAD>(wk_h, rvalid, 16 * (s + PF) + 8 * h, D);
            else wf[s % PF] = wfrag<PAD>(wq_h, rvalid, 16 * (s + PF - KS) + 8 * h, D);          // the ring rolls over into the q weights
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                // next: k-step s + 1 of the k half, or k-step 0 of the q half
                const f32x4q* pa = Ldv + (32 * tt + r - R0) * DROW4 + (s + 1 < KS ? K / 4 + 4 * (s + 1) : 0) + 2 * h;
                xf[(s + 1) & 1][tt][0] = pa[0]; xf[(s + 1) & 1][tt][1] = pa[1];
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const bf16x8 fa = to_bf16x8(xf[s & 1][tt][0], xf[s & 1][tt][1]);          // token 32 tt + r, inputs 16 s + 8 h .. + 7 of the k half
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, bk8, acc[tt], 0, 0, 0);
            }
        }
        const float* const Lv = Ld + hd * 32 + r;                 // this lane's channel of d: v
        f32x16 kv;
#pragma unroll
        for (int i = 0; i < 16; ++i) kv[i] = 0.f;
        float ksum = 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 fa, fb;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * s2 + j, t = 32 * tt + acc_row(i, h);
                    const float kk = t < n ? elu1(acc[tt][i] + bk) : 0.f;
                    ksum += kk;
                    fa[j] = (__bf16)kk;
                    fb[j] = (__bf16)Lv[(t - R0) * DROW];
                }
                kv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, kv, 0, 0, 0);
            }
        const float inv_n = 1.f / (float)n;
        const float kbar_mine = (ksum + __shfl_xor(ksum, 32)) * inv_n;          // lane (r, h): kbar of channel r of the head
        float* const Lkb = Lw + 11 * C + hd * 32;
        if (h == 0) Lkb[r] = kbar_mine;
        bf16x8 kv0, kv1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { kv0[j] = (__bf16)(kv[j] * inv_n); kv1[j] = (__bf16)(kv[8 + j] * inv_n); }
        wave_sync();

        // ---- 2. q^T tiles (lane = token), epilogue
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tt][i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 bq8 = __builtin_bit_cast(bf16x8, wf[(s + KS) % PF]);
            if (s + PF < KS) wf[(s + KS) % PF] = wfrag<PAD>(wq_h, rvalid, 16 * (s + PF) + 8 * h, D);
            else if (hh + 1 < HPW) wf[(s + KS) % PF] = wfrag<PAD>(wk_h + (size_t)D * Kg, rvalid, 16 * (s + PF - KS) + 8 * h, D);          // ... and into the wave's next head (slot (s + KS) % PF = (s + PF - KS) % PF: 2 KS % PF == 0)
            if (s + 1 < KS) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const f32x4q* pa = Ldv + (32 * tt + r - R0) * DROW4 + 4 * (s + 1) + 2 * h;
                    xf[(s + 1 + KS) & 1][tt][0] = pa[0]; xf[(s + 1 + KS) & 1][tt][1] = pa[1];
                }
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const bf16x8 fb = to_bf16x8(xf[(s + KS) & 1][tt][0], xf[(s + KS) & 1][tt][1]);          // token 32 tt + r, inputs 16 s + 8 h .. + 7 of the q half
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq8, fb, acc[tt], 0, 0, 0);
            }
        }
        const float4* const zrow = Ld4 + (size_t)rows * DROW4 + (hd * 32 + 4 * h) / 4;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = 32 * tt + r;
            if constexpr (FULL)             // a stays in LDS, padded like d (its padding columns come out as zeros: zero kv columns, zero pe)
                out_epilogue<C, false>(acc[tt], kv0, kv1, Lw + 10 * C + hd * 32, Lkb, Ld4 + (t - R0) * DROW4 + (hd * 32 + 4 * h) / 4, zrow,
                                       reinterpret_cast<const float4*>(Lw) + (hd * 32 + 4 * h) / 4, t, n, Wp, h, Lw + 12 * C + hd * 32 + 4 * h, 0, DROW, 32);
            else
                out_epilogue<C, PAD>(acc[tt], kv0, kv1, Lw + 10 * C + hd * 32, Lkb, Ld4 + (t - R0) * DROW4 + (hd * 32 + 4 * h) / 4, zrow,
                                     reinterpret_cast<const float4*>(Lw) + (hd * 32 + 4 * h) / 4, t, n, Wp, h, out + (size_t)b * n * Cg, hd * D + 4 * h, Cg, D - 4 * h);
        }
    }
    if constexpr (FULL) {
        // ---- 3. y = conv5(x + resize(a)) + bias: lane = (channel, upper / lower output rows), a[token][channel] float32 in LDS behind Lw
        __syncthreads();
        constexpr int ORP = NTHR == C ? XW : (XW + 1) / 2;     // output rows per lane (7 x 7: row 3 is formed by both lanes and stored by the first; one lane per channel: all rows)
        const int cp = threadIdx.x < C ? threadIdx.x : threadIdx.x - C, o0 = threadIdx.x < C ? 0 : XW - ORP;      // padded channel, as in the conv above
        const bool cvalid = (cp & 31) < D;
        const int c = cvalid ? (cp >> 5) * D + (cp & 31) : 0;
        const TX* xc = x + (size_t)b * XW * XW * Cg + c;
        const float* La = Lw + 12 * C + cp;
      if (cvalid) {                                            // (Cg = the pitch of x, y and the tap packs in memory)
      if constexpr (XW == 14 && C >= 64) {
        // Round 5: the 14 x 14 plane on PIXEL PAIRS -- v_pk_fma_f32 throughout, input-row stationary with five accumulator rows in
        // flight (70 registers instead of 98), the 35 values of a the lane's rows interpolate from read from LDS once (the scalar form read 14 per x row)
        // -- the row pieces of rcx_cpl14_pieces.h.  The upper / lower half is the wave's (C >= 64): rows and edge cases are compile-time in each branch.
        using cpl14::f32x2;
        cpl14::Taps t2;
        cpl14::load_taps<0>(t2, wcv, bcv, 0, Cg, (unsigned)c * 4u, bcv != nullptr);
        auto half = [&](auto hc) {
            constexpr int O0 = decltype(hc)::value ? XW - ORP : 0, T0 = O0 - 2 < 0 ? 0 : O0 - 2, T1 = O0 + ORP + 1 > XW - 1 ? XW - 1 : O0 + ORP + 1;   // input rows T0 .. T1
            constexpr int A0 = T0 >> 1, NA = (T1 >> 1) - A0 + 1;                   // rows of a (nearest: source = destination >> 1)
            float av[NA][7];
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) av[i][j] = La[((A0 + i) * Wp + j) * DROW];
            TX cur[XW], nxt[XW];
            auto load_xrow = [&](TX (&dst)[XW], int y) {
#pragma unroll
                for (int xx = 0; xx < XW; ++xx) dst[xx] = xc[(size_t)(y * XW + xx) * Cg];
            };
            load_xrow(cur, T0);
            f32x2 acc[5][7];
            TX* yc = yout + (size_t)b * XW * XW * Cg + c;
            lanes::sfor<T1 - T0 + 1>([&](auto tcn) {
                constexpr int t = T0 + decltype(tcn)::value;
                if constexpr (t < T1) load_xrow(nxt, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                // output rows that this input row opens start from the bias
#pragma unroll
                for (int o = O0; o < O0 + ORP; ++o) {
                    const bool opens = (o - 2 <= T0) ? (t == T0) : (t == o - 2);
                    if (opens && o >= t - 2 && o <= t + 2) {
#pragma unroll
                        for (int j = 0; j < 7; ++j) acc[o % 5][j] = cpl14::splat(t2.bias);
                    }
                }
                f32x2 row[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float a_ = av[(t >> 1) - A0][j];
                    row[j] = f32x2{elem_to_f32(cur[2 * j]) + a_, elem_to_f32(cur[2 * j + 1]) + a_};
                }
                // one input row into the accumulator rows of THIS half it feeds (cpl14::conv5_row restricted to O0 .. O0 + ORP - 1)
                {
                    f32x2 odd[8];
                    const f32x2 zero = f32x2{0.f, 0.f};
                    odd[0] = cpl14::shift1(zero, row[0]);
#pragma unroll
                    for (int j = 1; j < 7; ++j) odd[j] = cpl14::shift1(row[j - 1], row[j]);
                    odd[7] = cpl14::shift1(row[6], zero);
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const int o = t - u + 2;
                        if (o < O0 || o >= O0 + ORP) continue;
                        f32x2(&a)[7] = acc[o % 5];
#pragma unroll
                        for (int j = 1; j < 7; ++j) a[j] = cpl14::pfma(row[j - 1], cpl14::splat(t2.at(u, 0)), a[j]);
#pragma unroll
                        for (int j = 0; j < 7; ++j) a[j] = cpl14::pfma(odd[j], cpl14::splat(t2.at(u, 1)), a[j]);
#pragma unroll
                        for (int j = 0; j < 7; ++j) a[j] = cpl14::pfma(row[j], cpl14::splat(t2.at(u, 2)), a[j]);
#pragma unroll
                        for (int j = 0; j < 7; ++j) a[j] = cpl14::pfma(odd[j + 1], cpl14::splat(t2.at(u, 3)), a[j]);
#pragma unroll
                        for (int j = 0; j + 1 < 7; ++j) a[j] = cpl14::pfma(row[j + 1], cpl14::splat(t2.at(u, 4)), a[j]);
                    }
                }
                // output row t - 2 has seen its last input row (the plane's last rows: with the last input row)
#pragma unroll
                for (int o = O0; o < O0 + ORP; ++o) {
                    const bool done = (o + 2 >= T1) ? (t == T1) : (t == o + 2);
                    if (done) {
#pragma unroll
                        for (int j = 0; j < 7; ++j) {
                            const float v2[1] = {acc[o % 5][j].x}, v3[1] = {acc[o % 5][j].y};
                            store_vec<1>(yc + (size_t)(o * XW + 2 * j) * Cg, v2);
                            store_vec<1>(yc + (size_t)(o * XW + 2 * j + 1) * Cg, v3);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int xx = 0; xx < XW; ++xx) cur[xx] = nxt[xx];
            });
        };
        if (threadIdx.x < C) half(lanes::IC<0>{}); else half(lanes::IC<1>{});
      } else {
        float wt[25];
#pragma unroll
        for (int j = 0; j < 25; ++j) wt[j] = wcv[j * Cg + c];
        const float bias = bcv ? bcv[c] : 0.f;
        float a[ORP][XW];
#pragma unroll
        for (int o = 0; o < ORP; ++o)
#pragma unroll
            for (int j = 0; j < XW; ++j) a[o][j] = bias;
        // x rows one ahead of the row being used, and no further: left alone the compiler requests all 11 rows (154 registers) up front and spills
        TX cur[XW], nxt[XW];
        auto load_xrow = [&](TX (&dst)[XW], int y) {
            const int yc_ = y < 0 ? 0 : (y > XW - 1 ? XW - 1 : y);                  // rows outside the plane: a valid row, not used
#pragma unroll
            for (int xx = 0; xx < XW; ++xx) dst[xx] = xc[(size_t)(yc_ * XW + xx) * Cg];
        };
        load_xrow(cur, o0 - 2);
#pragma unroll
        for (int i = 0; i < ORP + 4; ++i) {
            const int y = o0 - 2 + i;
            if (i + 1 < ORP + 4) load_xrow(nxt, y + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (y >= 0 && y < XW) {
                float tr[XW];                                  // row y of x + resize(a): nearest, source index = destination >> 1 for 14 <- 7 and 7 <- 4
#pragma unroll
                for (int xx = 0; xx < XW; ++xx) tr[xx] = elem_to_f32(cur[xx]) + La[((y >> 1) * Wp + (xx >> 1)) * DROW];
#pragma unroll
                for (int o = 0; o < ORP; ++o) {
                    const int dy = i - o;                      // compile time after unrolling
                    if (dy >= 0 && dy < 5) {
#pragma unroll
                        for (int j = 0; j < XW; ++j)
#pragma unroll
                            for (int dx = 0; dx < 5; ++dx) {
                                const int xx = j - 2 + dx;
                                if (xx >= 0 && xx < XW) a[o][j] = fmaf(wt[dy * 5 + dx], tr[xx], a[o][j]);
                            }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xx = 0; xx < XW; ++xx) cur[xx] = nxt[xx];
        }
        TX* yc = yout + (size_t)b * XW * XW * Cg + c;
        // With two lanes per channel and an odd plane the middle row is formed by both (rows 0 .. 3 and 3 .. 6 of 7): ONE of them stores it.  Both storing "the same
        // value" was a race -- the two lanes' code is unrolled at different positions and the compiler is free to contract their sums differently: a float16 run of
        // 2 x 128 x 7 x 7 / 4 heads differed in one element by one ulp from launch to launch (round 6, tools/stress_recattn_unit.py).
        const bool second = NTHR != C && threadIdx.x >= C;
#pragma unroll
        for (int o = 0; o < ORP; ++o) {
            if (NTHR != C && 2 * ORP > XW && o == 0 && second) continue;          // the shared middle row: the first half's
#pragma unroll
            for (int j = 0; j < XW; ++j) {
                const float v1[1] = {a[o][j]};
                store_vec<1>(yc + (size_t)((o0 + o) * XW + j) * Cg, v1);
            }
        }
      }
      }
    }
}

// full = the whole-unit form: + a[Wp * Wp tokens][C + 4] behind the small arrays
static inline size_t short_lds_bytes(int NT, int Wp, int C, bool full = false)
{
    return sizeof(float) * ((size_t)(NT * 32 + 2 * Wp + 3 + (full ? Wp * Wp : 0)) * (C + DPAD) + 12 * (size_t)C);
}


constexpr int TILE_TOK = 32;
constexpr int PART_FLOATS = 17 * 64;       // per (image, range, head): 16 accumulator registers + the column-sum partial of each lane

struct LongGeo { int ntiles, tpg, G, rows, waves, S; size_t lds_out; int tpgA, GA, SA; size_t lds_kv, ws_bytes; };
static inline LongGeo long_geo(int B, int Hp, int Wp, int C, int heads)
{
    LongGeo g{};
    const int n = Hp * Wp;
    g.ntiles = (n + TILE_TOK - 1) / TILE_TOK;
    // second kernel: ranges whose rows (+ halo) take at most half a CU's LDS
    const int halo = 2 * Wp + 2;
    int tpg = ((int)((80 * 1024 - 48 * (size_t)C) / (4 * (size_t)(C + DPAD))) - halo - 1) / TILE_TOK;
    if (tpg < 1) tpg = 1;
    if (tpg > g.ntiles) tpg = g.ntiles;
    g.G = (g.ntiles + tpg - 1) / tpg;
    g.tpg = (g.ntiles + g.G - 1) / g.G;
    g.G = (g.ntiles + g.tpg - 1) / g.tpg;
    g.rows = g.tpg * TILE_TOK + halo;
    g.waves = 8;
    g.S = g.waves / heads;
    g.lds_out = sizeof(float) * ((size_t)(g.rows + 1) * (C + DPAD) + 12 * (size_t)C);      // rows of d, 9 + 1 rows of pe taps / bias, the q biases, kbar
    // first kernel: 16 waves, as few ranges per image as keeps a wave's serial tiles short (every range's partial is read by every workgroup of the second)
    g.SA = 16 / heads;
    g.GA = (g.ntiles + 63) / 64;
    g.tpgA = (g.ntiles + g.GA - 1) / g.GA;
    g.GA = (g.ntiles + g.tpgA - 1) / g.tpgA;
    g.lds_kv = sizeof(float) * 16 * (size_t)PART_FLOATS;
    g.ws_bytes = sizeof(float) * (size_t)B * g.GA * heads * PART_FLOATS;
    return g;
}

template <int KS, bool PAD = false>
__global__ void __launch_bounds__(1024)
k_recattn_kv(const float* __restrict__ d, const bf16_t* __restrict__ wqk, const float* __restrict__ bqk, float* __restrict__ part,
             int n, int Dr, int heads, int S, int tpg, int ntiles)
{
    const int D = PAD ? Dr : 32;
    extern __shared__ __attribute__((aligned(16))) float lds_kv[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int hd = wv % heads, sp = wv / heads;
    constexpr int C = 32 * KS;                    // C / heads == 32 and 16 inputs per k-step of each half: a compile-time C makes every offset an immediate
    const int g = blockIdx.x, b = blockIdx.y, G = gridDim.x, K = C / 2;
    const int Cg = KS * D, Kg = Cg / 2;             // C, K: padded (32 per head); Cg, Kg: what memory holds (see wfrag)
    const bool rvalid = r < D;
    const float* dimg = d + (size_t)b * n * Cg;
    const __amdgpu_buffer_rsrc_t dsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dimg, 0, n * Cg * 4, 0x00020000);
    const bf16_t* wk_row = wqk + (size_t)(Cg + hd * D + (rvalid ? r : 0)) * Kg;
    bf16x8 wk[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wk[s] = __builtin_bit_cast(bf16x8, wfrag<PAD>(wk_row, rvalid, 16 * s + 8 * h, D));
    const float bk = rvalid ? bqk[Cg + hd * D + r] : QK_PAD_BIAS;
    // the 8 inputs of k-step s of the k half for this lane: padded columns K + 16 s + 8 h .. = compact channels ci[s] .. (two groups of four, each real or padding)
    // as byte offsets within a token's row; a padding group: 2^31, which puts the request past the buffer whatever the token (the image is below 2^31 bytes) -- reads 0
    unsigned sl[KS], sh[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int pc = K + 16 * s + 8 * h, off = pc & 31, ci = (pc >> 5) * D + off;
        sl[s] = off < D ? 4u * ci : 0x80000000u;
        sh[s] = off + 4 < D ? 4u * ci + 16u : 0x80000000u;
    }
    const unsigned vlane = rvalid ? 4u * (4 * h * Cg + hd * D + r) : 0x80000000u;      // v of this lane's channel; a padding channel: past the buffer, reads 0
    f32x16 kv;
#pragma unroll
    for (int i = 0; i < 16; ++i) kv[i] = 0.f;
    float ksum = 0.f;
    const int t_end = min((g + 1) * tpg, ntiles);
    for (int tt = g * tpg + sp; tt < t_end; tt += S) {
        f32x16 ak;
#pragma unroll
        for (int i = 0; i < 16; ++i) ak[i] = 0.f;
        // v of this tile, in the token order of the accumulator registers (requested first: the longest wait)
        float vv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)              // (a uniform row offset + the lane's part: with Cg a run-time value the 16 products would otherwise be 16 loop-invariant registers)
            vv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dsrc, (int)((unsigned)((32 * tt + acc_row(i, 0)) * Cg) * 4u + vlane), 0, 0));
        // (8 heads of fewer than 32 channels: the 16 requests of a tile in four rounds -- their 16 selected offsets on top of the 64 registers of data do not fit the 128 of a
        // 1024-thread workgroup; 7 of the offsets still live in scratch, 44 B, reloaded once per tile)
        constexpr int SB = PAD && KS == 8 ? 2 : KS;
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += SB) {
            f32x4q xlo[SB], xhi[SB];
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                const unsigned trow = (unsigned)((32 * tt + r) * Cg) * 4u;              // token 32 tt + r, inputs 16 s + 8 h .. + 7 of the k half
                xlo[s] = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(dsrc, (int)(trow + sl[s0 + s]), 0, 0));
                xhi[s] = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(dsrc, (int)(trow + sh[s0 + s]), 0, 0));
            }
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                bf16x8 fa;
                fa[0] = (__bf16)xlo[s].x; fa[1] = (__bf16)xlo[s].y; fa[2] = (__bf16)xlo[s].z; fa[3] = (__bf16)xlo[s].w;
                fa[4] = (__bf16)xhi[s].x; fa[5] = (__bf16)xhi[s].y; fa[6] = (__bf16)xhi[s].z; fa[7] = (__bf16)xhi[s].w;
                ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, wk[s0 + s], ak, 0, 0, 0);
            }
            if constexpr (SB < KS) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 fa, fb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * s2 + j, t = 32 * tt + acc_row(i, h);
                const float kk = t < n ? elu1(ak[i] + bk) : 0.f;
                ksum += kk;
                fa[j] = (__bf16)kk;
                fb[j] = (__bf16)vv[i];
            }
            kv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, kv, 0, 0, 0);
        }
    }
    // the splits of a head -> one partial per (image, range, head), summed in split order
    float* mine = lds_kv + (size_t)wv * PART_FLOATS;
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = kv[i];
    mine[16 * 64 + lane] = ksum;
    __syncthreads();
    if (sp == 0) {
        float* dst = part + (((size_t)b * G + g) * heads + hd) * PART_FLOATS;
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            float a = 0.f;
            for (int q = 0; q < S; ++q) a += lds_kv[(size_t)(q * heads + hd) * PART_FLOATS + i * 64 + lane];
            dst[i * 64 + lane] = a;
        }
    }
}

template <int KS, int NTHR, bool PAD = false>
__global__ void __launch_bounds__(NTHR, KS <= 4 ? 4 : 2)          // two workgroups per CU (128 registers) where the weights leave room
k_recattn_out(const float* __restrict__ d, const bf16_t* __restrict__ wqk, const float* __restrict__ bqk, const float* __restrict__ wpe,
              const float* __restrict__ bpe, const float* __restrict__ part, float* __restrict__ out,
              int Hp, int Wp, int Dr, int heads, int S, int tpg, int ntiles, int rows, int GA)
{
    const int D = PAD ? Dr : 32;
    extern __shared__ __attribute__((aligned(16))) float lds_o[];
    constexpr int C = 32 * KS;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int hd = wv % heads, sp = wv / heads;
    constexpr int DROW = C + DPAD;
    const int g = blockIdx.x, b = blockIdx.y, n = Hp * Wp;
    float* const Ld = lds_o;                      // [rows][DROW]: tokens R0 .. R0 + rows - 1 of the image (zeros outside it)
    const float4* const zrow = reinterpret_cast<const float4*>(lds_o) + (size_t)rows * (DROW / 4) + (hd * 32 + 4 * h) / 4;      // row `rows`: zeros
    float* const Lw = Ld + (size_t)(rows + 1) * DROW;   // [9][C] pe taps, then [C] pe bias, [C] q biases, [C] kbar (a lane's 16 of each would be 32 registers)
    float* const Lbq = Lw + 10 * C + hd * 32;
    float* const Lkb = Lw + 11 * C + hd * 32;
    const float4* const Lw4 = reinterpret_cast<const float4*>(Lw);
    const int T0 = g * tpg * TILE_TOK, R0 = T0 - Wp - 1;
    const int Cg = KS * D, Kg = Cg / 2;             // C, K: padded; Cg, Kg: in memory
    const bool rvalid = r < D;
    const float* dimg = d + (size_t)b * n * Cg;
    const __amdgpu_buffer_rsrc_t dsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dimg, 0, n * Cg * 4, 0x00020000);

    stage_rows(Ld, dsrc, R0, rows, Cg, D, KS, DROW, NTHR);
    __builtin_amdgcn_sched_barrier(0);            // keep the loads below from being hoisted among the staging requests (register pressure)
    for (int i = threadIdx.x; i < DROW; i += NTHR) Ld[(size_t)rows * DROW + i] = 0.f;
    stage_small<C, NTHR, PAD>(Lw, wpe, bpe, bqk, D);
    // this head's weights and the partial sums (requested after the staging loop, whose eight 16-byte requests per lane would otherwise be live with them: 166-202 registers; waited for at the barrier)
    const bf16_t* wq_row = wqk + (size_t)(hd * D + (rvalid ? r : 0)) * Kg;
    bf16x8 wq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wq[s] = __builtin_bit_cast(bf16x8, wfrag<PAD>(wq_row, rvalid, 16 * s + 8 * h, D));
    float kvs[16], ks = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) kvs[i] = 0.f;
    for (int q = 0; q < GA; ++q) {
        const float* src = part + (((size_t)b * GA + q) * heads + hd) * PART_FLOATS;
#pragma unroll
        for (int i = 0; i < 16; ++i) kvs[i] += src[i * 64 + lane];
        ks += src[16 * 64 + lane];
    }
    const float inv_n = 1.f / (float)n;
    // kbar: channel c of the head = lane c's partial + lane 32 + c's
    const float kbar_mine = (ks + __shfl_xor(ks, 32)) * inv_n;       // lane (r, h): kbar of channel r
    bf16x8 kv0, kv1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { kv0[j] = (__bf16)(kvs[j] * inv_n); kv1[j] = (__bf16)(kvs[8 + j] * inv_n); }
    if (sp == 0 && h == 0) Lkb[r] = kbar_mine;
    __syncthreads();

    const int t_end = min((g + 1) * tpg, ntiles);
    for (int tt = g * tpg + sp; tt < t_end; tt += S) {
        const int t = 32 * tt + r;                                    // this lane's token
        const float4* drow = reinterpret_cast<const float4*>(lds_o) + (t - R0) * (DROW / 4);
        f32x16 aq;
#pragma unroll
        for (int i = 0; i < 16; ++i) aq[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4q* dv4 = reinterpret_cast<const f32x4q*>(drow);
            const bf16x8 fb = to_bf16x8(dv4[4 * s + 2 * h], dv4[4 * s + 2 * h + 1]);
            aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s], fb, aq, 0, 0, 0);          // q^T: rows c1 (registers), column = this lane's token
        }
        out_epilogue<C, PAD>(aq, kv0, kv1, Lbq, Lkb, drow + (hd * 32 + 4 * h) / 4, zrow, Lw4 + (hd * 32 + 4 * h) / 4, t, n, Wp, h, out + (size_t)b * n * Cg, hd * D + 4 * h, Cg, D - 4 * h);
    }
}

}  // namespace qkc

// 16-bit-activation callers only (the operands of the products are bf16): heads a power of two up to 16, head dimension as heads_ok() below,
// 16-byte-aligned d.  ONE launch when the plane has at most 64 tokens and its image (+ halo) fits the CU's LDS (at most 8 heads above 32 tokens);
// else two launches and a workspace (at most 8 heads).  RCX_ATTN_FUSED=0: off, RCX_ATTN_FUSED=short: the one-launch form only (A/B).
static bool pow2_heads(int heads) { return heads > 0 && heads <= 16 && !(heads & (heads - 1)); }
// head dimension 32, or (round 5) 4 .. 28 in steps of 4 with an even number of heads (a q / k half is whole heads): RecNeXt-A0 / A1 / A2's 20 / 24 / 28.
// Inside the kernels a head is 32 wide either way (padded): LDS sizes and tile counts follow 32 heads, memory follows C.
static bool heads_ok(int C, int heads)
{
    if (!pow2_heads(heads) || C % heads) return false;
    const int D = C / heads;
    return D == 32 || (D >= 4 && D < 32 && D % 4 == 0 && heads % 2 == 0);
}
static bool qkc_short(int Hp, int Wp, int C, int heads)
{
    const int n = Hp * Wp;
    if (n > 64 || (n > 32 && heads > 8)) return false;
    return qkc::short_lds_bytes(n <= 32 ? 1 : 2, Wp, C) <= 160 * 1024;
}
bool recattn_qkcore_applicable(int B, int Hp, int Wp, int C, int heads)
{
    const char* v = rcx::opt::value(rcx::opt::ATTN_FUSED);
    if (v && *v == '0') return false;
    if (!(B > 0 && Hp > 0 && Wp > 0 && heads_ok(C, heads))) return false;
    if (qkc_short(Hp, Wp, 32 * heads, heads)) return true;
    if ((v && *v == 's') || heads > 8) return false;
    if ((size_t)Hp * Wp * C * 4 >= (size_t)1 << 31 || B > 65535) return false;
    const qkc::LongGeo g = qkc::long_geo(B, Hp, Wp, 32 * heads, heads);
    return g.lds_out <= 160 * 1024 && g.G <= 65535;
}

int recattn_qkcore_launches(int B, int Hp, int Wp, int C, int heads)
{
    if (!recattn_qkcore_applicable(B, Hp, Wp, C, heads)) return 0;
    return qkc_short(Hp, Wp, 32 * heads, heads) ? 1 : 2;
}

size_t recattn_qkcore_workspace_bytes(int B, int Hp, int Wp, int C, int heads)
{
    if (!recattn_qkcore_applicable(B, Hp, Wp, C, heads) || qkc_short(Hp, Wp, 32 * heads, heads)) return 0;
    return qkc::long_geo(B, Hp, Wp, 32 * heads, heads).ws_bytes;
}

// PAD = the head dimension is below 32 (D = C / heads at run time); the D == 32 instantiations keep every offset and mask a compile-time constant
template <int KS, bool PAD>
static hipError_t launch_long_p(const float* d, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe, float* out, float* ws,
                                int B, int Hp, int Wp, int C, int heads, hipStream_t s)
{
    const int D = C / heads;
    const qkc::LongGeo g = qkc::long_geo(B, Hp, Wp, 32 * heads, heads);
    {
        auto kfn = qkc::k_recattn_kv<KS, PAD>;
        RCX_SET_LDS_ONCE(kfn, g.lds_kv);
        hipLaunchKernelGGL(kfn, dim3((unsigned)g.GA, (unsigned)B), dim3(1024), g.lds_kv, s, d, wqk, bqk, ws, Hp * Wp, D, heads, g.SA, g.tpgA, g.ntiles);
    }
    {
        auto kfn = qkc::k_recattn_out<KS, 512, PAD>;
        RCX_SET_LDS_ONCE(kfn, g.lds_out);
        hipLaunchKernelGGL(kfn, dim3((unsigned)g.G, (unsigned)B), dim3(512), g.lds_out, s, d, wqk, bqk, wpe, bpe, ws, out, Hp, Wp, D, heads, g.S, g.tpg, g.ntiles, g.rows, g.GA);
    }
    return hipGetLastError();
}
template <int KS>
static hipError_t launch_long(const float* d, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe, float* out, float* ws,
                              int B, int Hp, int Wp, int C, int heads, hipStream_t s)
{
    return C == 32 * heads ? launch_long_p<KS, false>(d, wqk, bqk, wpe, bpe, out, ws, B, Hp, Wp, C, heads, s)
                           : launch_long_p<KS, true>(d, wqk, bqk, wpe, bpe, out, ws, B, Hp, Wp, C, heads, s);
}

template <int NT, int KS, bool PAD>
static hipError_t launch_short_p(const float* d, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe, float* out, int B, int Hp, int Wp, int D, hipStream_t s)
{
    const size_t lds = qkc::short_lds_bytes(NT, Wp, 32 * KS);
    auto kfn = qkc::k_recattn_short<NT, KS, 0, bf16_t, false, PAD>;
    RCX_SET_LDS_ONCE(kfn, lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(64 * KS), lds, s, d, wqk, bqk, wpe, bpe, out, Hp, Wp, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, D);
    return hipGetLastError();
}
template <int NT, int KS>
static hipError_t launch_short(const float* d, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe, float* out, int B, int Hp, int Wp, int D, hipStream_t s)
{
    return D == 32 ? launch_short_p<NT, KS, false>(d, wqk, bqk, wpe, bpe, out, B, Hp, Wp, D, s) : launch_short_p<NT, KS, true>(d, wqk, bqk, wpe, bpe, out, B, Hp, Wp, D, s);
}

// the same with the stride-2 conv inside (XW = 14: 49 tokens, two tiles; XW = 7: 16 tokens, one tile)
template <int NT, int KS, int XW, typename TX, bool PAD>
static hipError_t launch_short_x_p(const void* x, const float* wdn, const float* bdn, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe,
                                   float* out, int B, int D, hipStream_t s)
{
    constexpr int WO = (XW + 1) / 2;
    const size_t lds = qkc::short_lds_bytes(NT, WO, 32 * KS);
    auto kfn = qkc::k_recattn_short<NT, KS, XW, TX, false, PAD>;
    RCX_SET_LDS_ONCE(kfn, lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(64 * KS), lds, s, (const float*)nullptr, wqk, bqk, wpe, bpe, out, WO, WO, (const TX*)x, wdn, bdn,
                       (const float*)nullptr, (const float*)nullptr, (TX*)nullptr, D);
    return hipGetLastError();
}
template <int NT, int KS, int XW, typename TX>
static hipError_t launch_short_x(const void* x, const float* wdn, const float* bdn, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe,
                                 float* out, int B, int D, hipStream_t s)
{
    return D == 32 ? launch_short_x_p<NT, KS, XW, TX, false>(x, wdn, bdn, wqk, bqk, wpe, bpe, out, B, D, s)
                   : launch_short_x_p<NT, KS, XW, TX, true>(x, wdn, bdn, wqk, bqk, wpe, bpe, out, B, D, s);
}

// RecAttn2d.forward in ONE launch (nearest resize): the conv inside, the attention output kept in LDS, the final conv inside
template <int NT, int KS, int XW, typename TX, bool PAD>
static hipError_t launch_unit_p(const void* x, const float* wdn, const float* bdn, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe,
                                const float* wcv, const float* bcv, void* y, int B, int D, hipStream_t s)
{
    constexpr int WO = (XW + 1) / 2, HPW = KS == 16 ? 2 : 1;          // 16 heads: two per wave (8 waves, 256 registers)
    const size_t lds = qkc::short_lds_bytes(NT, WO, 32 * KS, true);
    auto kfn = qkc::k_recattn_short<NT, KS, XW, TX, true, PAD, HPW>;
    RCX_SET_LDS_ONCE(kfn, lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(64 * KS / HPW), lds, s, (const float*)nullptr, wqk, bqk, wpe, bpe, (float*)nullptr, WO, WO, (const TX*)x, wdn, bdn, wcv, bcv, (TX*)y, D);
    return hipGetLastError();
}
template <int NT, int KS, int XW, typename TX>
static hipError_t launch_unit(const void* x, const float* wdn, const float* bdn, const bf16_t* wqk, const float* bqk, const float* wpe, const float* bpe,
                              const float* wcv, const float* bcv, void* y, int B, int D, hipStream_t s)
{
    return D == 32 ? launch_unit_p<NT, KS, XW, TX, false>(x, wdn, bdn, wqk, bqk, wpe, bpe, wcv, bcv, y, B, D, s)
                   : launch_unit_p<NT, KS, XW, TX, true>(x, wdn, bdn, wqk, bqk, wpe, bpe, wcv, bcv, y, B, D, s);
}

// RCX_ATTN_FUSED=twostep: the unit stays conv + attention | final conv (A/B)
bool recattn2d_unit_applicable(int B, int H, int W, int C, int heads, int x_dt, int mode)
{
    const char* v = rcx::opt::value(rcx::opt::ATTN_FUSED);
    if (v && *v == 't') return false;
    if (mode != 1 || !recattn_down_qkcore_applicable(B, H, W, C, heads, x_dt)) return false;      // (16 heads, 7 x 7 only: two heads per wave -- 16 waves of 128 registers do not hold the final conv)
    const int wo = (W + 1) / 2;
    return qkc::short_lds_bytes(wo * wo <= 32 ? 1 : 2, wo, 32 * heads, true) <= 160 * 1024;
}

hipError_t recattn2d_unit(const void* x, const float* wdn, const float* bdn, const void* wqk_bf16, const float* bqk, const float* wpe, const float* bpe,
                          const float* wcv, const float* bcv, void* y, int B, int H, int C, int heads, int x_dt, hipStream_t s)
{
    const bf16_t* w = (const bf16_t*)wqk_bf16;
#define RCX_UX(KS_)                                                                                                                            \
    (H == 14 ? (x_dt == 1 ? launch_unit<2, KS_, 14, bf16_t>(x, wdn, bdn, w, bqk, wpe, bpe, wcv, bcv, y, B, C / heads, s)                                  \
                          : launch_unit<2, KS_, 14, f16_t>(x, wdn, bdn, w, bqk, wpe, bpe, wcv, bcv, y, B, C / heads, s))                                  \
             : (x_dt == 1 ? launch_unit<1, KS_, 7, bf16_t>(x, wdn, bdn, w, bqk, wpe, bpe, wcv, bcv, y, B, C / heads, s)                                   \
                          : launch_unit<1, KS_, 7, f16_t>(x, wdn, bdn, w, bqk, wpe, bpe, wcv, bcv, y, B, C / heads, s)))
    switch (heads) {
        case 1: return RCX_UX(1);
        case 2: return RCX_UX(2);
        case 4: return RCX_UX(4);
        case 8: return RCX_UX(8);
        case 16: return H == 7 ? (x_dt == 1 ? launch_unit<1, 16, 7, bf16_t>(x, wdn, bdn, w, bqk, wpe, bpe, wcv, bcv, y, B, C / heads, s)
                                            : launch_unit<1, 16, 7, f16_t>(x, wdn, bdn, w, bqk, wpe, bpe, wcv, bcv, y, B, C / heads, s))
                               : hipErrorInvalidConfiguration;
        default: return hipErrorInvalidConfiguration;
    }
#undef RCX_UX
}

// RecAttn2d's stride-2 conv + coarse level in one launch: the 14 x 14 plane with 1 .. 8 heads, the 7 x 7 plane with 1 .. 16; 16-bit x.
// RCX_ATTN_FUSED=nodown: off (A/B)
bool recattn_down_qkcore_applicable(int B, int H, int W, int C, int heads, int x_dt)
{
    const char* v = rcx::opt::value(rcx::opt::ATTN_FUSED);
    if (v && (*v == '0' || *v == 'n')) return false;
    if (!(B > 0 && heads_ok(C, heads) && (x_dt == 1 || x_dt == 2))) return false;
    if (!((H == 14 && W == 14 && heads <= 8) || (H == 7 && W == 7))) return false;
    const int wo = (W + 1) / 2;
    return qkc_short(wo, wo, 32 * heads, heads);
}

hipError_t recattn_down_qkcore(const void* x, const float* wdn, const float* bdn, const void* wqk_bf16, const float* bqk, const float* wpe, const float* bpe,
                               float* out, int B, int H, int C, int heads, int x_dt, hipStream_t s)
{
    const bf16_t* w = (const bf16_t*)wqk_bf16;
#define RCX_DX(KS_)                                                                                                                            \
    (H == 14 ? (x_dt == 1 ? launch_short_x<2, KS_, 14, bf16_t>(x, wdn, bdn, w, bqk, wpe, bpe, out, B, C / heads, s)                                       \
                          : launch_short_x<2, KS_, 14, f16_t>(x, wdn, bdn, w, bqk, wpe, bpe, out, B, C / heads, s))                                       \
             : (x_dt == 1 ? launch_short_x<1, KS_, 7, bf16_t>(x, wdn, bdn, w, bqk, wpe, bpe, out, B, C / heads, s)                                        \
                          : launch_short_x<1, KS_, 7, f16_t>(x, wdn, bdn, w, bqk, wpe, bpe, out, B, C / heads, s)))
    switch (heads) {
        case 1: return RCX_DX(1);
        case 2: return RCX_DX(2);
        case 4: return RCX_DX(4);
        case 8: return RCX_DX(8);
        case 16: return H == 7 ? (x_dt == 1 ? launch_short_x<1, 16, 7, bf16_t>(x, wdn, bdn, w, bqk, wpe, bpe, out, B, C / heads, s)
                                            : launch_short_x<1, 16, 7, f16_t>(x, wdn, bdn, w, bqk, wpe, bpe, out, B, C / heads, s))
                               : hipErrorInvalidConfiguration;
        default: return hipErrorInvalidConfiguration;
    }
#undef RCX_DX
}

hipError_t recattn_qkcore(const float* d, const void* wqk_bf16, const float* bqk, const float* wpe, const float* bpe, float* out, void* workspace,
                          int B, int Hp, int Wp, int C, int heads, hipStream_t s)
{
    const bf16_t* w = (const bf16_t*)wqk_bf16;
    if (qkc_short(Hp, Wp, 32 * heads, heads)) {
        const bool one = Hp * Wp <= 32;
        switch (heads) {                  // = C / 32 = the k-steps of the projection (C / 2 inputs, 16 per step)
            case 1: return one ? launch_short<1, 1>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s) : launch_short<2, 1>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s);
            case 2: return one ? launch_short<1, 2>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s) : launch_short<2, 2>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s);
            case 4: return one ? launch_short<1, 4>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s) : launch_short<2, 4>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s);
            case 8: return one ? launch_short<1, 8>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s) : launch_short<2, 8>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s);
            case 16: return one ? launch_short<1, 16>(d, w, bqk, wpe, bpe, out, B, Hp, Wp, C / heads, s) : hipErrorInvalidConfiguration;
            default: return hipErrorInvalidConfiguration;
        }
    }
    float* ws = (float*)workspace;
    switch (heads) {
        case 1: return launch_long<1>(d, w, bqk, wpe, bpe, out, ws, B, Hp, Wp, C, heads, s);
        case 2: return launch_long<2>(d, w, bqk, wpe, bpe, out, ws, B, Hp, Wp, C, heads, s);
        case 4: return launch_long<4>(d, w, bqk, wpe, bpe, out, ws, B, Hp, Wp, C, heads, s);
        case 8: return launch_long<8>(d, w, bqk, wpe, bpe, out, ws, B, Hp, Wp, C, heads, s);
        default: return hipErrorInvalidConfiguration;
    }
}

}  // namespace rcx
