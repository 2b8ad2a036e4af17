// Linear-attention core of RecAttn2d's coarse level (model/recattn.py:16-28 LinearAttention1, :39-51 LinearAttention2) as
// one kernel per (image, head) -- SURVEY.md section 8f row 4.  The caller supplies the pre-activations of the grouped 1x1
// conv (a plain GEMM, left to the GEMM library) in token-major layout; this kernel does everything after it:
//
//     q = elu(qpre) + 1,  k = elu(kpre) + 1                         (:21 / :44)
//     kv = (k * s) @ (v^T * s),  kbar = mean_tokens(k),  s = n^-1/2  (:25 / :48-49 regrouped)
//     out = q^T @ kv / (q^T @ kbar + 1e-6)  +  pe                   (:26-27 / :50)
//
// LinearAttention2 materialises the n x n map A = q^T k, divides by its row mean and multiplies by v: the same function
// (the reference asserts it to 1e-4, lsnet/model/recattn.py:481-501), so both variants run this O(n d^2) form in fp32.
// Layout: all five tensors are (B, n, C) = NHWC; head h owns channels [h*D, (h+1)*D), D = C / heads.  G = 4 output columns per
// thread when D % 4 == 0 (D <= 64), else G = 1 (D <= 32).
#include <stdlib.h>
#include "rcx_opts.h"

#include "rcx_common.h"
#include "rcx_launch.h"

namespace rcx {

constexpr int LA_NT = 256;      // threads per block
constexpr int LA_TT = 64;       // tokens per LDS tile
constexpr int LA_DMAX = 64;     // largest head dimension
constexpr int LA_ITEMS = LA_DMAX * LA_DMAX / 4 / LA_NT;   // (e1, e2-quad) items per thread in phase 1: 4

__device__ __forceinline__ float la_ld(const float* p) { return *p; }
__device__ __forceinline__ float la_ld(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void la_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void la_st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ float la_ld(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void la_st(f16_t* p, float v) { *p = (f16_t)v; }
// elu(x) + 1 = x + 1 (x > 0) | exp(x) (x <= 0): no expm1 needed, and none of its cancellation
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : __expf(x); }

// G consecutive channels in one access (G = 4: 8 bytes of bf16 / 16 bytes of float32)
template <int G> __device__ __forceinline__ void la_ldv(const float* p, float (&o)[G]) { load_vec<G>(p, o); }
template <int G> __device__ __forceinline__ void la_ldv(const bf16_t* p, float (&o)[G]) { load_vec<G>(p, o); }
template <int G> __device__ __forceinline__ void la_stv(float* p, const float (&o)[G]) { store_vec<G>(p, o); }
template <int G> __device__ __forceinline__ void la_stv(bf16_t* p, const float (&o)[G]) { store_vec<G>(p, o); }
template <int G> __device__ __forceinline__ void la_ldv(const f16_t* p, float (&o)[G]) { load_vec<G>(p, o); }
template <int G> __device__ __forceinline__ void la_stv(f16_t* p, const float (&o)[G]) { store_vec<G>(p, o); }

// `pe` computed where it is used (round 3): pe = depthwise 3x3 conv (pad 1) of v + bias (model/recattn.py:27 / :50; LinearAttention.pe with
// its BatchNorm folded), the taps as a float32 (3, 3, C) pack.  w == nullptr: pe is read from memory as before.  Wp = plane width (tokens
// are the plane's pixels row by row), n = Hp * Wp.  float32 accumulation, nothing rounded in between.
struct PeConv {
    const float* w;
    const float* b;
    int Wp;
};
// G consecutive channels from absolute channel c of token `tok`; vimg = v + image offset (channel 0)
template <typename T, int G>
__device__ __forceinline__ void pe_conv3(const T* __restrict__ vimg, const PeConv& pc, int tok, int n, int C, int c, float (&pp)[G])
{
    const int y = tok / pc.Wp, x = tok - y * pc.Wp, Hp = n / pc.Wp;
#pragma unroll
    for (int j = 0; j < G; ++j) pp[j] = pc.b ? pc.b[c + j] : 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= Hp) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= pc.Wp) continue;
            float vv[G], ww[G];
            la_ldv<G>(vimg + (size_t)(yy * pc.Wp + xx) * C + c, vv);
            la_ldv<G>(pc.w + (size_t)((dy + 1) * 3 + dx + 1) * C + c, ww);
#pragma unroll
            for (int j = 0; j < G; ++j) pp[j] = fmaf(ww[j], vv[j], pp[j]);
        }
    }
}

template <typename T, int G>
__global__ void __launch_bounds__(LA_NT)
k_linattn_core(const T* __restrict__ qpre, const T* __restrict__ kpre, const T* __restrict__ v, const T* __restrict__ pe,
               T* __restrict__ out, int n, int C, int heads)
{
    __shared__ float a_s[LA_TT][LA_DMAX + 1];                 // k tile (phase 1) / q tile (phase 2); +1: odd pitch for column reads
    __shared__ __attribute__((aligned(16))) float v_s[LA_TT][LA_DMAX];
    __shared__ __attribute__((aligned(16))) float kv_s[LA_DMAX][LA_DMAX];
    __shared__ float kbar_s[LA_DMAX];
    __shared__ float den_s[LA_TT];
    const int D = C / heads, Q = D / G;                        // Q = column groups per head row
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const size_t base = (size_t)b * n * C + (size_t)h * D;
    const int tid = threadIdx.x;
    const float s2 = 1.f / (float)n;                            // s * s
    const int items = D * Q;                                    // (e1, e2-quad) pairs

    // ---- phase 1: kv = sum_t k[t][e1] * v[t][e2] ; kbar = mean_t k[t][e1]
    float acc[LA_ITEMS][G];
#pragma unroll
    for (int j = 0; j < LA_ITEMS; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[j][g] = 0.f;
    float ksum = 0.f;
    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        __syncthreads();
        for (int i = tid; i < tt * Q; i += LA_NT) {
            const int t = i / Q, e = (i - t * Q) * G;
            const size_t g = base + (size_t)(t0 + t) * C + e;
            float kk[G], vv[G];
            la_ldv<G>(kpre + g, kk);
            la_ldv<G>(v + g, vv);
#pragma unroll
            for (int j = 0; j < G; ++j) { a_s[t][e + j] = elu1(kk[j]); v_s[t][e + j] = vv[j]; }
        }
        const int tt8 = (tt + 7) & ~7;
        for (int i = tt * Q + tid; i < tt8 * Q; i += LA_NT) {         // zero tail: the token loops below run in steps of 8
            const int t = i / Q, e = (i - t * Q) * G;
#pragma unroll
            for (int j = 0; j < G; ++j) { a_s[t][e + j] = 0.f; v_s[t][e + j] = 0.f; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < LA_ITEMS; ++j) {
            const int it = tid + j * LA_NT;
            if (it < items) {
                const int e1 = it / Q, q4 = (it - e1 * Q) * G;
                for (int t8 = 0; t8 < tt8; t8 += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float kk = a_s[t8 + u][e1];
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[j][g] = fmaf(kk, v_s[t8 + u][q4 + g], acc[j][g]);
                    }
                }
            }
        }
        if (tid < D) {
            for (int t8 = 0; t8 < tt8; t8 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) ksum += a_s[t8 + u][tid];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LA_ITEMS; ++j) {
        const int it = tid + j * LA_NT;
        if (it < items) {
            const int e1 = it / Q, q4 = (it - e1 * Q) * G;
#pragma unroll
            for (int g = 0; g < G; ++g) kv_s[e1][q4 + g] = acc[j][g] * s2;
        }
    }
    if (tid < D) kbar_s[tid] = ksum / (float)n;

    // ---- phase 2: out[t][e2] = sum_e1 q[t][e1] * kv[e1][e2] / (sum_e1 q[t][e1] * kbar[e1] + 1e-6) + pe[t][e2]
    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        __syncthreads();
        for (int i = tid; i < tt * Q; i += LA_NT) {
            const int t = i / Q, e = (i - t * Q) * G;
            float qq[G];
            la_ldv<G>(qpre + base + (size_t)(t0 + t) * C + e, qq);
#pragma unroll
            for (int j = 0; j < G; ++j) a_s[t][e + j] = elu1(qq[j]);
        }
        __syncthreads();
        if (tid < tt) {
            float d = 0.f;
#pragma unroll 4
            for (int e = 0; e < D; ++e) d = fmaf(a_s[tid][e], kbar_s[e], d);
            den_s[tid] = d + 1e-6f;
        }
        __syncthreads();
        for (int it = tid; it < tt * Q; it += LA_NT) {
            const int t = it / Q, q4 = (it - t * Q) * G;
            float o[G];
#pragma unroll
            for (int g = 0; g < G; ++g) o[g] = 0.f;
#pragma unroll 4
            for (int e = 0; e < D; ++e) {
                const float qq = a_s[t][e];
#pragma unroll
                for (int g = 0; g < G; ++g) o[g] = fmaf(qq, kv_s[e][q4 + g], o[g]);
            }
            const float inv = 1.f / den_s[t];
            const size_t gi = base + (size_t)(t0 + t) * C + q4;
            float pp[G];
            la_ldv<G>(pe + gi, pp);
#pragma unroll
            for (int g = 0; g < G; ++g) o[g] = fmaf(o[g], inv, pp[g]);
            la_stv<G>(out + gi, o);
        }
    }
}


// Register-tiled form for head dimensions that are multiples of 4 (every A-series model): phase 1 gives each thread a 4x4
// block of k v^T over a 1/TG share of the tokens (two 16-byte LDS reads per 16 FMAs; the TG partial blocks are then summed in
// a fixed order), phase 2 a (two tokens) x (four columns) block of the output (six 16-byte reads per 32 FMAs).

// DM = 32 or 64: capacity of the LDS tiles (head dimensions up to 32 take 21 KB instead of 50 KB: 7 blocks per CU instead of 3)
template <typename T, int DM>
__global__ void __launch_bounds__(LA_NT)
k_linattn_core4(const T* __restrict__ qpre, const T* __restrict__ kpre, const T* __restrict__ v, const T* __restrict__ pe,
                T* __restrict__ out, int n, int C, int heads, PeConv pc)
{
    __shared__ __attribute__((aligned(16))) float a_s[LA_TT][DM + 4];      // + 4: rows stay 16-byte aligned, odd multiple of 16 B
    __shared__ __attribute__((aligned(16))) float v_s[LA_TT][DM];
    __shared__ __attribute__((aligned(16))) float kv_s[DM][DM];
    __shared__ float kbar_s[DM];
    __shared__ float den_s[LA_TT];
    const int D = C / heads, Q = D / 4;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const size_t base = (size_t)b * n * C + (size_t)h * D;
    const int tid = threadIdx.x;
    const float s2 = 1.f / (float)n;
    const int NI = Q * Q;                                         // 4x4 blocks of kv (<= 256)
    int TG = 1;                                                   // token groups: power of two, <= 8, NI * TG <= 256
    while (TG < 8 && NI * TG * 2 <= LA_NT) TG *= 2;
    const int tg = tid / NI, blk = tid - tg * NI;
    const bool p1 = tid < NI * TG;
    const int e1q = blk / Q, e2q = blk - e1q * Q;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float ksum = 0.f;
    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        const int tt8 = (tt + 7) & ~7;
        __syncthreads();
        for (int i = tid; i < tt8 * Q; i += LA_NT) {
            const int t = i / Q, e = (i - t * Q) * 4;
            float kk[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
            if (t < tt) {
                const size_t g = base + (size_t)(t0 + t) * C + e;
                la_ldv<4>(kpre + g, kk);
                la_ldv<4>(v + g, vv);
#pragma unroll
                for (int j = 0; j < 4; ++j) kk[j] = elu1(kk[j]);
            }
            *reinterpret_cast<float4*>(&a_s[t][e]) = make_float4(kk[0], kk[1], kk[2], kk[3]);   // rows past tt are zero
            *reinterpret_cast<float4*>(&v_s[t][e]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
        __syncthreads();
        if (p1) {
            for (int t = tg; t < tt8; t += TG) {
                const float4 k4 = *reinterpret_cast<const float4*>(&a_s[t][e1q * 4]);
                const float4 v4 = *reinterpret_cast<const float4*>(&v_s[t][e2q * 4]);
                const float kk[4] = {k4.x, k4.y, k4.z, k4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(kk[i], vv[j], acc[i][j]);
            }
        }
        if (tid < D) {
            for (int t8 = 0; t8 < tt8; t8 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) ksum += a_s[t8 + u][tid];
            }
        }
    }
    // sum the TG partial blocks in a fixed order
    for (int g = 0; g < TG; ++g) {
        __syncthreads();
        if (p1 && tg == g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4* dst = reinterpret_cast<float4*>(&kv_s[e1q * 4 + i][e2q * 4]);
                float4 cur = g == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *dst;
                cur.x += acc[i][0] * s2; cur.y += acc[i][1] * s2; cur.z += acc[i][2] * s2; cur.w += acc[i][3] * s2;
                *dst = cur;
            }
        }
    }
    if (tid < D) kbar_s[tid] = ksum / (float)n;

    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        const int tt2 = (tt + 1) & ~1;
        __syncthreads();
        for (int i = tid; i < tt2 * Q; i += LA_NT) {
            const int t = i / Q, e = (i - t * Q) * 4;
            float qq[4] = {0.f, 0.f, 0.f, 0.f};
            if (t < tt) {
                la_ldv<4>(qpre + base + (size_t)(t0 + t) * C + e, qq);
#pragma unroll
                for (int j = 0; j < 4; ++j) qq[j] = elu1(qq[j]);
            }
            *reinterpret_cast<float4*>(&a_s[t][e]) = make_float4(qq[0], qq[1], qq[2], qq[3]);
        }
        __syncthreads();
        if (tid < tt) {
            float d = 0.f;
#pragma unroll 4
            for (int e = 0; e < D; ++e) d = fmaf(a_s[tid][e], kbar_s[e], d);
            den_s[tid] = d + 1e-6f;
        }
        __syncthreads();
        for (int it = tid; it < (tt2 / 2) * Q; it += LA_NT) {
            const int tp = it / Q, e2 = (it - tp * Q) * 4, ta = 2 * tp, tb = ta + 1;
            float o[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[0][j] = o[1][j] = 0.f;
            for (int eq = 0; eq < Q; ++eq) {
                const float4 qa4 = *reinterpret_cast<const float4*>(&a_s[ta][eq * 4]);
                const float4 qb4 = *reinterpret_cast<const float4*>(&a_s[tb][eq * 4]);
                const float qa[4] = {qa4.x, qa4.y, qa4.z, qa4.w}, qb[4] = {qb4.x, qb4.y, qb4.z, qb4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 r = *reinterpret_cast<const float4*>(&kv_s[eq * 4 + i][e2]);
                    o[0][0] = fmaf(qa[i], r.x, o[0][0]); o[0][1] = fmaf(qa[i], r.y, o[0][1]);
                    o[0][2] = fmaf(qa[i], r.z, o[0][2]); o[0][3] = fmaf(qa[i], r.w, o[0][3]);
                    o[1][0] = fmaf(qb[i], r.x, o[1][0]); o[1][1] = fmaf(qb[i], r.y, o[1][1]);
                    o[1][2] = fmaf(qb[i], r.z, o[1][2]); o[1][3] = fmaf(qb[i], r.w, o[1][3]);
                }
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int t = ta + w;
                if (t < tt) {
                    const float inv = 1.f / den_s[t];
                    const size_t gi = base + (size_t)(t0 + t) * C + e2;
                    float pp[4], r[4];
                    if (pc.w) pe_conv3<T, 4>(v + (size_t)b * n * C, pc, t0 + t, n, C, h * D + e2, pp);
                    else la_ldv<4>(pe + gi, pp);
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = fmaf(o[w][j], inv, pp[j]);
                    la_stv<4>(out + gi, r);
                }
            }
        }
    }
}

// ---- matrix-core form for head dimension 32 and 16-bit I/O (every head of the A-series: 64/2 = 128/4 = 256/8 = 512/16 = 32).
// The two products of the core ARE dense contractions -- kv = k^T v over the tokens, out = q kv over the head dimension -- so they go
// to MFMA (v_mfma_f32_32x32x16_bf16 / _f16), float32 accumulation; q and k are rounded to the I/O type after the activation and kv
// before the second product, which is what the reference's own 16-bit autocast run does with these matmuls.
// LM_NW waves per (image, head), the tokens dealt over them:
//   phase 1: 16 tokens per MFMA, A[e1][token] = k, B[token][e2] = v: lane (r, h) loads tokens 8h .. 8h+7 of column r (two tokens x 64
//            contiguous bytes per load instruction); column sums of k for the normaliser ride along on the vector pipe;
//   phase 2: 32 tokens per pair of MFMAs.  The kv accumulator tile has its column e2 on the lane and its rows e1 in the registers,
//            so it is the B operand of the second product without any lane movement (cdna_hip_programming.md, "An accumulator tile
//            as the next MFMA's operand"): registers 8s .. 8s+7 converted pairwise are the fragment of k-step s, whose element j of
//            lane half h is row 16s + 8(j>>2) + 4h + (j&3) -- the q fragment is loaded in that same permuted order (two 8-byte
//            runs per step).  The normaliser q . kbar is a 16-term partial per lane, the two halves meet in LDS.
template <typename T> struct Mf;
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <> struct Mf<bf16_t> {
    typedef __bf16 frag __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ __bf16 cvt(float f) { return (__bf16)f; }
    static __device__ __forceinline__ __bf16 raw(bf16_t v) { return __builtin_bit_cast(__bf16, v); }
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mf<f16_t> {
    typedef _Float16 frag __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ _Float16 cvt(float f) { return (_Float16)f; }
    static __device__ __forceinline__ _Float16 raw(f16_t v) { return v; }
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

#ifndef RCX_LM_NW
#define RCX_LM_NW 4
#endif
constexpr int LM_NW = RCX_LM_NW;    // waves per (image, head): the tokens are dealt over them (phase 1 partial sums meet in LDS)

template <typename T>
__global__ void __launch_bounds__(LM_NW * 64)
k_linattn_mfma(const T* __restrict__ qpre, const T* __restrict__ kpre, const T* __restrict__ v, const T* __restrict__ pe,
               T* __restrict__ out, int n, int C, int heads, PeConv pc)
{
    typedef typename Mf<T>::frag frag;
    __shared__ float kvp_s[LM_NW][16][64];                    // partial kv tiles in accumulator layout
    __shared__ float ksum_s[LM_NW][2][32];
    __shared__ float kbar_s[32];
    __shared__ float den_s[LM_NW][2][32];                      // per wave: the two halves of its 32 tokens' normalisers
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
    const size_t base = (size_t)b * n * C + (size_t)hd * 32;

    // ---- phase 1: kv[e1][e2] = sum_t k[t][e1] v[t][e2] ; column sums of k.  Wave w takes the token steps w, w + 4, ...
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float ksum = 0.f;
#pragma unroll 2
    for (int t0 = 16 * w; t0 < n; t0 += 16 * LM_NW) {
        float kk[8];
        frag A, B;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                             // all sixteen loads first, then the arithmetic
            const int t = t0 + 8 * h + j;
            const bool ok = t < n;
            const size_t g = base + (size_t)(ok ? t : n - 1) * C + r;
            kk[j] = ok ? la_ld(kpre + g) : -1e30f;               // elu1(-1e30) = 0: a token past the end contributes nothing
            B[j] = ok ? Mf<T>::raw(v[g]) : Mf<T>::cvt(0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = elu1(kk[j]);
            ksum += e;
            A[j] = Mf<T>::cvt(e);
        }
        acc = Mf<T>::mma(A, B, acc);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) kvp_s[w][i][lane] = acc[i];
    ksum_s[w][h][r] = ksum;
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < LM_NW; ++ww) t += ksum_s[ww][0][tid] + ksum_s[ww][1][tid];
        kbar_s[tid] = t / (float)n;
    }
    const float s2 = 1.f / (float)n;
    frag KV0, KV1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < LM_NW; ++ww) t += kvp_s[ww][i][lane];
        if (i < 8) KV0[i] = Mf<T>::cvt(t * s2);
        else KV1[i - 8] = Mf<T>::cvt(t * s2);
    }
    __syncthreads();

    // ---- phase 2: out[t][e2] = (q[t] . kv[.][e2]) / (q[t] . kbar + 1e-6) + pe[t][e2]; wave w takes the 32-token groups w, w + 4, ...
    for (int g0 = 32 * w; g0 < n; g0 += 32 * LM_NW) {
        const int t = g0 + r;
        const size_t gq = base + (size_t)(t < n ? t : n - 1) * C;
        frag A0, A1;
        float dpart = 0.f;
        float qv[4][4];
#pragma unroll
        for (int sj = 0; sj < 4; ++sj) la_ldv<4>(qpre + gq + 16 * (sj >> 1) + 8 * (sj & 1) + 4 * h, qv[sj]);   // rows 16s + 8(j>>2) + 4h + (j&3) of kv
        float pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tok = g0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (pc.w) {
                float p1[1];
                pe_conv3<T, 1>(v + (size_t)b * n * C, pc, tok < n ? tok : n - 1, n, C, hd * 32 + r, p1);
                pv[i] = p1[0];
            } else pv[i] = la_ld(pe + base + (size_t)(tok < n ? tok : n - 1) * C + r);
        }
#pragma unroll
        for (int sj = 0; sj < 4; ++sj) {
            const int e0 = 16 * (sj >> 1) + 8 * (sj & 1) + 4 * h;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float qq = elu1(qv[sj][i]);
                dpart = fmaf(qq, kbar_s[e0 + i], dpart);
                if (sj < 2) A0[4 * (sj & 1) + i] = Mf<T>::cvt(qq);
                else A1[4 * (sj & 1) + i] = Mf<T>::cvt(qq);
            }
        }
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
        o = Mf<T>::mma(A0, KV0, o);
        o = Mf<T>::mma(A1, KV1, o);
        den_s[w][h][r] = dpart;                                   // wave-private: LDS operations of one wave execute in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;          // token of accumulator register i; the lane is the column e2 = r
            const int tok = g0 + row;
            if (tok < n) {
                const float den = den_s[w][0][row] + den_s[w][1][row] + 1e-6f;
                la_st(out + base + (size_t)tok * C + r, o[i] / den + pv[i]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- backward of the core (engine.py:48-64 reaches it through RecAttn2d in the A-series).  With u = q kv, w = q . kbar + 1e-6,
// out = u / w + pe and g = dL/dout:
//     du = g / w ;  dw = -(g . u) / w^2 ;  dq = du kv^T + dw kbar ;  dkv = q^T du ;  dkbar = q^T dw
//     dk = (1/n) v dkv^T + dkbar / n ;  dv = (1/n) k dkv ;  dqpre = dq * elu'(qpre), dkpre = dk * elu'(kpre), elu'(x) = x > 0 ? 1 : e^x
// (dpe = g is the caller's).  One block per (image, head), three sweeps over its tokens: kv and kbar as in the forward, then dq
// with dkv / dkbar accumulated per owner thread in a fixed order (deterministic), then dk and dv.
template <typename T>
__global__ void __launch_bounds__(LA_NT)
k_linattn_bwd(const T* __restrict__ qpre, const T* __restrict__ kpre, const T* __restrict__ v, const T* __restrict__ gout,
              T* __restrict__ gq, T* __restrict__ gk, T* __restrict__ gv, int n, int C, int heads)
{
    __shared__ float a_s[LA_TT][LA_DMAX + 1];                 // k / q tile
    __shared__ float b_s[LA_TT][LA_DMAX + 1];                 // v / du tile
    __shared__ float c_s[LA_TT][LA_DMAX + 1];                 // g * u, elementwise
    __shared__ float kv_s[LA_DMAX][LA_DMAX + 1];
    __shared__ float dkv_s[LA_DMAX][LA_DMAX + 1];
    __shared__ float kbar_s[LA_DMAX], dkbar_s[LA_DMAX];
    __shared__ float w_s[LA_TT], dw_s[LA_TT];
    const int D = C / heads;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const size_t base = (size_t)b * n * C + (size_t)h * D;
    const int tid = threadIdx.x;
    const float s2 = 1.f / (float)n;
    const int items = D * D;                                   // (e1, e2) pairs, owner thread = it % LA_NT
    constexpr int MAXI = LA_DMAX * LA_DMAX / LA_NT;            // 16

    // ---- sweep 1: kv = (1/n) k^T v, kbar = mean k
    float acc[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) acc[j] = 0.f;
    float ksum = 0.f;
    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        __syncthreads();
        for (int i = tid; i < tt * D; i += LA_NT) {
            const int t = i / D, e = i - t * D;
            const size_t g = base + (size_t)(t0 + t) * C + e;
            a_s[t][e] = elu1(la_ld(kpre + g));
            b_s[t][e] = la_ld(v + g);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const int it = tid + j * LA_NT;
            if (it < items) {
                const int e1 = it / D, e2 = it - e1 * D;
                float sacc = acc[j];
                for (int t = 0; t < tt; ++t) sacc = fmaf(a_s[t][e1], b_s[t][e2], sacc);
                acc[j] = sacc;
            }
        }
        if (tid < D) for (int t = 0; t < tt; ++t) ksum += a_s[t][tid];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int it = tid + j * LA_NT;
        if (it < items) kv_s[it / D][it % D] = acc[j] * s2;
        acc[j] = 0.f;                                           // becomes the dkv accumulator
    }
    if (tid < D) kbar_s[tid] = ksum / (float)n;
    float dkb = 0.f;

    // ---- sweep 2: per token u, w, du, dw; dq -> gq; dkv += q^T du, dkbar += q^T dw
    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        __syncthreads();
        for (int i = tid; i < tt * D; i += LA_NT) {
            const int t = i / D, e = i - t * D;
            const size_t g = base + (size_t)(t0 + t) * C + e;
            a_s[t][e] = elu1(la_ld(qpre + g));
            b_s[t][e] = la_ld(gout + g);                        // g for now, du below
        }
        __syncthreads();
        // g[t][e2] * u[t][e2] with every thread (one thread per token walking all D * D products left three quarters of the block
        // idle for two thirds of the sweep), then one thread per token adds its row up
        for (int i = tid; i < tt * D; i += LA_NT) {
            const int t = i / D, e2 = i - t * D;
            float u = 0.f;
            for (int e1 = 0; e1 < D; ++e1) u = fmaf(a_s[t][e1], kv_s[e1][e2], u);
            c_s[t][e2] = b_s[t][e2] * u;
        }
        __syncthreads();
        if (tid < tt) {
            float wsum = 0.f, gu = 0.f;
            for (int e = 0; e < D; ++e) {
                wsum = fmaf(a_s[tid][e], kbar_s[e], wsum);
                gu += c_s[tid][e];
            }
            wsum += 1e-6f;
            w_s[tid] = wsum;
            dw_s[tid] = -gu / (wsum * wsum);
        }
        __syncthreads();
        for (int i = tid; i < tt * D; i += LA_NT) {             // du = g / w (in place)
            const int t = i / D, e = i - t * D;
            b_s[t][e] = b_s[t][e] / w_s[t];
        }
        __syncthreads();
        for (int i = tid; i < tt * D; i += LA_NT) {             // dq[t][e1] = sum_e2 du[t][e2] kv[e1][e2] + dw[t] kbar[e1]
            const int t = i / D, e1 = i - t * D;
            float dq = dw_s[t] * kbar_s[e1];
            for (int e2 = 0; e2 < D; ++e2) dq = fmaf(b_s[t][e2], kv_s[e1][e2], dq);
            const size_t g = base + (size_t)(t0 + t) * C + e1;
            const float xp = la_ld(qpre + g);
            la_st(gq + g, dq * (xp > 0.f ? 1.f : a_s[t][e1]));  // elu'(x) = e^x = elu(x) + 1 for x <= 0
        }
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const int it = tid + j * LA_NT;
            if (it < items) {
                const int e1 = it / D, e2 = it - e1 * D;
                float sacc = acc[j];
                for (int t = 0; t < tt; ++t) sacc = fmaf(a_s[t][e1], b_s[t][e2], sacc);
                acc[j] = sacc;
            }
        }
        if (tid < D) for (int t = 0; t < tt; ++t) dkb = fmaf(dw_s[t], a_s[t][tid], dkb);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int it = tid + j * LA_NT;
        if (it < items) dkv_s[it / D][it % D] = acc[j] * s2;    // the 1/n of kv = (1/n) k^T v
    }
    if (tid < D) dkbar_s[tid] = dkb / (float)n;

    // ---- sweep 3: dk[t][e1] = sum_e2 v[t][e2] dkv[e1][e2] + dkbar[e1] ; dv[t][e2] = sum_e1 k[t][e1] dkv[e1][e2]
    for (int t0 = 0; t0 < n; t0 += LA_TT) {
        const int tt = n - t0 < LA_TT ? n - t0 : LA_TT;
        __syncthreads();
        for (int i = tid; i < tt * D; i += LA_NT) {
            const int t = i / D, e = i - t * D;
            const size_t g = base + (size_t)(t0 + t) * C + e;
            a_s[t][e] = elu1(la_ld(kpre + g));
            b_s[t][e] = la_ld(v + g);
        }
        __syncthreads();
        for (int i = tid; i < tt * D; i += LA_NT) {
            const int t = i / D, e = i - t * D;
            float dk = dkbar_s[e], dv = 0.f;
            for (int f = 0; f < D; ++f) {
                dk = fmaf(b_s[t][f], dkv_s[e][f], dk);
                dv = fmaf(a_s[t][f], dkv_s[f][e], dv);
            }
            const size_t g = base + (size_t)(t0 + t) * C + e;
            const float xp = la_ld(kpre + g);
            la_st(gk + g, dk * (xp > 0.f ? 1.f : a_s[t][e]));
            la_st(gv + g, dv);
        }
    }
}

// ---- the same backward on the matrix cores (head dimension 32, 16-bit I/O): every product above is a 32-wide contraction.
// Layout conventions of v_mfma_f32_32x32x16 as in k_linattn_mfma: lane = (r, h) = (lane & 31, lane >> 5); an A fragment holds
// A[row r][k = 8h + j], a B fragment B[k = 8h + j][col r], accumulator register i holds C[rho(h, i)][r], rho(h, i) = 8 (i >> 2) + 4 h + (i & 3).
// Two orientations are used.  TOKEN PER LANE (lane r owns token g0 + r and reads its features rho(h, 0..15) as four 4-element
// vectors; the products are taken transposed, C[feature][token], so that per-token scalars -- w, g . u, dw -- are lane-local up to
// one exchange between the two halves): u^T = kv^T q^T, dq^T = kv du^T, dv^T = dkv^T k^T, dk^T = dkv v^T; the 32 x 32 operand is the A
// fragment, either the accumulator registers themselves (kv^T, dkv^T: k runs over rho) or their LDS transpose (kv, dkv).  FEATURE PER
// LANE (lane r owns feature r, k runs over 16 tokens, as the forward's first phase): kv = k^T v and dkv = q^T du, summed over the
// block's waves through LDS in a fixed order.  Operands are rounded to the I/O type (as the forward's), accumulation is float32.
template <typename T>
__global__ void __launch_bounds__(LM_NW * 64)
k_linattn_bwd_mfma(const T* __restrict__ qpre, const T* __restrict__ kpre, const T* __restrict__ v, const T* __restrict__ gout,
                   T* __restrict__ gq, T* __restrict__ gk, T* __restrict__ gv, int n, int C, int heads)
{
    typedef typename Mf<T>::frag frag;
    __shared__ float part_s[LM_NW][16][64];                   // partial kv / dkv tiles in accumulator layout
    __shared__ float mat_s[32][33];                           // kv, later dkv: [e1][e2] for the transposed reads
    __shared__ float sum_s[LM_NW][2][32];                     // partial column sums (k, later dw * q)
    __shared__ float kbar_s[32], dkbar_s[32];
    __shared__ float xa_s[LM_NW][2][32], xb_s[LM_NW][2][32];  // per wave: halves of w and of g . u
    __shared__ float w_s[LM_NW][32], dw_s[LM_NW][32];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
    const size_t base = (size_t)b * n * C + (size_t)hd * 32;
    const float s2 = 1.f / (float)n;
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // ---- phase 1 (feature per lane): kv = (1/n) k^T v, kbar = mean k
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float csum = 0.f;
#pragma unroll 2
    for (int t0 = 16 * w; t0 < n; t0 += 16 * LM_NW) {
        float kk[8];
        frag A, B;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + 8 * h + j;
            const bool ok = t < n;
            const size_t g = base + (size_t)(ok ? t : n - 1) * C + r;
            kk[j] = ok ? la_ld(kpre + g) : -1e30f;
            B[j] = ok ? Mf<T>::raw(v[g]) : Mf<T>::cvt(0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = elu1(kk[j]);
            csum += e;
            A[j] = Mf<T>::cvt(e);
        }
        acc = Mf<T>::mma(A, B, acc);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) part_s[w][i][lane] = acc[i];
    sum_s[w][h][r] = csum;
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < LM_NW; ++ww) t += sum_s[ww][0][tid] + sum_s[ww][1][tid];
        kbar_s[tid] = t * s2;
    }
    frag KVa0, KVa1;                                          // kv^T as the A operand: the accumulator registers (k over rho)
    {
        float kvv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < LM_NW; ++ww) t += part_s[ww][i][lane];
            kvv[i] = t * s2;
            if (i < 8) KVa0[i] = Mf<T>::cvt(kvv[i]);
            else KVa1[i - 8] = Mf<T>::cvt(kvv[i]);
        }
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mat_s[8 * (i >> 2) + 4 * h + (i & 3)][r] = kvv[i];
        }
    }
    __syncthreads();
    frag KVr0, KVr1;                                          // kv as the A operand: row r, k over rho
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float t = mat_s[r][8 * (i >> 2) + 4 * h + (i & 3)];
        if (i < 8) KVr0[i] = Mf<T>::cvt(t);
        else KVr1[i - 8] = Mf<T>::cvt(t);
    }

    // ---- phase 2: per 32-token group (wave w takes the groups w, w + LM_NW, ...)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;                // becomes dkv
    float dsum = 0.f;                                         // dkbar[r], this lane's tokens
    for (int g0 = 32 * w; g0 < n; g0 += 32 * LM_NW) {
        // (a) token per lane
        const int t = g0 + r;
        const bool tok_ok = t < n;
        const size_t gt = base + (size_t)(tok_ok ? t : n - 1) * C;
        float qv[4][4], gvv[4][4], qe[16], gg[16];
#pragma unroll
        for (int sj = 0; sj < 4; ++sj) {
            la_ldv<4>(qpre + gt + 8 * sj + 4 * h, qv[sj]);
            la_ldv<4>(gout + gt + 8 * sj + 4 * h, gvv[sj]);
        }
        frag Q0, Q1;
        float wpart = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            qe[i] = elu1(qv[i >> 2][i & 3]);
            gg[i] = tok_ok ? gvv[i >> 2][i & 3] : 0.f;
            wpart = fmaf(qe[i], kbar_s[8 * (i >> 2) + 4 * h + (i & 3)], wpart);
            if (i < 8) Q0[i] = Mf<T>::cvt(qe[i]);
            else Q1[i - 8] = Mf<T>::cvt(qe[i]);
        }
        f32x16 uT;
#pragma unroll
        for (int i = 0; i < 16; ++i) uT[i] = 0.f;
        uT = Mf<T>::mma(KVa0, Q0, uT);
        uT = Mf<T>::mma(KVa1, Q1, uT);
        float gpart = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) gpart = fmaf(gg[i], uT[i], gpart);
        xa_s[w][h][r] = wpart;
        xb_s[w][h][r] = gpart;
        wave_sync();
        const float wt = xa_s[w][0][r] + xa_s[w][1][r] + 1e-6f;
        const float gu = xb_s[w][0][r] + xb_s[w][1][r];
        const float dwt = -gu / (wt * wt);
        if (h == 0) { w_s[w][r] = tok_ok ? wt : 1.f; dw_s[w][r] = tok_ok ? dwt : 0.f; }
        frag DU0, DU1;
        const float rw = 1.f / wt;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < 8) DU0[i] = Mf<T>::cvt(gg[i] * rw);
            else DU1[i - 8] = Mf<T>::cvt(gg[i] * rw);
        }
        f32x16 dqT;
#pragma unroll
        for (int i = 0; i < 16; ++i) dqT[i] = 0.f;
        dqT = Mf<T>::mma(KVr0, DU0, dqT);
        dqT = Mf<T>::mma(KVr1, DU1, dqT);
        if (tok_ok) {
#pragma unroll
            for (int sj = 0; sj < 4; ++sj) {
                float o[4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = 4 * sj + ii;
                    const float dq = fmaf(dwt, kbar_s[8 * sj + 4 * h + ii], dqT[i]);
                    o[ii] = dq * (qv[sj][ii] > 0.f ? 1.f : qe[i]);      // elu'(x) = e^x = elu(x) + 1 for x <= 0
                }
                store_vec<4>(gq + gt + 8 * sj + 4 * h, o);
            }
        }
        wave_sync();                                           // w_s / dw_s of this group are visible to the wave
        // (b) feature per lane: dkv += q^T du, dkbar += q^T dw, 16 tokens per product
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            frag A, B;
            float qq[8], g8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tl = 16 * hh + 8 * h + j, tk = g0 + tl;
                const bool ok = tk < n;
                const size_t g = base + (size_t)(ok ? tk : n - 1) * C + r;
                qq[j] = ok ? la_ld(qpre + g) : -1e30f;
                g8[j] = ok ? la_ld(gout + g) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tl = 16 * hh + 8 * h + j;
                const float e = elu1(qq[j]);
                dsum = fmaf(dw_s[w][tl], e, dsum);
                A[j] = Mf<T>::cvt(e);
                B[j] = Mf<T>::cvt(g8[j] / w_s[w][tl]);
            }
            acc = Mf<T>::mma(A, B, acc);
        }
        wave_sync();                                           // before the next group overwrites the wave's exchange lines
    }
    __syncthreads();                                           // every wave is done with part_s / mat_s as kv
#pragma unroll
    for (int i = 0; i < 16; ++i) part_s[w][i][lane] = acc[i];
    sum_s[w][h][r] = dsum;
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < LM_NW; ++ww) t += sum_s[ww][0][tid] + sum_s[ww][1][tid];
        dkbar_s[tid] = t * s2;
    }
    frag DKa0, DKa1;                                          // dkv^T as the A operand (accumulator registers)
    {
        float dk[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < LM_NW; ++ww) t += part_s[ww][i][lane];
            dk[i] = t * s2;                                     // the 1/n of kv = (1/n) k^T v
            if (i < 8) DKa0[i] = Mf<T>::cvt(dk[i]);
            else DKa1[i - 8] = Mf<T>::cvt(dk[i]);
        }
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mat_s[8 * (i >> 2) + 4 * h + (i & 3)][r] = dk[i];
        }
    }
    __syncthreads();
    frag DKr0, DKr1;                                          // dkv as the A operand: row r, k over rho
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float t = mat_s[r][8 * (i >> 2) + 4 * h + (i & 3)];
        if (i < 8) DKr0[i] = Mf<T>::cvt(t);
        else DKr1[i - 8] = Mf<T>::cvt(t);
    }

    // ---- phase 3 (token per lane): dk = v dkv^T + dkbar, dv = k dkv
    for (int g0 = 32 * w; g0 < n; g0 += 32 * LM_NW) {
        const int t = g0 + r;
        const bool tok_ok = t < n;                             // (no divergence around the matrix instructions: clamp, and mask the stores)
        const size_t gt = base + (size_t)(tok_ok ? t : n - 1) * C;
        float kv4[4][4], vv4[4][4], ke[16];
#pragma unroll
        for (int sj = 0; sj < 4; ++sj) {
            la_ldv<4>(kpre + gt + 8 * sj + 4 * h, kv4[sj]);
            la_ldv<4>(v + gt + 8 * sj + 4 * h, vv4[sj]);
        }
        frag K0, K1, V0, V1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ke[i] = elu1(kv4[i >> 2][i & 3]);
            if (i < 8) { K0[i] = Mf<T>::cvt(ke[i]); V0[i] = Mf<T>::cvt(vv4[i >> 2][i & 3]); }
            else { K1[i - 8] = Mf<T>::cvt(ke[i]); V1[i - 8] = Mf<T>::cvt(vv4[i >> 2][i & 3]); }
        }
        f32x16 dvT, dkT;
#pragma unroll
        for (int i = 0; i < 16; ++i) { dvT[i] = 0.f; dkT[i] = 0.f; }
        dvT = Mf<T>::mma(DKa0, K0, dvT);
        dvT = Mf<T>::mma(DKa1, K1, dvT);
        dkT = Mf<T>::mma(DKr0, V0, dkT);
        dkT = Mf<T>::mma(DKr1, V1, dkT);
        if (tok_ok) {
#pragma unroll
        for (int sj = 0; sj < 4; ++sj) {
            float ok_[4], ov[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * sj + ii;
                const float dk = dkT[i] + dkbar_s[8 * sj + 4 * h + ii];
                ok_[ii] = dk * (kv4[sj][ii] > 0.f ? 1.f : ke[i]);
                ov[ii] = dvT[i];
            }
            store_vec<4>(gk + gt + 8 * sj + 4 * h, ok_);
            store_vec<4>(gv + gt + 8 * sj + 4 * h, ov);
        }
        }
    }
}

hipError_t linattn_core_bwd(const void* qpre, const void* kpre, const void* v, const void* gout, void* gq, void* gk, void* gv,
                            int B, int n, int C, int heads, int dtype, hipStream_t s)
{
    const dim3 grid((unsigned)(B * heads)), block(LA_NT);
    {
        const char* m = rcx::opt::value(rcx::opt::ATTN_MFMA);                     // A/B knob: 0 = the vector-pipe kernel
        if (C / heads == 32 && dtype != 0 && !(m && *m == '0')) {
            if (dtype == 1) hipLaunchKernelGGL((k_linattn_bwd_mfma<bf16_t>), grid, dim3(LM_NW * 64), 0, s, (const bf16_t*)qpre, (const bf16_t*)kpre, (const bf16_t*)v, (const bf16_t*)gout, (bf16_t*)gq, (bf16_t*)gk, (bf16_t*)gv, n, C, heads);
            else hipLaunchKernelGGL((k_linattn_bwd_mfma<f16_t>), grid, dim3(LM_NW * 64), 0, s, (const f16_t*)qpre, (const f16_t*)kpre, (const f16_t*)v, (const f16_t*)gout, (f16_t*)gq, (f16_t*)gk, (f16_t*)gv, n, C, heads);
            return hipGetLastError();
        }
    }
#define RCX_LAB(T) hipLaunchKernelGGL((k_linattn_bwd<T>), grid, block, 0, s, (const T*)qpre, (const T*)kpre, (const T*)v, (const T*)gout, \
                                      (T*)gq, (T*)gk, (T*)gv, n, C, heads)
    if (dtype == 1) RCX_LAB(bf16_t);
    else if (dtype == 2) RCX_LAB(f16_t);
    else RCX_LAB(float);
#undef RCX_LAB
    return hipGetLastError();
}

// pew != nullptr: pe is computed inside the kernel from v (pe is ignored); only the kernels of head dimensions that are multiples of four
// have that form (every head of the A-series is 32 wide)
// ... and only the vector-pipe kernel gains from it: on the matrix-core kernel (head dimension 32, 16-bit I/O, >= 512 tokens) a lane owns ONE
// channel of 16 tokens, its nine taps are scalar loads, and the unit of RecNeXt-A3's stage 0 went from 239 to 394 us; the shorter
// sequences (vector-pipe kernel, four channels per thread) went 80.4 -> 74.2 us (49 tokens x 256 channels), 118 -> 116.5, 99 -> 97.5.
static bool linattn_uses_mfma(int n, int C, int heads, int dtype)
{
    const char* m = rcx::opt::value(rcx::opt::ATTN_MFMA);
    return C / heads == 32 && dtype != 0 && n >= 512 && !(m && *m == '0');
}

bool linattn_core_fuses_pe(int n, int C, int heads, int dtype)
{
    const char* old = rcx::opt::value(rcx::opt::ATTN_SCALAR);
    return ((C / heads) % 4) == 0 && !(old && *old == '1') && !linattn_uses_mfma(n, C, heads, dtype);
}

hipError_t linattn_core(const void* qpre, const void* kpre, const void* v, const void* pe, void* out,
                        int B, int n, int C, int heads, int dtype, hipStream_t s, const float* pew, const float* peb, int Wp)
{
    const dim3 grid((unsigned)(B * heads)), block(LA_NT);
    const bool wide = ((C / heads) % 4) == 0;
    const PeConv pc{pew, peb, Wp > 0 ? Wp : 1};
    if (pew && !linattn_core_fuses_pe(n, C, heads, dtype)) return hipErrorInvalidConfiguration;
    {
        const char* m = rcx::opt::value(rcx::opt::ATTN_MFMA);                     // A/B knob: 0 = the vector-pipe kernels for every head dimension
        // Measured (batch 256, bf16): 784 tokens 88 -> 61 us with 4 waves per head; 196 / 49 / 16 tokens no faster than the vector-pipe kernel
        // (36 / 23 / 31 against 36 / 20 / 20 us: too few tokens per head to amortise the partial-sum exchange) -- so only the long sequences
        (void)m;
        if (linattn_uses_mfma(n, C, heads, dtype)) {
            if (dtype == 1) hipLaunchKernelGGL((k_linattn_mfma<bf16_t>), grid, dim3(LM_NW * 64), 0, s, (const bf16_t*)qpre, (const bf16_t*)kpre, (const bf16_t*)v, (const bf16_t*)pe, (bf16_t*)out, n, C, heads, pc);
            else hipLaunchKernelGGL((k_linattn_mfma<f16_t>), grid, dim3(LM_NW * 64), 0, s, (const f16_t*)qpre, (const f16_t*)kpre, (const f16_t*)v, (const f16_t*)pe, (f16_t*)out, n, C, heads, pc);
            return hipGetLastError();
        }
    }
#define RCX_LA_LAUNCH(T, G) hipLaunchKernelGGL((k_linattn_core<T, G>), grid, block, 0, s, (const T*)qpre, (const T*)kpre, (const T*)v, \
                                               (const T*)pe, (T*)out, n, C, heads)
    const char* old = rcx::opt::value(rcx::opt::ATTN_SCALAR);                 // A/B knob: the untiled kernel for every head dimension
    const bool tiled = wide && !(old && *old == '1');
    if (tiled) {
#define RCX_LA4(T, DM) hipLaunchKernelGGL((k_linattn_core4<T, DM>), grid, block, 0, s, (const T*)qpre, (const T*)kpre, (const T*)v, (const T*)pe, (T*)out, n, C, heads, pc)
        const bool small = C / heads <= 32;
        if (dtype == 1) { if (small) RCX_LA4(bf16_t, 32); else RCX_LA4(bf16_t, 64); }
        else if (dtype == 2) { if (small) RCX_LA4(f16_t, 32); else RCX_LA4(f16_t, 64); }
        else { if (small) RCX_LA4(float, 32); else RCX_LA4(float, 64); }
#undef RCX_LA4
    }
    else if (dtype == 1) { if (wide) RCX_LA_LAUNCH(bf16_t, 4); else RCX_LA_LAUNCH(bf16_t, 1); }
    else if (dtype == 2) { if (wide) RCX_LA_LAUNCH(f16_t, 4); else RCX_LA_LAUNCH(f16_t, 1); }
    else { if (wide) RCX_LA_LAUNCH(float, 4); else RCX_LA_LAUNCH(float, 1); }
#undef RCX_LA_LAUNCH
    return hipGetLastError();
}

}  // namespace rcx
