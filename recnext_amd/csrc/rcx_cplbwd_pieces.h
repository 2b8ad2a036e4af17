// Adjoint pieces of the channel-per-lane backward kernels (rcx_cplbwd.hip: whole-block backward of the 14x14 / 7x7 blocks;
// rcx_cplwgrad.hip: tiled weight gradient of the larger planes).  See rcx_cplbwd.hip for the formulation.
#pragma once
#include "rcx_cpl14_pieces.h"

namespace rcx {
namespace cplbwd {

using namespace cpl14;
using lanes::sfor;
using lanes::IC;

__device__ __forceinline__ float elem(const f32x2& p, int half) { return half ? p.y : p.x; }
template <int NP> __device__ __forceinline__ float at(const f32x2 (&row)[NP], int i) { return (i & 1) ? row[i >> 1].y : row[i >> 1].x; }
template <int NP> __device__ __forceinline__ void add_at(f32x2 (&row)[NP], int i, float v)
{
    if (i & 1) row[i >> 1].y += v;
    else row[i >> 1].x += v;
}
template <int NP> __device__ __forceinline__ void fma_at(f32x2 (&row)[NP], int i, float a, float b)
{
    if (i & 1) row[i >> 1].y = fmaf(a, b, row[i >> 1].y);
    else row[i >> 1].x = fmaf(a, b, row[i >> 1].x);
}

// weight-gradient accumulators of one stride-1 5x5 conv: E[u] = tap pairs (0,1)(2,3)(4,-) from even columns, O[u] = (-,0)(1,2)(3,4)
struct WAcc {
    f32x2 E[5][3], O[5][3];
    f32x2 bs;
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) { E[u][k] = f32x2{0.f, 0.f}; O[u][k] = f32x2{0.f, 0.f}; }
        bs = f32x2{0.f, 0.f};
    }
    __device__ __forceinline__ float tap(int u, int v) const
    {
        switch (v) {
        case 0: return E[u][0].x + O[u][0].y;
        case 1: return E[u][0].y + O[u][1].x;
        case 2: return E[u][1].x + O[u][1].y;
        case 3: return E[u][1].y + O[u][2].x;
        default: return E[u][2].x + O[u][2].y;
        }
    }
    __device__ __forceinline__ float bias() const { return bs.x + bs.y; }
};
// the shared stride-2 conv: tap pairs (0,1)(2,3)(4,-), one set
struct DAcc {
    f32x2 a[5][3];
    f32x2 bs;
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) a[u][k] = f32x2{0.f, 0.f};
        bs = f32x2{0.f, 0.f};
    }
    __device__ __forceinline__ float tap(int u, int v) const { return (v & 1) ? a[u][v >> 1].y : a[u][v >> 1].x; }
    __device__ __forceinline__ float bias() const { return bs.x + bs.y; }
};

// one gradient row t of a stride-1 conv's output against the rows of its input: gW[u][v] += sum_c g[t][c] * T[t+u-2][c+v-2].
// g: the N-wide row as pairs (odd N: last .y = 0); T_of(r): the pairs of input row r (pad column zero).
template <int N, class RowOf>
__device__ __forceinline__ void wgrad_row(const f32x2 (&g)[(N + 1) / 2], int t, RowOf&& T_of, WAcc& a)
{
    constexpr int NP = (N + 1) / 2;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const f32x2 gv = splat(at<NP>(g, c));
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int r = t + u - 2;
            if (r < 0 || r >= N) continue;
            const f32x2(&T)[NP] = T_of(r);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int p = (c & 1) ? (c - 3) / 2 + k : c / 2 - 1 + k;
                if (p < 0 || p >= NP) continue;
                if (c & 1) a.O[u][k] = pfma(gv, T[p], a.O[u][k]);
                else a.E[u][k] = pfma(gv, T[p], a.E[u][k]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) a.bs = a.bs + g[j];
}

// one input row r of the stride-2 conv against the gradient of its output: gW[u][v] += sum_i G[o][i] * x[r][2i+v-2], u = r-2o+2
template <int NI, int NO>
__device__ __forceinline__ void wgrad2_row(const f32x2 (&xr)[(NI + 1) / 2], int r, const f32x2 (&G)[NO][(NO + 1) / 2], DAcc& d)
{
    constexpr int PI = (NI + 1) / 2, PO = (NO + 1) / 2;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int u = r - 2 * o + 2;
            if (u < 0 || u > 4) continue;
            const f32x2 gv = splat(at<PO>(G[o], i));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int p = i - 1 + k;
                if (p < 0 || p >= PI) continue;
                d.a[u][k] = pfma(gv, xr[p], d.a[u][k]);
            }
        }
    }
}
template <int NO>
__device__ __forceinline__ void wgrad2_bias(const f32x2 (&G)[NO][(NO + 1) / 2], DAcc& d)
{
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int j = 0; j < (NO + 1) / 2; ++j) d.bs = d.bs + G[o][j];       // pad column of an odd-width plane is kept zero
}

// out(row r of the NI-wide input gradient) += down^T(G): out(2j,2j+1) += G[o][j+1]*(w0,w1) + G[o][j]*(w2,w3) + (G[o][j-1]*w4, 0),
// o = (r+2-u)/2 for the u of r's parity.  Odd NI: the pad column collects garbage -- the caller clears it where it matters.
template <int NI, int NO>
__device__ __forceinline__ void downT_row(const f32x2 (&G)[NO][(NO + 1) / 2], int r, const Taps& td, f32x2 (&out)[(NI + 1) / 2])
{
    constexpr int PI = (NI + 1) / 2, PO = (NO + 1) / 2;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        if ((r + 2 - u) & 1) continue;
        const int o = (r + 2 - u) / 2;
        if (r + 2 - u < 0 || o >= NO) continue;
#pragma unroll
        for (int j = 0; j < PI; ++j) if (j + 1 < NO) out[j] = pfma(splat(at<PO>(G[o], j + 1)), td.p[u][0], out[j]);
#pragma unroll
        for (int j = 0; j < PI; ++j) if (j < NO) out[j] = pfma(splat(at<PO>(G[o], j)), td.p[u][1], out[j]);
#pragma unroll
        for (int j = 1; j < PI; ++j) if (j - 1 < NO) out[j].x = fmaf(at<PO>(G[o], j - 1), td.p[u][2].x, out[j].x);
    }
}

// gC (NI x NI) += R^T(row d of the NO x NO fine gradient): horizontal adjoint into NI scalars, then the vertical one
template <int MODE, int NI, int NO>
__device__ __forceinline__ void resizeT_row(const f32x2 (&row)[(NO + 1) / 2], int d, f32x2 (&gC)[NI][(NI + 1) / 2])
{
    constexpr int PO = (NO + 1) / 2, PI = (NI + 1) / 2;
    float h[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) h[i] = 0.f;
#pragma unroll
    for (int q = 0; q < NO; ++q) {
        const VT t = vtab(MODE, NI, NO, q);
        const float v = at<PO>(row, q);
        if (MODE == 1 || t.i0 == t.i1) h[t.i0] += v;
        else { h[t.i0] = fmaf(1.f - t.l, v, h[t.i0]); h[t.i1] = fmaf(t.l, v, h[t.i1]); }
    }
    const VT tv = vtab(MODE, NI, NO, d);
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const f32x2 hp = f32x2{h[2 * j], 2 * j + 1 < NI ? h[2 * j + 1] : 0.f};
        if (MODE == 1 || tv.i0 == tv.i1) gC[tv.i0][j] = gC[tv.i0][j] + hp;
        else {
            gC[tv.i0][j] = pfma(splat(1.f - tv.l), hp, gC[tv.i0][j]);
            gC[tv.i1][j] = pfma(splat(tv.l), hp, gC[tv.i1][j]);
        }
    }
}

// a float32 plane of the saved pyramid (N x NW x NW x C) for this lane's (image, channel), as pairs
template <int NW>
__device__ __forceinline__ void load_plane(const char* base, unsigned long long off, int n, int C, int c, f32x2 (&p)[NW][(NW + 1) / 2])
{
    const float* q = reinterpret_cast<const float*>(base + off) + ((size_t)n * NW * NW) * C + c;
#pragma unroll
    for (int o = 0; o < NW; ++o)
#pragma unroll
        for (int j = 0; j < (NW + 1) / 2; ++j)
            p[o][j] = f32x2{q[(size_t)(o * NW + 2 * j) * C], 2 * j + 1 < NW ? q[(size_t)(o * NW + 2 * j + 1) * C] : 0.f};
}

// one row of weight-gradient partial sums per image: part[(n * 26 + tap) * C + c], tap 25 = the bias gradient
template <class Acc>
__device__ __forceinline__ void store_wacc(float* part, int n, int C, int c, const Acc& a)
{
    float* q = part + (size_t)n * 26 * C + c;
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 5; ++v) q[(size_t)(u * 5 + v) * C] = a.tap(u, v);
    q[(size_t)25 * C] = a.bias();
}

}  // namespace cplbwd
}  // namespace rcx
