// Piece-level check of rcx_cplbwd.hip's register-plane adjoints against plain loops on the host (one lane = one random plane).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize tools/cplbwd_probe.hip -o tools/cplbwd_probe
#include "../recnext_amd/csrc/rcx_cplbwd.hip"
#include <stdio.h>
#include <vector>
#include <math.h>
namespace rcx { bool cpl7b_applicable(int, int, int, int, int, int, int) { return true; } }   // link stub (rcx_cpl14.hip is not part of this harness)
using namespace rcx;
using namespace rcx::cplbwd;

template <int N>
__device__ void ld(const float* p, f32x2 (&P)[N][(N + 1) / 2])
{
    for (int r = 0; r < N; ++r)
        for (int j = 0; j < (N + 1) / 2; ++j) P[r][j] = f32x2{p[r * N + 2 * j], 2 * j + 1 < N ? p[r * N + 2 * j + 1] : 0.f};
}

// out: [0..24] wgrad s1 (g: NxN, T: NxN), [25] bias; [26..50] wgrad s2 (x: NxN, G: MxM); [51..] downT(G) NxN; then resizeT(g) MxM
template <int N, int M>
__global__ void k_probe(const float* g, const float* T, const float* G, const float* w, float* out)
{
    f32x2 gp[N][(N + 1) / 2], Tp[N][(N + 1) / 2], Gp[M][(M + 1) / 2];
    ld<N>(g, gp); ld<N>(T, Tp); ld<M>(G, Gp);
    WAcc a; a.zero();
#pragma unroll
    for (int t = 0; t < N; ++t) wgrad_row<N>(gp[t], t, [&](int r) -> const f32x2(&)[(N + 1) / 2] { return Tp[r]; }, a);
    for (int u = 0; u < 5; ++u) for (int v = 0; v < 5; ++v) out[u * 5 + v] = a.tap(u, v);
    out[25] = a.bias();
    DAcc d; d.zero();
#pragma unroll
    for (int r = 0; r < N; ++r) wgrad2_row<N, M>(Tp[r], r, Gp, d);
    for (int u = 0; u < 5; ++u) for (int v = 0; v < 5; ++v) out[26 + u * 5 + v] = d.tap(u, v);
    Taps td;
    for (int u = 0; u < 5; ++u) { td.p[u][0] = f32x2{w[u * 5], w[u * 5 + 1]}; td.p[u][1] = f32x2{w[u * 5 + 2], w[u * 5 + 3]}; td.p[u][2] = f32x2{w[u * 5 + 4], 0.f}; }
    td.bias = 0.f;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        f32x2 o[(N + 1) / 2];
        for (int j = 0; j < (N + 1) / 2; ++j) o[j] = f32x2{0.f, 0.f};
        downT_row<N, M>(Gp, r, td, o);
        for (int q = 0; q < N; ++q) out[51 + r * N + q] = (q & 1) ? o[q >> 1].y : o[q >> 1].x;
    }
    f32x2 gC[M][(M + 1) / 2];
    for (int i = 0; i < M; ++i) for (int j = 0; j < (M + 1) / 2; ++j) gC[i][j] = f32x2{0.f, 0.f};
#pragma unroll
    for (int dd = 0; dd < N; ++dd) resizeT_row<0, M, N>(gp[dd], dd, gC);
    for (int i = 0; i < M; ++i) for (int q = 0; q < M; ++q) out[51 + N * N + i * M + q] = (q & 1) ? gC[i][q >> 1].y : gC[i][q >> 1].x;
}

template <int N, int M>
int run()
{
    std::vector<float> g(N * N), T(N * N), G(M * M), w(25), out(51 + N * N + M * M), ref(out.size(), 0.f);
    unsigned s = 12345u + N;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : g) v = rnd(); for (auto& v : T) v = rnd(); for (auto& v : G) v = rnd(); for (auto& v : w) v = rnd();
    auto Tz = [&](int r, int c) { return (r < 0 || r >= N || c < 0 || c >= N) ? 0.f : T[r * N + c]; };
    for (int u = 0; u < 5; ++u) for (int v = 0; v < 5; ++v) {
        double a = 0, b = 0;
        for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) a += (double)g[r * N + c] * Tz(r + u - 2, c + v - 2);
        for (int o = 0; o < M; ++o) for (int i = 0; i < M; ++i) b += (double)G[o * M + i] * Tz(2 * o + u - 2, 2 * i + v - 2);
        ref[u * 5 + v] = (float)a; ref[26 + u * 5 + v] = (float)b;
    }
    { double a = 0; for (auto v : g) a += v; ref[25] = (float)a; }
    for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) {
        double a = 0;
        for (int u = 0; u < 5; ++u) for (int v = 0; v < 5; ++v) {
            const int ro = r + 2 - u, co = c + 2 - v;
            if (ro < 0 || co < 0 || (ro & 1) || (co & 1) || ro / 2 >= M || co / 2 >= M) continue;
            a += (double)G[(ro / 2) * M + co / 2] * w[u * 5 + v];
        }
        ref[51 + r * N + c] = (float)a;
    }
    for (int dd = 0; dd < N; ++dd) for (int q = 0; q < N; ++q) {
        const lanes::VT tv = lanes::vtab(0, M, N, dd), th = lanes::vtab(0, M, N, q);
        const float v = g[dd * N + q];
        const float wv[2] = {1.f - tv.l, tv.l}, wh[2] = {1.f - th.l, th.l};
        const int iv[2] = {tv.i0, tv.i1}, ih[2] = {th.i0, th.i1};
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) ref[51 + N * N + iv[a] * M + ih[b]] += wv[a] * wh[b] * v;
    }
    float *dg, *dT, *dG, *dw, *dout;
    hipMalloc(&dg, 4 * g.size()); hipMalloc(&dT, 4 * T.size()); hipMalloc(&dG, 4 * G.size()); hipMalloc(&dw, 100); hipMalloc(&dout, 4 * out.size());
    hipMemcpy(dg, g.data(), 4 * g.size(), hipMemcpyHostToDevice); hipMemcpy(dT, T.data(), 4 * T.size(), hipMemcpyHostToDevice);
    hipMemcpy(dG, G.data(), 4 * G.size(), hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), 100, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_probe<N, M>), dim3(1), dim3(1), 0, 0, dg, dT, dG, dw, dout);
    hipMemcpy(out.data(), dout, 4 * out.size(), hipMemcpyDeviceToHost);
    const char* names[4] = {"wgrad_row", "wgrad2_row", "downT_row", "resizeT_row"};
    const int lo[5] = {0, 26, 51, 51 + N * N, (int)out.size()};
    int bad = 0;
    for (int k = 0; k < 4; ++k) {
        float e = 0; int at = -1;
        for (int i = lo[k]; i < lo[k + 1]; ++i) if (fabsf(out[i] - ref[i]) > e) { e = fabsf(out[i] - ref[i]); at = i - lo[k]; }
        printf("N=%d M=%d %-12s max err %.3g at %d\n", N, M, names[k], e, at);
        if (e > 1e-4f) {
            ++bad;
            for (int i = lo[k]; i < lo[k + 1] && i < lo[k] + 50; ++i) printf("  [%d] got %.5f want %.5f%s\n", i - lo[k], out[i], ref[i], fabsf(out[i] - ref[i]) > 1e-4f ? "  <--" : "");
        }
    }
    return bad;
}

int main() { int b = run<4, 2>() + run<7, 4>() + run<14, 7>(); printf(b ? "FAILED\n" : "ok\n"); return b != 0; }
