// Backward pass of RecConv2d for gfx950: the gradients a training
// step of engine.py:48-64 needs -- d/dx, d/dW_down (accumulated over all levels: the ladder shares one weight,
// model/recnext.py:21,28), d/dW_convs[j], and the biases.
//
// Forward (training) keeps F_1..F_L and C_1..C_L in fp32 (rcx_recconv2d_fwd_train); T_l = F_l + resize(C_{l+1}) is
// re-formed on the fly.  With g* the gradient of *:
//
//   gT_0 = K_L^T gy                         gW_L += <T_0, gy>
//   for l = 1..L:   gC_l = R^T gT_{l-1}     gT_l = K_j^T gC_l        gW_j += <T_l, gC_l>        (j = L-l)
//   gF_L = gT_L;  for l = L..1:  gW_d += <F_{l-1}, gF_l>_s2 ;  gF_{l-1} = gT_{l-1} + D^T gF_l   (F_0 = x, gF_0 = gx)
//
// K^T is the forward depthwise kernel with the taps flipped (the host passes a flipped pack); D^T (stride-2 adjoint)
// and R^T (resize adjoint) are gathers, so every result is deterministic; weight gradients are reduced in two stages
// (per-block partials, then a fixed-order sum).  These kernels favour clarity over speed: they are correct HIP, not tuned.
#include "rcx_common.h"
#include "rcx_launch.h"

namespace rcx {

constexpr int BW_V = 4;        // channels per thread (C % 4 == 0 is required by the training path)

struct BwGeom {
    int N, C;
    int H, W;          // fine extent
    int Hc, Wc;        // coarse extent
    int k;
    float sy, sx;      // Hc/H, Wc/W
    int mode;
};

// ---- D^T: out(fine) = base(fine) + sum_{u,v} W[u,v] * g(coarse) over the coarse pixels whose stride-2 window covers it ----
template <typename TO>
__global__ void __launch_bounds__(256)
k_down_bwd_input(const float* __restrict__ base, const float* __restrict__ g, TO* __restrict__ out, const float* __restrict__ w, BwGeom q)
{
    const int cvecs = q.C / BW_V, p = q.k / 2;
    const long long total = (long long)q.N * q.H * q.W * cvecs;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        long long r = t;
        const int c = (int)(r % cvecs) * BW_V; r /= cvecs;
        const int ix = (int)(r % q.W); r /= q.W;
        const int iy = (int)(r % q.H);
        const int n = (int)(r / q.H);
        float acc[BW_V];
        load_vec<BW_V>(base + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, acc);
        for (int u = 0; u < q.k; ++u) {
            const int ty = iy + p - u;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= q.Hc) continue;
            for (int v = 0; v < q.k; ++v) {
                const int tx = ix + p - v;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= q.Wc) continue;
                float gv[BW_V], wv[BW_V];
                load_vec<BW_V>(g + (((size_t)n * q.Hc + oy) * q.Wc + ox) * q.C + c, gv);
                load_vec<BW_V>(w + ((size_t)u * q.k + v) * q.C + c, wv);
#pragma unroll
                for (int i = 0; i < BW_V; ++i) acc[i] = fmaf(wv[i], gv[i], acc[i]);
            }
        }
        store_vec<BW_V>(out + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, acc);
    }
}

// ---- R^T: gc(coarse) = sum over the fine pixels that read this coarse pixel, with the forward's exact weights ----
__device__ __forceinline__ float axis_weight(int d, int target, int n_in, float scale, int mode)
{
    if (mode == 1) return nearest_src(d, n_in, scale) == target ? 1.f : 0.f;
    const Lerp l = bilinear_src(d, n_in, scale);
    float w = 0.f;
    if (l.i0 == target) w += 1.f - l.lam;
    if (l.i1 == target) w += l.lam;
    return w;
}

__global__ void __launch_bounds__(256)
k_resize_bwd(const float* __restrict__ gfine, float* __restrict__ gcoarse, BwGeom q)
{
    const int cvecs = q.C / BW_V;
    const long long total = (long long)q.N * q.Hc * q.Wc * cvecs;
    // fine indices that can reference coarse index t lie within ((t-1.5)/s - 1, (t+1.5)/s + 1)
    const float isy = (float)q.H / (float)q.Hc, isx = (float)q.W / (float)q.Wc;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        long long r = t;
        const int c = (int)(r % cvecs) * BW_V; r /= cvecs;
        const int cx = (int)(r % q.Wc); r /= q.Wc;
        const int cy = (int)(r % q.Hc);
        const int n = (int)(r / q.Hc);
        int y0 = (int)floorf(((float)cy - 1.5f) * isy) - 1, y1 = (int)ceilf(((float)cy + 1.5f) * isy) + 1;
        int x0 = (int)floorf(((float)cx - 1.5f) * isx) - 1, x1 = (int)ceilf(((float)cx + 1.5f) * isx) + 1;
        if (cy == q.Hc - 1) y1 = q.H - 1;                    // the last coarse row also serves every clamped fine row
        if (cx == q.Wc - 1) x1 = q.W - 1;
        y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
        y1 = y1 > q.H - 1 ? q.H - 1 : y1; x1 = x1 > q.W - 1 ? q.W - 1 : x1;
        float acc[BW_V] = {0.f, 0.f, 0.f, 0.f};
        for (int y = y0; y <= y1; ++y) {
            const float wy = axis_weight(y, cy, q.Hc, q.sy, q.mode);
            if (wy == 0.f) continue;
            for (int x = x0; x <= x1; ++x) {
                const float wx = axis_weight(x, cx, q.Wc, q.sx, q.mode);
                if (wx == 0.f) continue;
                float gv[BW_V];
                load_vec<BW_V>(gfine + (((size_t)n * q.H + y) * q.W + x) * q.C + c, gv);
                const float wgt = wy * wx;
#pragma unroll
                for (int i = 0; i < BW_V; ++i) acc[i] = fmaf(wgt, gv[i], acc[i]);
            }
        }
        store_vec<BW_V>(gcoarse + (((size_t)n * q.Hc + cy) * q.Wc + cx) * q.C + c, acc);
    }
}

// ---- weight / bias gradients, stage 1: per-row partial sums ----
// In(n, iy, ix) = A (+ resize(coarse)) with zero padding; gW[u,v,c] = sum_{n,oy,ox} In(n, S*oy+u-p, S*ox+v-p, c) * g(n,oy,ox,c)
// A row of `partial` is [k*k taps + 1 bias row][C]; block (bx, by) thread (tx, ty) writes row by*blockDim.y + ty.
template <typename TA, int S, bool HAS_COARSE>
__global__ void __launch_bounds__(256)
k_wgrad_partial(const TA* __restrict__ a, const float* __restrict__ coarse, const float* __restrict__ g,
                float* __restrict__ partial, BwGeom q, int Ho, int Wo)
{
    const int cvecs = q.C / BW_V;
    const int cv = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y * blockDim.y + threadIdx.y;
    const int rows = gridDim.y * blockDim.y;
    if (cv >= cvecs) return;
    const int c = cv * BW_V, p = q.k / 2, kk = q.k * q.k;
    float* prow = partial + (size_t)row * (kk + 1) * q.C + c;
    // taps are visited one at a time so that only one accumulator vector is live (k is a runtime value here)
    const long long npix = (long long)q.N * Ho * Wo;
    for (int tap = 0; tap <= kk; ++tap) {
        const int u = tap / q.k, v = tap % q.k;
        float acc[BW_V] = {0.f, 0.f, 0.f, 0.f};
        for (long long pi = row; pi < npix; pi += rows) {
            long long r = pi;
            const int ox = (int)(r % Wo); r /= Wo;
            const int oy = (int)(r % Ho);
            const int n = (int)(r / Ho);
            float gv[BW_V];
            load_vec<BW_V>(g + (((size_t)n * Ho + oy) * Wo + ox) * q.C + c, gv);
            if (tap == kk) {                                 // bias row: sum of g
#pragma unroll
                for (int i = 0; i < BW_V; ++i) acc[i] += gv[i];
                continue;
            }
            const int iy = oy * S + u - p, ix = ox * S + v - p;
            if (iy < 0 || iy >= q.H || ix < 0 || ix >= q.W) continue;
            float in[BW_V];
            load_vec<BW_V>(a + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, in);
            if constexpr (HAS_COARSE) {
                const float* cn = coarse + (size_t)n * q.Hc * q.Wc * q.C;
                if (q.mode == 1) {
                    const int cy = nearest_src(iy, q.Hc, q.sy), cx = nearest_src(ix, q.Wc, q.sx);
                    float t[BW_V];
                    load_vec<BW_V>(cn + ((size_t)cy * q.Wc + cx) * q.C + c, t);
#pragma unroll
                    for (int i = 0; i < BW_V; ++i) in[i] += t[i];
                } else {
                    const Lerp ly = bilinear_src(iy, q.Hc, q.sy), lx = bilinear_src(ix, q.Wc, q.sx);
                    float a00[BW_V], a01[BW_V], a10[BW_V], a11[BW_V];
                    load_vec<BW_V>(cn + ((size_t)ly.i0 * q.Wc + lx.i0) * q.C + c, a00);
                    load_vec<BW_V>(cn + ((size_t)ly.i0 * q.Wc + lx.i1) * q.C + c, a01);
                    load_vec<BW_V>(cn + ((size_t)ly.i1 * q.Wc + lx.i0) * q.C + c, a10);
                    load_vec<BW_V>(cn + ((size_t)ly.i1 * q.Wc + lx.i1) * q.C + c, a11);
                    const float wy1 = ly.lam, wy0 = 1.f - ly.lam, wx1 = lx.lam, wx0 = 1.f - lx.lam;
#pragma unroll
                    for (int i = 0; i < BW_V; ++i)
                        in[i] += wy0 * (wx0 * a00[i] + wx1 * a01[i]) + wy1 * (wx0 * a10[i] + wx1 * a11[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < BW_V; ++i) acc[i] = fmaf(in[i], gv[i], acc[i]);
        }
        store_vec<BW_V>(prow + (size_t)tap * q.C, acc);
    }
}

// stage 2: dst[i] (+)= sum over rows of partial[row][i], fixed order
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ gw, float* __restrict__ gb, int rows, int kk, int C, int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (kk + 1) * C) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += partial[(size_t)r * (kk + 1) * C + i];
    if (i < kk * C) gw[i] = accumulate ? gw[i] + s : s;
    else if (gb) gb[i - kk * C] = accumulate ? gb[i - kk * C] + s : s;
}

// ---------------- host side ----------------
static unsigned grid_for(long long total)
{
    long long b = (total + 255) / 256;
    if (b > 256LL * 32) b = 256LL * 32;
    return (unsigned)(b < 1 ? 1 : b);
}

constexpr int WG_ROWS_Y = 8;       // blockDim.y of k_wgrad_partial
constexpr int WG_BLOCKS_Y = 64;    // gridDim.y

size_t wgrad_partial_bytes(int C, int k) { return sizeof(float) * (size_t)WG_ROWS_Y * WG_BLOCKS_Y * (k * k + 1) * C; }

template <typename TA>
static hipError_t wgrad_launch(const void* a, const float* coarse, const float* g, float* partial, float* gw, float* gb,
                               BwGeom q, int Ho, int Wo, int stride, int accumulate, hipStream_t s)
{
    const int cvecs = q.C / BW_V;
    dim3 block(32, WG_ROWS_Y), grid((cvecs + 31) / 32, WG_BLOCKS_Y);
    if (stride == 2) hipLaunchKernelGGL((k_wgrad_partial<TA, 2, false>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo);
    else if (coarse) hipLaunchKernelGGL((k_wgrad_partial<TA, 1, true>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo);
    else hipLaunchKernelGGL((k_wgrad_partial<TA, 1, false>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int kk = q.k * q.k, n = (kk + 1) * q.C;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((n + 255) / 256), dim3(256), 0, s, partial, gw, gb, WG_ROWS_Y * WG_BLOCKS_Y, kk, q.C, accumulate);
    return hipGetLastError();
}

hipError_t bwd_wgrad(const void* a, int a_dt, const float* coarse, const float* g, float* partial, float* gw, float* gb,
                     int N, int C, int H, int W, int Hc, int Wc, int Ho, int Wo, int k, int stride, int mode, int accumulate, hipStream_t s)
{
    BwGeom q{};
    q.N = N; q.C = C; q.H = H; q.W = W; q.Hc = Hc; q.Wc = Wc; q.k = k; q.mode = mode;
    q.sy = Hc > 0 ? (float)Hc / (float)H : 0.f;
    q.sx = Wc > 0 ? (float)Wc / (float)W : 0.f;
    if (a_dt == 1) return wgrad_launch<bf16_t>(a, coarse, g, partial, gw, gb, q, Ho, Wo, stride, accumulate, s);
    return wgrad_launch<float>(a, coarse, g, partial, gw, gb, q, Ho, Wo, stride, accumulate, s);
}

hipError_t bwd_down_input(const float* base, const float* g, void* out, int out_dt, const float* w,
                          int N, int C, int H, int W, int Hc, int Wc, int k, hipStream_t s)
{
    BwGeom q{};
    q.N = N; q.C = C; q.H = H; q.W = W; q.Hc = Hc; q.Wc = Wc; q.k = k;
    const unsigned grid = grid_for((long long)N * H * W * (C / BW_V));
    if (out_dt == 1) hipLaunchKernelGGL(k_down_bwd_input<bf16_t>, dim3(grid), dim3(256), 0, s, base, g, (bf16_t*)out, w, q);
    else hipLaunchKernelGGL(k_down_bwd_input<float>, dim3(grid), dim3(256), 0, s, base, g, (float*)out, w, q);
    return hipGetLastError();
}

hipError_t bwd_resize(const float* gfine, float* gcoarse, int N, int C, int H, int W, int Hc, int Wc, int mode, hipStream_t s)
{
    BwGeom q{};
    q.N = N; q.C = C; q.H = H; q.W = W; q.Hc = Hc; q.Wc = Wc; q.mode = mode;
    q.sy = (float)Hc / (float)H; q.sx = (float)Wc / (float)W;
    hipLaunchKernelGGL(k_resize_bwd, dim3(grid_for((long long)N * Hc * Wc * (C / BW_V))), dim3(256), 0, s, gfine, gcoarse, q);
    return hipGetLastError();
}

}  // namespace rcx
