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
// (per-block partials, then a fixed-order sum).  The k = 5 weight-gradient kernel (k_wgrad_rows) is the tuned one; the
// rest favour clarity over speed.
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
        float acc[BW_V] = {0.f, 0.f, 0.f, 0.f};
        if (base) load_vec<BW_V>(base + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, acc);
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

// The same two adjoints with the kernel size known at compile time (K = 5: the RecConv2d ladder, K = 7: Downsample).  A fine pixel
// (iy, ix) only meets the taps u = (iy + P) mod 2 + 2 uu, v = (ix + P) mod 2 + 2 vv: ceil(K/2)^2 candidates instead of K^2 loop
// trips with a parity test each (the generic kernels above: 381 us for the first Downsample of RecNeXt-M3 at batch 128, whose
// traffic is worth 25).  MULT = 1: out = base + D^T g on BW_V channels per thread; MULT = 2: two input = four output channels.
template <typename TO, int K, int MULT>
__global__ void __launch_bounds__(256)
k_down_bwd_input_k(const float* __restrict__ base, const float* __restrict__ g, TO* __restrict__ out, const float* __restrict__ w, BwGeom q)
{
    constexpr int P = K / 2, NT = (K + 1) / 2, CV = MULT == 2 ? 2 : BW_V;       // input channels per thread
    const int cvecs = q.C / CV, Cg = MULT * q.C;                                 // g and w have Cg channels
    const long long total = (long long)q.N * q.H * q.W * cvecs;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        long long r = t;
        const int c = (int)(r % cvecs) * CV; r /= cvecs;
        const int ix = (int)(r % q.W); r /= q.W;
        const int iy = (int)(r % q.H);
        const int n = (int)(r / q.H);
        float acc[CV];
#pragma unroll
        for (int i = 0; i < CV; ++i) acc[i] = 0.f;
        if (MULT == 1 && base) load_vec<CV>(base + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, acc);
        const int pu = (iy + P) & 1, pv = (ix + P) & 1;
        const int oy0 = (iy + P - pu) >> 1, ox0 = (ix + P - pv) >> 1;            // the output pixel tap (pu, pv) pairs with
        const float* gimg = g + (size_t)n * q.Hc * q.Wc * Cg + MULT * c;
        const float* wc = w + MULT * c;
#pragma unroll
        for (int uu = 0; uu < NT; ++uu) {
            const int u = pu + 2 * uu, oy = oy0 - uu;
            if (u >= K || oy < 0 || oy >= q.Hc) continue;
#pragma unroll
            for (int vv = 0; vv < NT; ++vv) {
                const int v = pv + 2 * vv, ox = ox0 - vv;
                if (v >= K || ox < 0 || ox >= q.Wc) continue;
                float gv[4], wv[4];
                load_vec<4>(gimg + ((size_t)oy * q.Wc + ox) * Cg, gv);
                load_vec<4>(wc + (size_t)(u * K + v) * Cg, wv);
                if constexpr (MULT == 2) {
                    acc[0] = fmaf(wv[0], gv[0], fmaf(wv[1], gv[1], acc[0]));
                    acc[1] = fmaf(wv[2], gv[2], fmaf(wv[3], gv[3], acc[1]));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(wv[i], gv[i], acc[i]);
                }
            }
        }
        store_vec<CV>(out + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, acc);
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
        // the column weights once per coarse pixel, not once per candidate row (the 2x step has nine candidates per axis of which four
        // are non-zero: 81 weight evaluations became 18); wider windows (scale > 2, the clamped last row) keep the direct form
        constexpr int MAXW = 12;
        if (x1 - x0 < MAXW) {
            float wxs[MAXW];
#pragma unroll
            for (int j = 0; j < MAXW; ++j) wxs[j] = x0 + j <= x1 ? axis_weight(x0 + j, cx, q.Wc, q.sx, q.mode) : 0.f;
            for (int y = y0; y <= y1; ++y) {
                const float wy = axis_weight(y, cy, q.Hc, q.sy, q.mode);
                if (wy == 0.f) continue;
                const float* grow = gfine + (((size_t)n * q.H + y) * q.W + x0) * q.C + c;
#pragma unroll
                for (int j = 0; j < MAXW; ++j) {
                    if (wxs[j] == 0.f) continue;
                    float gv[BW_V];
                    load_vec<BW_V>(grow + (size_t)j * q.C, gv);
                    const float wgt = wy * wxs[j];
#pragma unroll
                    for (int i = 0; i < BW_V; ++i) acc[i] = fmaf(wgt, gv[i], acc[i]);
                }
            }
        } else {
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
        }
        store_vec<BW_V>(gcoarse + (((size_t)n * q.Hc + cy) * q.Wc + cx) * q.C + c, acc);
    }
}

// ---- D^T for nn.Conv2d(C, 2C, k, stride 2, groups=C): gx(n,iy,ix,c) = sum_m sum_{u,v} W[2c+m,u,v] * g(n,oy,ox,2c+m) ----
template <typename TO>
__global__ void __launch_bounds__(256)
k_down_bwd_input_mult2(const float* __restrict__ g, TO* __restrict__ out, const float* __restrict__ w, BwGeom q)
{
    // q.C = input channels, q.Hc/q.Wc = output extent; g and w have 2*q.C channels
    const int p = q.k / 2, Co = 2 * q.C;
    const long long total = (long long)q.N * q.H * q.W * (q.C / 2);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        long long r = t;
        const int c = (int)(r % (q.C / 2)) * 2; r /= (q.C / 2);            // two input channels = four output channels
        const int ix = (int)(r % q.W); r /= q.W;
        const int iy = (int)(r % q.H);
        const int n = (int)(r / q.H);
        float acc[2] = {0.f, 0.f};
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
                float gv[4], wv[4];
                load_vec<4>(g + (((size_t)n * q.Hc + oy) * q.Wc + ox) * Co + 2 * c, gv);
                load_vec<4>(w + ((size_t)u * q.k + v) * Co + 2 * c, wv);
                acc[0] = fmaf(wv[0], gv[0], fmaf(wv[1], gv[1], acc[0]));
                acc[1] = fmaf(wv[2], gv[2], fmaf(wv[3], gv[3], acc[1]));
            }
        }
        store_vec<2>(out + (((size_t)n * q.H + iy) * q.W + ix) * q.C + c, acc);
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


// ---- weight / bias gradients for k = 5, the fast path ----
// One thread = (channel pair, row slot); all 25 tap sums (+ the bias sum) of its channel pair live in registers while it
// walks whole image rows, so the output gradient is read five times from L1 and the input once per row:
//   stride 1:  rows are INPUT rows (n, iy): T = a (+ resize(coarse)) is formed once per pixel and multiplied against a
//              5-wide sliding window of each of the five output-gradient rows iy-2 .. iy+2;
//   stride 2:  rows are OUTPUT rows (n, oy): for each of the five input rows a 5-wide window slides two pixels per output.
// A block reduces its 8 row slots through LDS (fixed order) and writes one partial row; k_wgrad_reduce sums the partial
// rows in a fixed order: bit-reproducible.
constexpr int WR_LANES = 32, WR_SLOTS = 8;

__device__ __forceinline__ void ld2(const float* p, float (&o)[2]) { float2 t = *reinterpret_cast<const float2*>(p); o[0] = t.x; o[1] = t.y; }
__device__ __forceinline__ void ld2(const bf16_t* p, float (&o)[2])
{
    uint32_t t = *reinterpret_cast<const uint32_t*>(p);
    o[0] = __uint_as_float(t << 16); o[1] = __uint_as_float(t & 0xffff0000u);
}

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ float ld1(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void ld2(const f16_t* p, float (&o)[2]) { load_vec<2>(p, o); }

// K = 3, 5, 7.  MULT = 2: nn.Conv2d(C, 2C, groups=C) -- the thread's two OUTPUT channels (2cp, 2cp+1) share input channel cp
// (q.C is the number of output channels, `a` has q.C / MULT).
template <typename TA, int S, bool HAS_COARSE, int K, int MULT>
__global__ void __launch_bounds__(WR_LANES * WR_SLOTS)
k_wgrad_rows(const TA* __restrict__ a, const float* __restrict__ coarse, const float* __restrict__ g,
             float* __restrict__ partial, BwGeom q, int Ho, int Wo)
{
    constexpr int P = K / 2, KK = K * K, AV = MULT == 2 ? 1 : 2, RP = 13, NP = (KK + 1 + RP - 1) / RP;
    static_assert(!(HAS_COARSE && MULT == 2), "the coarse operand only exists for the plain depthwise conv");
    __shared__ float red[WR_SLOTS][RP][WR_LANES * 2];
    const int cp = blockIdx.x * WR_LANES + threadIdx.x;
    const bool ok = cp < q.C / 2;
    const int c = (ok ? cp : 0) * 2;                 // first of the thread's two output channels
    const int ca = MULT == 2 ? c / 2 : c;            // its (first) input channel
    const int Ca = q.C / MULT;                       // channels of `a`
    const int slot = blockIdx.y * WR_SLOTS + threadIdx.y, nslots = gridDim.y * WR_SLOTS;
    float acc[KK][2], accb[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t][0] = acc[t][1] = 0.f;
    auto lda = [&](const TA* p, float (&o)[AV]) __attribute__((always_inline)) {
        if constexpr (AV == 2) ld2(p, o); else o[0] = ld1(p);
    };

    if (ok) {
        if constexpr (S == 1) {
            const int rows = q.N * q.H;
            for (int row = slot; row < rows; row += nslots) {
                const int n = row / q.H, iy = row - n * q.H;
                const TA* arow = a + ((size_t)n * q.H + iy) * q.W * Ca + ca;
                const float* gimg = g + (size_t)n * Ho * Wo * q.C + c;
                const float* cimg = HAS_COARSE ? coarse + (size_t)n * q.Hc * q.Wc * q.C + c : nullptr;
                Lerp ly{0, 0, 0.f};
                int ny = 0;
                if constexpr (HAS_COARSE) {
                    if (q.mode == 1) ny = nearest_src(iy, q.Hc, q.sy);
                    else ly = bilinear_src(iy, q.Hc, q.sy);
                }
                // gw[u][d] = g(oy_u, ix - P + d), oy_u = iy - u + P; rows outside the image contribute nothing
                float gw[K][K][2];
                const float* grow[K];
                bool gok[K];
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const int oy = iy - u + P;
                    gok[u] = oy >= 0 && oy < Ho;
                    grow[u] = gimg + (size_t)(gok[u] ? oy : 0) * Wo * q.C;
#pragma unroll
                    for (int d = 0; d < K; ++d) {
                        const int ox = d - P;
                        gw[u][d][0] = gw[u][d][1] = 0.f;
                        if (gok[u] && ox >= 0 && ox < Wo) ld2(grow[u] + (size_t)ox * q.C, gw[u][d]);
                    }
                }
                // the window is a ring: K pixels per trip of the loop, so that its slots keep their registers (a shifting window costs
                // 2 K (K - 1) moves per pixel, as many as a third of the FMAs); logical slot d sits in physical slot (d + jj) % K
                for (int ix0 = 0; ix0 < q.W; ix0 += K) {
#pragma unroll
                  for (int jj = 0; jj < K; ++jj) {
                    const int ix = ix0 + jj;
                    if (ix >= q.W) break;
                    float tv[AV], t[2];
                    lda(arow + (size_t)ix * Ca, tv);
                    t[0] = tv[0]; t[1] = tv[AV - 1];
                    if constexpr (HAS_COARSE) {
                        if (q.mode == 1) {
                            float cv[2];
                            ld2(cimg + ((size_t)ny * q.Wc + nearest_src(ix, q.Wc, q.sx)) * q.C, cv);
                            t[0] += cv[0]; t[1] += cv[1];
                        } else {
                            const Lerp lx = bilinear_src(ix, q.Wc, q.sx);
                            float a00[2], a01[2], a10[2], a11[2];
                            ld2(cimg + ((size_t)ly.i0 * q.Wc + lx.i0) * q.C, a00);
                            ld2(cimg + ((size_t)ly.i0 * q.Wc + lx.i1) * q.C, a01);
                            ld2(cimg + ((size_t)ly.i1 * q.Wc + lx.i0) * q.C, a10);
                            ld2(cimg + ((size_t)ly.i1 * q.Wc + lx.i1) * q.C, a11);
                            const float wy1 = ly.lam, wy0 = 1.f - ly.lam, wx1 = lx.lam, wx0 = 1.f - lx.lam;
#pragma unroll
                            for (int i = 0; i < 2; ++i) t[i] += wy0 * (wx0 * a00[i] + wx1 * a01[i]) + wy1 * (wx0 * a10[i] + wx1 * a11[i]);
                        }
                    }
                    // tap (u, v) pairs input column ix with output column ix - v + P, i.e. window slot d = K - 1 - v
#pragma unroll
                    for (int u = 0; u < K; ++u)
#pragma unroll
                        for (int v = 0; v < K; ++v) {
                            acc[u * K + v][0] = fmaf(t[0], gw[u][(K - 1 - v + jj) % K][0], acc[u * K + v][0]);
                            acc[u * K + v][1] = fmaf(t[1], gw[u][(K - 1 - v + jj) % K][1], acc[u * K + v][1]);
                        }
                    accb[0] += gw[P][(P + jj) % K][0]; accb[1] += gw[P][(P + jj) % K][1];       // g(iy, ix): every output pixel exactly once
                    // the slot that held logical 0 takes the next pixel's logical K - 1
#pragma unroll
                    for (int u = 0; u < K; ++u) {
                        const int ox = ix + 1 + P;
                        gw[u][jj][0] = gw[u][jj][1] = 0.f;
                        if (gok[u] && ox < Wo) ld2(grow[u] + (size_t)ox * q.C, gw[u][jj]);
                    }
                  }
                }
            }
        } else {
            const int rows = q.N * Ho;
            for (int row = slot; row < rows; row += nslots) {
                const int n = row / Ho, oy = row - n * Ho;
                const float* grow = g + ((size_t)n * Ho + oy) * Wo * q.C + c;
                const TA* aimg = a + (size_t)n * q.H * q.W * Ca + ca;
                // aw[u][d] = a(2*oy + u - P, 2*ox - P + d)
                float aw[K][K][AV];
                const TA* arow[K];
                bool aok[K];
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const int iy = 2 * oy + u - P;
                    aok[u] = iy >= 0 && iy < q.H;
                    arow[u] = aimg + (size_t)(aok[u] ? iy : 0) * q.W * Ca;
#pragma unroll
                    for (int d = 0; d < K; ++d) {
                        const int ix = d - P;
#pragma unroll
                        for (int i = 0; i < AV; ++i) aw[u][d][i] = 0.f;
                        if (aok[u] && ix >= 0 && ix < q.W) lda(arow[u] + (size_t)ix * Ca, aw[u][d]);
                    }
                }
                // ring window again: two slots retire per output pixel, K odd, so K pixels bring every slot back to its register;
                // logical slot d sits in physical slot (d + 2 jj) % K
                for (int ox0 = 0; ox0 < Wo; ox0 += K) {
#pragma unroll
                  for (int jj = 0; jj < K; ++jj) {
                    const int ox = ox0 + jj;
                    if (ox >= Wo) break;
                    float gv[2];
                    ld2(grow + (size_t)ox * q.C, gv);
                    accb[0] += gv[0]; accb[1] += gv[1];
#pragma unroll
                    for (int u = 0; u < K; ++u)
#pragma unroll
                        for (int v = 0; v < K; ++v) {
                            acc[u * K + v][0] = fmaf(aw[u][(v + 2 * jj) % K][0], gv[0], acc[u * K + v][0]);
                            acc[u * K + v][1] = fmaf(aw[u][(v + 2 * jj) % K][AV - 1], gv[1], acc[u * K + v][1]);
                        }
#pragma unroll
                    for (int u = 0; u < K; ++u) {
#pragma unroll
                        for (int d = K - 2; d < K; ++d) {
                            const int ix = 2 * (ox + 1) - P + d;
                            const int ph = (d + 2 * (jj + 1)) % K;          // where logical d of the NEXT pixel lives: a slot that just retired
#pragma unroll
                            for (int i = 0; i < AV; ++i) aw[u][ph][i] = 0.f;
                            if (aok[u] && ix < q.W) lda(arow[u] + (size_t)ix * Ca, aw[u][ph]);
                        }
                    }
                  }
                }
            }
        }
    }
    // block reduction over the row slots, RP tap rows at a time (KK taps + the bias row), fixed order
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int t = pass * RP + i;
            red[threadIdx.y][i][threadIdx.x * 2 + 0] = t < KK ? acc[t < KK ? t : 0][0] : accb[0];
            red[threadIdx.y][i][threadIdx.x * 2 + 1] = t < KK ? acc[t < KK ? t : 0][1] : accb[1];
        }
        __syncthreads();
        for (int e = threadIdx.y * WR_LANES + threadIdx.x; e < RP * WR_LANES * 2; e += WR_LANES * WR_SLOTS) {
            const int i = e / (WR_LANES * 2), l = e - i * (WR_LANES * 2);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < WR_SLOTS; ++r) sum += red[r][i][l];
            const int ch = blockIdx.x * WR_LANES * 2 + l, t = pass * RP + i;
            if (ch < q.C && t <= KK) partial[((size_t)blockIdx.y * (KK + 1) + t) * q.C + ch] = sum;
        }
    }
}

// stage 2, parallel form: 8 threads share one output (rows r = j mod 8 each), fixed-order combine through LDS
__global__ void __launch_bounds__(256)
k_wgrad_reduce8(const float* __restrict__ partial, float* __restrict__ gw, float* __restrict__ gb, int rows, int kk, int C, int accumulate)
{
    __shared__ float red[8][32];
    const int i = blockIdx.x * 32 + threadIdx.x, j = threadIdx.y, total = (kk + 1) * C;
    float s = 0.f;
    if (i < total) {
        int r = j;
        for (; r + 24 < rows; r += 32) {                       // four rows in flight, fixed order (see k_wgrad_reduce_jobs)
            const float a0 = partial[(size_t)r * total + i], a1 = partial[(size_t)(r + 8) * total + i];
            const float a2 = partial[(size_t)(r + 16) * total + i], a3 = partial[(size_t)(r + 24) * total + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; r < rows; r += 8) s += partial[(size_t)r * total + i];
    }
    red[j][threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
        if (i < kk * C) gw[i] = accumulate ? gw[i] + t : t;
        else if (gb) gb[i - kk * C] = accumulate ? gb[i - kk * C] + t : t;
    }
}

// stage 2 for a whole block's backward in ONE launch (blockIdx.y = job): every conv of the block has its own partial buffer(s); the
// shared down conv sums the buffers of all its levels (coarsest first).  Fixed order throughout: deterministic.
constexpr int RJ = 32;             // threads that share one output (rows r = j mod RJ each): the tiled weight-gradient kernels leave up to 512 rows per buffer
__global__ void __launch_bounds__(32 * RJ)
k_wgrad_reduce_jobs(WgradJobs J)
{
    __shared__ float red[RJ][32];
    const int job = blockIdx.y;
    const int i = blockIdx.x * 32 + threadIdx.x, j = threadIdx.y, total = (J.kk + 1) * J.C;
    float s = 0.f;
    if (i < total)
        for (int sl = 0; sl < J.nslots[job]; ++sl) {
            const float* part = J.part[job][sl];
            const int rows = J.rows[job][sl];
            int r = j;
            // four rows in flight at a time (the loads are independent, the sum keeps its fixed order): a row per iteration is one
            // exposed memory latency per row
            for (; r + 3 * RJ < rows; r += 4 * RJ) {
                const float a0 = part[(size_t)r * total + i], a1 = part[(size_t)(r + RJ) * total + i];
                const float a2 = part[(size_t)(r + 2 * RJ) * total + i], a3 = part[(size_t)(r + 3 * RJ) * total + i];
                s += a0; s += a1; s += a2; s += a3;
            }
            for (; r < rows; r += RJ) s += part[(size_t)r * total + i];
        }
    red[j][threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < RJ; ++r) t += red[r][threadIdx.x];
        if (!J.param_layout) {
            if (i < J.kk * J.C) J.gw[job][i] = t;
            else if (J.gb[job]) J.gb[job][i - J.kk * J.C] = t;
        } else {
            // straight into the parameter's gradient: i = tap * C + c  ->  (C, 1, k, k) element c * kk + tap, in the parameter's own type
            void* dst = i < J.kk * J.C ? (void*)J.gw[job] : (void*)J.gb[job];
            const int e = i < J.kk * J.C ? (i % J.C) * J.kk + i / J.C : i - J.kk * J.C;
            if (dst) {
                if (J.param_dt == 0) reinterpret_cast<float*>(dst)[e] = t;
                else if (J.param_dt == 1) reinterpret_cast<bf16_t*>(dst)[e] = f32_to_bf16(t);
                else reinterpret_cast<f16_t*>(dst)[e] = (f16_t)t;
            }
        }
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

// partial rows of the k = 5 path: about 1024 blocks in total whatever C is
static int wr_grid_y(int C)
{
    const int gx = (C / 2 + WR_LANES - 1) / WR_LANES;
    int gy = 1024 / gx;
    return gy < 32 ? 32 : gy;
}

size_t wgrad_partial_bytes(int C, int k)
{
    const size_t old = sizeof(float) * (size_t)WG_ROWS_Y * WG_BLOCKS_Y * (k * k + 1) * C;
    const size_t fast = sizeof(float) * (size_t)wr_grid_y(C) * (k * k + 1) * C;
    return old > fast ? old : fast;
}

template <typename TA>
static hipError_t wgrad_launch(const void* a, const float* coarse, const float* g, float* partial, float* gw, float* gb,
                               BwGeom q, int Ho, int Wo, int stride, int accumulate, hipStream_t s, int* rows_out)
{
    if ((q.k == 3 || q.k == 5 || q.k == 7) && (q.C % 2) == 0 && !(coarse && q.k != 5)) {
        const int rows_total = stride == 2 ? q.N * Ho : q.N * q.H;
        int gy = wr_grid_y(q.C);
        const int need = (rows_total + WR_SLOTS - 1) / WR_SLOTS;
        if (gy > need) gy = need < 1 ? 1 : need;
        dim3 block(WR_LANES, WR_SLOTS), grid((q.C / 2 + WR_LANES - 1) / WR_LANES, gy);
#define RCX_WG(S_, HC_, K_) hipLaunchKernelGGL((k_wgrad_rows<TA, S_, HC_, K_, 1>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo)
        if (q.k == 5) { if (stride == 2) RCX_WG(2, false, 5); else if (coarse) RCX_WG(1, true, 5); else RCX_WG(1, false, 5); }
        else if (q.k == 3) { if (stride == 2) RCX_WG(2, false, 3); else RCX_WG(1, false, 3); }
        else { if (stride == 2) RCX_WG(2, false, 7); else RCX_WG(1, false, 7); }
#undef RCX_WG
        hipError_t e5 = hipGetLastError();
        if (e5 != hipSuccess) return e5;
        if (rows_out) { *rows_out = gy; return hipSuccess; }          // the caller reduces all its partial buffers in one launch
        const int kk = q.k * q.k, n5 = (kk + 1) * q.C;
        hipLaunchKernelGGL(k_wgrad_reduce8, dim3((n5 + 31) / 32), dim3(32, 8), 0, s, partial, gw, gb, gy, kk, q.C, accumulate);
        return hipGetLastError();
    }
    const int cvecs = q.C / BW_V;
    dim3 block(32, WG_ROWS_Y), grid((cvecs + 31) / 32, WG_BLOCKS_Y);
    if (stride == 2) hipLaunchKernelGGL((k_wgrad_partial<TA, 2, false>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo);
    else if (coarse) hipLaunchKernelGGL((k_wgrad_partial<TA, 1, true>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo);
    else hipLaunchKernelGGL((k_wgrad_partial<TA, 1, false>), grid, block, 0, s, (const TA*)a, coarse, g, partial, q, Ho, Wo);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (rows_out) { *rows_out = WG_ROWS_Y * WG_BLOCKS_Y; return hipSuccess; }
    const int kk = q.k * q.k, n = (kk + 1) * q.C;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((n + 255) / 256), dim3(256), 0, s, partial, gw, gb, WG_ROWS_Y * WG_BLOCKS_Y, kk, q.C, accumulate);
    return hipGetLastError();
}

hipError_t bwd_wgrad(const void* a, int a_dt, const float* coarse, const float* g, float* partial, float* gw, float* gb,
                     int N, int C, int H, int W, int Hc, int Wc, int Ho, int Wo, int k, int stride, int mode, int accumulate, hipStream_t s,
                     int* rows_out, int g_dt)
{
    // the 56x56 / 28x28 convs over T = a + R(coarse): tiled channel-per-lane kernel (rcx_cplwgrad.hip), when the caller reduces itself
    {
        const bool t1 = Ho == H && Wo == W && wgrad_cpl_applicable(N, C, H, W, Hc, Wc, k, stride, coarse != nullptr);
        const bool t2 = !t1 && wgrad2_cpl_applicable(N, C, H, W, Ho, Wo, k, stride, coarse != nullptr);
        if (g_dt != 0 && !t1) return hipErrorInvalidValue;
        if (t1 || t2) {
            int rows = 0;
            hipError_t e = t1 ? wgrad_cpl(a, a_dt, coarse, g, g_dt, partial, N, C, H, mode, s, &rows) : wgrad2_cpl(a, a_dt, g, partial, N, C, H, s, &rows);
            if (e != hipSuccess) return e;
            if (rows_out) { *rows_out = rows; return hipSuccess; }        // the caller reduces all its partial buffers in one launch
            const int kk = k * k, n5 = (kk + 1) * C;
            hipLaunchKernelGGL(k_wgrad_reduce8, dim3((n5 + 31) / 32), dim3(32, 8), 0, s, partial, gw, gb, rows, kk, C, accumulate);
            return hipGetLastError();
        }
    }
    BwGeom q{};
    q.N = N; q.C = C; q.H = H; q.W = W; q.Hc = Hc; q.Wc = Wc; q.k = k; q.mode = mode;
    q.sy = Hc > 0 ? (float)Hc / (float)H : 0.f;
    q.sx = Wc > 0 ? (float)Wc / (float)W : 0.f;
    if (a_dt == 1) return wgrad_launch<bf16_t>(a, coarse, g, partial, gw, gb, q, Ho, Wo, stride, accumulate, s, rows_out);
    if (a_dt == 2) return wgrad_launch<f16_t>(a, coarse, g, partial, gw, gb, q, Ho, Wo, stride, accumulate, s, rows_out);
    return wgrad_launch<float>(a, coarse, g, partial, gw, gb, q, Ho, Wo, stride, accumulate, s, rows_out);
}

hipError_t bwd_wgrad_reduce_jobs(const WgradJobs& J, hipStream_t s)
{
    const int n5 = (J.kk + 1) * J.C;
    hipLaunchKernelGGL(k_wgrad_reduce_jobs, dim3((n5 + 31) / 32, J.njobs), dim3(32, RJ), 0, s, J);
    return hipGetLastError();
}

// Downsample conv (channel multiplier 2, stride 2): input gradient and weight/bias gradients; Cin % 2 == 0
hipError_t bwd_mult2(const void* x, int x_dt, const float* g, const float* w, void* gx, float* partial, float* gw, float* gb,
                     int N, int Cin, int H, int W, int k, hipStream_t s)
{
    if (k != 3 && k != 5 && k != 7) return hipErrorInvalidConfiguration;
    const int p = k / 2, Ho = (H + 2 * p - k) / 2 + 1, Wo = (W + 2 * p - k) / 2 + 1;
    BwGeom q{};
    q.N = N; q.C = Cin; q.H = H; q.W = W; q.Hc = Ho; q.Wc = Wo; q.k = k;
    if (gx) {
        const unsigned grid = grid_for((long long)N * H * W * (Cin / 2));
        if (bwd_down7m2_cpt_applicable(N, Cin, H, W, k)) {          // RecNeXt's three Downsample convs at 224 x 224: tile kernel (rcx_cptbwd_kernels.h)
            hipError_t et = bwd_down7m2_cpt(g, gx, x_dt, w, N, Cin, H, s);
            if (et != hipSuccess) return et;
        } else
        if (k == 7) {
            if (x_dt == 1) hipLaunchKernelGGL((k_down_bwd_input_k<bf16_t, 7, 2>), dim3(grid), dim3(256), 0, s, (const float*)nullptr, g, (bf16_t*)gx, w, q);
            else if (x_dt == 2) hipLaunchKernelGGL((k_down_bwd_input_k<f16_t, 7, 2>), dim3(grid), dim3(256), 0, s, (const float*)nullptr, g, (f16_t*)gx, w, q);
            else hipLaunchKernelGGL((k_down_bwd_input_k<float, 7, 2>), dim3(grid), dim3(256), 0, s, (const float*)nullptr, g, (float*)gx, w, q);
        } else
        if (x_dt == 1) hipLaunchKernelGGL(k_down_bwd_input_mult2<bf16_t>, dim3(grid), dim3(256), 0, s, g, (bf16_t*)gx, w, q);
        else if (x_dt == 2) hipLaunchKernelGGL(k_down_bwd_input_mult2<f16_t>, dim3(grid), dim3(256), 0, s, g, (f16_t*)gx, w, q);
        else hipLaunchKernelGGL(k_down_bwd_input_mult2<float>, dim3(grid), dim3(256), 0, s, g, (float*)gx, w, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    q.C = 2 * Cin;                                     // the weight-gradient kernel counts output channels
    if (bwd_wgrad_dm_cpt_applicable(N, q.C, H, W, k)) {  // RecNeXt's three Downsample convs at 224 x 224: tile kernel (rcx_cptbwd_kernels.h)
        int rows = 0;
        hipError_t e3 = bwd_wgrad_dm_cpt(x, x_dt, g, partial, N, q.C, H, s, &rows);
        if (e3 != hipSuccess) return e3;
        const int kk3 = k * k, n53 = (kk3 + 1) * q.C;
        hipLaunchKernelGGL(k_wgrad_reduce8, dim3((n53 + 31) / 32), dim3(32, 8), 0, s, partial, gw, gb, rows, kk3, q.C, 0);
        return hipGetLastError();
    }
    if (wgrad2m_cpl_applicable(N, q.C, H, W, k)) {      // the first two Downsample convs of RecNeXt at 224x224: tiled kernel (rcx_cplwgrad.hip)
        int rows = 0;
        hipError_t e2 = wgrad2m_cpl(x, x_dt, g, partial, N, q.C, H, s, &rows);
        if (e2 != hipSuccess) return e2;
        const int kk2 = k * k, n52 = (kk2 + 1) * q.C;
        hipLaunchKernelGGL(k_wgrad_reduce8, dim3((n52 + 31) / 32), dim3(32, 8), 0, s, partial, gw, gb, rows, kk2, q.C, 0);
        return hipGetLastError();
    }
    int gy = wr_grid_y(q.C);
    const int need = (N * Ho + WR_SLOTS - 1) / WR_SLOTS;
    if (gy > need) gy = need < 1 ? 1 : need;
    dim3 block(WR_LANES, WR_SLOTS), grid((q.C / 2 + WR_LANES - 1) / WR_LANES, gy);
#define RCX_WGM(TA_, K_) hipLaunchKernelGGL((k_wgrad_rows<TA_, 2, false, K_, 2>), grid, block, 0, s, (const TA_*)x, (const float*)nullptr, g, partial, q, Ho, Wo)
    if (x_dt == 1) { if (k == 3) RCX_WGM(bf16_t, 3); else if (k == 5) RCX_WGM(bf16_t, 5); else RCX_WGM(bf16_t, 7); }
    else if (x_dt == 2) { if (k == 3) RCX_WGM(f16_t, 3); else if (k == 5) RCX_WGM(f16_t, 5); else RCX_WGM(f16_t, 7); }
    else { if (k == 3) RCX_WGM(float, 3); else if (k == 5) RCX_WGM(float, 5); else RCX_WGM(float, 7); }
#undef RCX_WGM
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int kk = k * k, n5 = (kk + 1) * q.C;
    hipLaunchKernelGGL(k_wgrad_reduce8, dim3((n5 + 31) / 32), dim3(32, 8), 0, s, partial, gw, gb, gy, kk, q.C, 0);
    return hipGetLastError();
}

hipError_t bwd_down_input(const float* base, const float* g, void* out, int out_dt, const float* w,
                          int N, int C, int H, int W, int Hc, int Wc, int k, hipStream_t s)
{
    BwGeom q{};
    q.N = N; q.C = C; q.H = H; q.W = W; q.Hc = Hc; q.Wc = Wc; q.k = k;
    const unsigned grid = grid_for((long long)N * H * W * (C / BW_V));
    if (k == 5) {
        if (out_dt == 1) hipLaunchKernelGGL((k_down_bwd_input_k<bf16_t, 5, 1>), dim3(grid), dim3(256), 0, s, base, g, (bf16_t*)out, w, q);
        else if (out_dt == 2) hipLaunchKernelGGL((k_down_bwd_input_k<f16_t, 5, 1>), dim3(grid), dim3(256), 0, s, base, g, (f16_t*)out, w, q);
        else hipLaunchKernelGGL((k_down_bwd_input_k<float, 5, 1>), dim3(grid), dim3(256), 0, s, base, g, (float*)out, w, q);
        return hipGetLastError();
    }
    if (out_dt == 1) hipLaunchKernelGGL(k_down_bwd_input<bf16_t>, dim3(grid), dim3(256), 0, s, base, g, (bf16_t*)out, w, q);
    else if (out_dt == 2) hipLaunchKernelGGL(k_down_bwd_input<f16_t>, dim3(grid), dim3(256), 0, s, base, g, (f16_t*)out, w, q);
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
