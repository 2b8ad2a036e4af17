// Channel-per-lane weight gradient of a stride-1 5x5 depthwise conv whose input is T = a + R(coarse) (model/recnext.py:33-34:
// the final conv and the level convs of the RecConv2d block) on the 56x56 and 28x28 planes -- the per-step backward's largest
// kernels (k_wgrad_rows: 182 / 88 us at batch 128 where the traffic is worth 23 / 12: a thread there walks one image row with ten
// dependent loads per pixel).  Here a wave owns 64 channels of a 14-column x 14-row tile of the gradient plane:
//   gW[u][v] += sum_{t,c} g[t][c] * T[t+u-2][c+v-2]
// with the tap-pair accumulation of rcx_cplbwd_pieces.h (acc(v,v+1) += splat(g) * T(c+v-2, c+v-1), T read only as aligned pairs).
// T rows are built once per tile row into a five-row ring of 18 columns (tile + 2 halo columns each side, zeros outside the
// image): a row of `a`, plus the bilinear (or nearest) resize of the coarse plane -- for the exact 2x step of these planes its
// weights are the constants 0.25 / 0.75 and its indices a fixed pattern of the tile (clamping the indices reproduces ATen's
// border cases: both taps land on the same element there).  Loads run a tile row ahead.  The W/14 waves of a workgroup (one per
// column tile) add their 26 sums per channel through LDS in a fixed order and leave ONE partial row per (image, 14-row band):
// deterministic, summed over the batch by the same second stage as everything else (k_wgrad_reduce_jobs).
#include "rcx_cplbwd_pieces.h"
#include "rcx_opts.h"

namespace rcx {
namespace cplwgrad {

using namespace cplbwd;

// MODE 0 / 1: T = a + bilinear / nearest resize of `coarse`; MODE 2: T = a (a plain depthwise conv, e.g. RecAttn2d's: model/recattn.py:163-171)
// TG: the gradient's element type (float32; or the block's own 16-bit type at level 0, where g is the incoming gy as autograd hands it over)
template <int MODE, typename TA, int H, typename TG = float>
__global__ __launch_bounds__(64 * (H / 14))
void k_wgrad_cpl(const TA* __restrict__ a, const float* __restrict__ coarse, const TG* __restrict__ g, float* __restrict__ partial,
                 int N, int C)
{
    constexpr int W = H, NT = W / 14, NB = H / 14, Hc = H / 2, Wc = W / 2, TP = 9;      // TP: pairs of a T row (18 columns)
    __shared__ float red[NT][26][64];
    const int tile = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int nb = (C + 63) / 64;
    const unsigned unit = blockIdx.x;
    const int cb = (int)(unit % (unsigned)nb), band = (int)((unit / (unsigned)nb) % (unsigned)NB), n = (int)(unit / (unsigned)(nb * NB));
    const int c = cb * 64 + lane;
    const bool live = c < C;
    const unsigned cl = (unsigned)(live ? c : C - 1);                      // ragged last block: the spare lanes shadow the last channel
    const int r0 = 14 * band, c0 = 14 * tile;
    const unsigned voa = cl * (unsigned)sizeof(TA), vof = cl * 4u, vog = cl * (unsigned)sizeof(TG);
    const size_t pixa = (size_t)C * sizeof(TA), pixf = (size_t)C * 4, pixg = (size_t)C * sizeof(TG);
    const gcptr ab = (gcptr)a + (size_t)n * H * W * pixa;
    const gcptr gb = (gcptr)g + (size_t)n * H * W * pixg;
    const gcptr cbp = (gcptr)coarse + (size_t)n * Hc * Wc * pixf;

    // ---- loads (ordered, compiler-counted: SafeLd) ----
    uint32_t ra[18][18], rg[14][14], rc[11][11];                           // as loaded; only the rows in flight are live
    auto row_ok = [&](int s) { const int r = r0 - 2 + s; return r >= 0 && r < H; };     // wave-uniform
    // T local row s = image row r0 - 2 + s, columns c0 - 2 .. c0 + 15.  No branches: rows and columns outside the image are clamped
    // here (a valid address) and zeroed when the T row is built -- conditional loads leave the compiler with conditionally defined
    // registers, and it then keeps all eighteen rows apart (511 registers).
    auto ld_a = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        int r = r0 - 2 + s;
        r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        const gcptr rowp = ab + (size_t)r * W * pixa;
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            int col = c0 + q - 2;
            if (q < 2) col = col < 0 ? 0 : col;
            if (q >= 16) col = col > W - 1 ? W - 1 : col;
            ra[s][q] = SafeLd<TA>::ld(rowp + (size_t)col * pixa + voa);
        }
    };
    auto ld_g = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const gcptr rowp = gb + ((size_t)(r0 + t) * W + c0) * pixg;
#pragma unroll
        for (int q = 0; q < 14; ++q) rg[t][q] = SafeLd<TG>::ld(rowp + (size_t)q * pixg + vog);
    };
    auto ld_c = [&](auto ic) {                                             // coarse local row il = coarse row 7 band - 2 + il (clamped), 11 columns
        constexpr int il = decltype(ic)::value;
        int i = 7 * band - 2 + il;
        i = i < 0 ? 0 : (i > Hc - 1 ? Hc - 1 : i);
        const gcptr rowp = cbp + (size_t)i * Wc * pixf;
#pragma unroll
        for (int jl = 0; jl < 11; ++jl) {
            int j = 7 * tile - 2 + jl;
            j = j < 0 ? 0 : (j > Wc - 1 ? Wc - 1 : j);
            rc[il][jl] = SafeLd<float>::ld(rowp + (size_t)j * pixf + vof);
        }
    };

    // ---- the coarse plane resized horizontally for this tile's 18 columns: Hr[il % 3] ----
    f32x2 Hr[3][TP];
    auto build_h = [&](auto ic) {
        constexpr int il = decltype(ic)::value;
        pin_raw(rc[il]);
        float cv[11];
#pragma unroll
        for (int jl = 0; jl < 11; ++jl) cv[jl] = __uint_as_float(rc[il][jl]);
#pragma unroll
        for (int m = 0; m < TP; ++m) {                                     // columns q = 2m (even image column), 2m + 1 (odd)
            if (MODE == 1) Hr[il % 3][m] = f32x2{cv[m + 1], cv[m + 1]};        // nearest: column >> 1
            else Hr[il % 3][m] = f32x2{fmaf(0.25f, cv[m], 0.75f * cv[m + 1]), fmaf(0.75f, cv[m + 1], 0.25f * cv[m + 2])};
        }
    };
    // ---- T rows: ring of five ----
    f32x2 T[5][TP];
    const f32x2 keep_lo = splat(c0 == 0 ? 0.f : 1.f), keep_hi = splat(c0 + 14 == W ? 0.f : 1.f);
    auto build_t = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        pin_raw(ra[s]);
        // vertical taps of the exact 2x step: even image row -> coarse rows (il, il + 1) = (s/2, s/2 + 1) with (0.25, 0.75),
        // odd -> ((s+1)/2, (s+1)/2 + 1) with (0.75, 0.25); nearest: the row >> 1 = local (s + 2) >> 1
        constexpr int i0 = (s & 1) ? (s + 1) / 2 : s / 2;
        constexpr float w0 = (s & 1) ? 0.75f : 0.25f, w1 = 1.f - w0;
        const f32x2 keep = splat(row_ok(s) ? 1.f : 0.f);                   // rows outside the image are zero padding (wave-uniform)
#pragma unroll
        for (int m = 0; m < TP; ++m) {
            f32x2 v = f32x2{SafeLd<TA>::cvt(ra[s][2 * m]), SafeLd<TA>::cvt(ra[s][2 * m + 1])};
            if (MODE == 1) v = v + Hr[((s + 2) >> 1) % 3][m];
            else if (MODE == 0) v = pfma(splat(w1), Hr[(i0 + 1) % 3][m], pfma(splat(w0), Hr[i0 % 3][m], v));
            T[s % 5][m] = v * keep;
        }
        // and so are the halo columns outside it (a multiply, not a branch: a uniform branch here doubles the code and the registers)
        T[s % 5][0] = T[s % 5][0] * keep_lo;
        T[s % 5][TP - 1] = T[s % 5][TP - 1] * keep_hi;
    };
    // which coarse rows T row s needs: its first use builds the H row
    auto need_h = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (MODE == 2) return;
        constexpr int hi = MODE == 1 ? (s + 2) >> 1 : ((s & 1) ? (s + 1) / 2 : s / 2) + 1;       // the highest local coarse row T row s reads
        constexpr int prev = s == 0 ? -1 : (MODE == 1 ? (s + 1) >> 1 : (((s - 1) & 1) ? s / 2 : (s - 1) / 2) + 1);
        if constexpr (s == 0) {
            if constexpr (MODE == 1) build_h(IC<1>{});
            else { build_h(IC<0>{}); build_h(IC<1>{}); }
        } else if constexpr (hi > prev) build_h(IC<hi>{});
    };

    WAcc acc;
    acc.zero();
    // prologue: everything the first gradient row needs
    sfor<5>([&](auto sc) { ld_a(sc); });
    if constexpr (MODE != 2) sfor<4>([&](auto ic) { ld_c(ic); });
    ld_g(IC<0>{});
    sfor<5>([&](auto sc) { need_h(sc); build_t(sc); });
    RCX_FENCE;
    sfor<14>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // a tile row ahead: T row t + 5, the coarse row it may need, gradient row t + 1
        if constexpr (t + 5 < 18) {
            ld_a(IC<t + 5>{});
            constexpr int s = t + 5;
            constexpr int hi = MODE == 1 ? (s + 2) >> 1 : ((s & 1) ? (s + 1) / 2 : s / 2) + 1;
            constexpr int prev = MODE == 1 ? (s + 1) >> 1 : (((s - 1) & 1) ? s / 2 : (s - 1) / 2) + 1;
            if constexpr (MODE != 2 && hi > prev) ld_c(IC<hi>{});
        }
        if constexpr (t + 1 < 14) ld_g(IC<t + 1>{});
        pin_raw(rg[t]);
        f32x2 gp[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) gp[j] = f32x2{SafeLd<TG>::cvt(rg[t][2 * j]), SafeLd<TG>::cvt(rg[t][2 * j + 1])};
        // gradient column cc (local) meets T local columns cc .. cc + 4, i.e. pairs cc/2 + k (even) / (cc - 1)/2 + k (odd, set O)
#pragma unroll
        for (int cc = 0; cc < 14; ++cc) {
            const f32x2 gv = splat(at<7>(gp, cc));
#pragma unroll
            for (int u = 0; u < 5; ++u) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (cc & 1) acc.O[u][k] = pfma(gv, T[(t + u) % 5][(cc - 1) / 2 + k], acc.O[u][k]);
                    else acc.E[u][k] = pfma(gv, T[(t + u) % 5][cc / 2 + k], acc.E[u][k]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) acc.bs = acc.bs + gp[j];
#pragma unroll
        for (int u = 0; u < 5; ++u) { pin(acc.E[u]); pin(acc.O[u]); }
        RCX_FENCE;
        if constexpr (t + 5 < 18) { need_h(IC<t + 5>{}); build_t(IC<t + 5>{}); }
        RCX_FENCE;
    });

    // ---- the workgroup's column tiles, fixed order ----
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 5; ++v) red[tile][u * 5 + v][lane] = acc.tap(u, v);
    red[tile][25][lane] = acc.bias();
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

template <int MODE, typename TA, int H, typename TG>
static hipError_t launch(const void* a, const float* coarse, const void* g, float* partial, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * (H / 14) * ((C + 63) / 64));
    hipLaunchKernelGGL((k_wgrad_cpl<MODE, TA, H, TG>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, coarse, (const TG*)g, partial, N, C);
    return hipGetLastError();
}

// g_same: the gradient has a's (16-bit) element type instead of float32
template <int MODE, typename TA>
static hipError_t launch_h(const void* a, const float* coarse, const void* g, bool g_same, float* partial, int N, int C, int H, hipStream_t s)
{
    if constexpr (sizeof(TA) == 2) {
        if (g_same) return H == 56 ? launch<MODE, TA, 56, TA>(a, coarse, g, partial, N, C, s) : launch<MODE, TA, 28, TA>(a, coarse, g, partial, N, C, s);
    }
    return H == 56 ? launch<MODE, TA, 56, float>(a, coarse, g, partial, N, C, s) : launch<MODE, TA, 28, float>(a, coarse, g, partial, N, C, s);
}

// ---- stride-2 convs: gW[u][v] += sum_{o,i} G[o][i] * a[2o+u-P][2i+v-P] ----
// K = 5, MULT = 1: the block's shared down conv (model/recnext.py:21, :28); K = 7, MULT = 2: the Downsample conv between stages
// (nn.Conv2d(C, 2C, 7, stride 2, groups=C): a lane is an OUTPUT channel and reads input channel c / 2).
// A wave owns 64 channels of a 14 x 14 tile of G = the gradient of the conv's output, i.e. (26 + K) rows x (26 + K) columns of its
// input a (P halo pixels on the low side, P - 1 on the high side, zeros outside the image).  Input-row stationary: an `a` row meets
// the (K + 1) / 2 G rows whose window covers it, tap pairs (v, v+1) against a's aligned pairs (rcx_cplbwd_pieces.h, wgrad2_row).
template <int K>
struct DAccK {
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

template <typename TA, int H, int K, int MULT>
__global__ __launch_bounds__(64 * (H / 28))
void k_wgrad2_cpl(const TA* __restrict__ a, const float* __restrict__ g, float* __restrict__ partial, int N, int C)
{
    constexpr int W = H, Ho = H / 2, Wo = W / 2, NT = Wo / 14, NB = Ho / 14, P = K / 2, KP = (K + 1) / 2, KK1 = K * K + 1;
    constexpr int NS = 26 + K, AP = (NS + 1) / 2, RG = (K + 1) / 2;          // `a` rows / pairs per row of a tile, G ring
    __shared__ float red[NT][KK1][64];
    const int tile = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int nb = (C + 63) / 64;                                          // C: channels of g and of the weight
    const unsigned unit = blockIdx.x;
    const int cb = (int)(unit % (unsigned)nb), band = (int)((unit / (unsigned)nb) % (unsigned)NB), n = (int)(unit / (unsigned)(nb * NB));
    const int c = cb * 64 + lane;
    const bool live = c < C;
    const unsigned cl = (unsigned)(live ? c : C - 1);
    const int o0 = 14 * band, i0 = 14 * tile;
    const int Ca = C / MULT;
    const unsigned voa = (cl / (unsigned)MULT) * (unsigned)sizeof(TA), vof = cl * 4u;
    const size_t pixa = (size_t)Ca * sizeof(TA), pixf = (size_t)C * 4;
    const gcptr ab = (gcptr)a + (size_t)n * H * W * pixa;
    const gcptr gb = (gcptr)g + (size_t)n * Ho * Wo * pixf;

    uint32_t ra[NS][2 * AP], rg[14][14];
    auto ld_a = [&](auto sc) {                                             // local row s = image row 2 o0 - P + s, local column q = image column 2 i0 - P + q (clamped)
        constexpr int s = decltype(sc)::value;
        int r = 2 * o0 - P + s;
        r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        const gcptr rowp = ab + (size_t)r * W * pixa;
#pragma unroll
        for (int q = 0; q < 2 * AP; ++q) {
            int col = 2 * i0 - P + q;
            if (q < P) col = col < 0 ? 0 : col;
            if (q >= 28 + P) col = col > W - 1 ? W - 1 : col;
            ra[s][q] = SafeLd<TA>::ld(rowp + (size_t)col * pixa + voa);
        }
    };
    auto ld_g = [&](auto oc) {
        constexpr int o = decltype(oc)::value;
        const gcptr rowp = gb + ((size_t)(o0 + o) * Wo + i0) * pixf;
#pragma unroll
        for (int q = 0; q < 14; ++q) rg[o][q] = SafeLd<float>::ld(rowp + (size_t)q * pixf + vof);
    };
    f32x2 G[RG][7];                                                        // ring: G row o in slot o % RG
    // columns outside the image (zero padding): masks of the two lowest and the two highest pairs (a multiply, not a branch)
    f32x2 mlo[2], mhi[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        mlo[m] = f32x2{2 * i0 - P + 2 * m >= 0 ? 1.f : 0.f, 2 * i0 - P + 2 * m + 1 >= 0 ? 1.f : 0.f};
        mhi[m] = f32x2{2 * i0 - P + 2 * (AP - 2 + m) < W ? 1.f : 0.f, 2 * i0 - P + 2 * (AP - 2 + m) + 1 < W ? 1.f : 0.f};
    }
    DAccK<K> acc;
    acc.zero();
    auto take_g = [&](auto oc) {
        constexpr int o = decltype(oc)::value;
        pin_raw(rg[o]);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            G[o % RG][j] = f32x2{__uint_as_float(rg[o][2 * j]), __uint_as_float(rg[o][2 * j + 1])};
            acc.bs = acc.bs + G[o % RG][j];
        }
    };
    ld_g(IC<0>{});
    ld_a(IC<0>{});
    ld_a(IC<1>{});
    take_g(IC<0>{});
    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + 2 < NS) ld_a(IC<s + 2>{});
        // G row o' is first met by `a` row s = 2 o': fetched two rows earlier, taken one row earlier
        if constexpr ((s & 1) == 0 && s / 2 + 1 < 14) ld_g(IC<s / 2 + 1>{});
        pin_raw(ra[s]);
        f32x2 ar[AP];
#pragma unroll
        for (int m = 0; m < AP; ++m) ar[m] = f32x2{SafeLd<TA>::cvt(ra[s][2 * m]), SafeLd<TA>::cvt(ra[s][2 * m + 1])};
        if constexpr (s < P || s >= 28 + P) {                              // rows that can fall outside the image: zero padding
            const int r = 2 * o0 - P + s;
            const f32x2 keep = splat(r >= 0 && r < H ? 1.f : 0.f);
#pragma unroll
            for (int m = 0; m < AP; ++m) ar[m] = ar[m] * keep;
        }
        ar[0] = ar[0] * mlo[0]; ar[1] = ar[1] * mlo[1];
        ar[AP - 2] = ar[AP - 2] * mhi[0]; ar[AP - 1] = ar[AP - 1] * mhi[1];
#pragma unroll
        for (int il = 0; il < 14; ++il) {
#pragma unroll
            for (int o = 0; o < 14; ++o) {
                const int u = s - 2 * o;
                if (u < 0 || u > K - 1) continue;
                const f32x2 gv = splat(at<7>(G[o % RG], il));
#pragma unroll
                for (int k = 0; k < KP; ++k) acc.a[u][k] = pfma(gv, ar[il + k], acc.a[u][k]);
            }
        }
#pragma unroll
        for (int u = 0; u < K; ++u) pin(acc.a[u]);
        RCX_FENCE;
        if constexpr ((s & 1) == 1 && (s + 1) / 2 < 14) take_g(IC<(s + 1) / 2>{});       // needed from row s + 1 on; its slot retired with row s
        RCX_FENCE;
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

template <typename TA, int H, int K, int MULT>
static hipError_t launch2(const void* a, const float* g, float* partial, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * (H / 28) * ((C + 63) / 64));
    hipLaunchKernelGGL((k_wgrad2_cpl<TA, H, K, MULT>), dim3(grid), dim3(64 * (H / 28)), 0, s, (const TA*)a, g, partial, N, C);
    return hipGetLastError();
}
template <typename TA, int K, int MULT>
static hipError_t launch2_h(const void* a, const float* g, float* partial, int N, int C, int H, hipStream_t s)
{
    return H == 56 ? launch2<TA, 56, K, MULT>(a, g, partial, N, C, s) : launch2<TA, 28, K, MULT>(a, g, partial, N, C, s);
}

}  // namespace cplwgrad

// rows of `partial` the kernel leaves: one per (image, 14-row band)
bool wgrad_cpl_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int stride, bool has_coarse)
{
    const char* v = rcx::opt::value(rcx::opt::WGRAD_CPL);
    if (v && *v == '0') return false;
    return k == 5 && stride == 1 && H == W && (H == 56 || H == 28) && (!has_coarse || (Hc * 2 == H && Wc * 2 == W)) && C >= 1 &&
           N * (H / 14) <= 512;
}

bool wgrad2_cpl_applicable(int N, int C, int H, int W, int Ho, int Wo, int k, int stride, bool has_coarse)
{
    const char* v = rcx::opt::value(rcx::opt::WGRAD_CPL);
    if (v && *v == '0') return false;
    return !has_coarse && k == 5 && stride == 2 && H == W && (H == 56 || H == 28) && Ho * 2 == H && Wo * 2 == W && C >= 1 &&
           N * (H / 28) <= 512;
}

hipError_t wgrad2_cpl(const void* a, int a_dt, const float* g, float* partial, int N, int C, int H, hipStream_t s, int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 28);
    return a_dt == 1 ? cplwgrad::launch2_h<bf16_t, 5, 1>(a, g, partial, N, C, H, s) : a_dt == 2 ? cplwgrad::launch2_h<f16_t, 5, 1>(a, g, partial, N, C, H, s)
                                                                                                  : cplwgrad::launch2_h<float, 5, 1>(a, g, partial, N, C, H, s);
}

// the Downsample conv (7x7, stride 2, channel multiplier 2): C = OUTPUT channels; rows of `partial`: N * (H / 28), each (49 + 1) * C
bool wgrad2m_cpl_applicable(int N, int Cout, int H, int W, int k)
{
    const char* v = rcx::opt::value(rcx::opt::WGRAD_CPL);
    if (v && *v == '0') return false;
    return k == 7 && H == W && (H == 56 || H == 28) && Cout >= 2 && Cout % 2 == 0 && N * (H / 28) <= 512;
}

hipError_t wgrad2m_cpl(const void* a, int a_dt, const float* g, float* partial, int N, int Cout, int H, hipStream_t s, int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 28);
    return a_dt == 1 ? cplwgrad::launch2_h<bf16_t, 7, 2>(a, g, partial, N, Cout, H, s) : a_dt == 2 ? cplwgrad::launch2_h<f16_t, 7, 2>(a, g, partial, N, Cout, H, s)
                                                                                                     : cplwgrad::launch2_h<float, 7, 2>(a, g, partial, N, Cout, H, s);
}

hipError_t wgrad_cpl(const void* a, int a_dt, const float* coarse, const void* g, int g_dt, float* partial, int N, int C, int H, int mode,
                     hipStream_t s, int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 14);
    if (g_dt != 0 && g_dt != a_dt) return hipErrorInvalidValue;             // float32, or the same 16-bit type as a
    const bool gs = g_dt != 0;
#define RCX_WC(MD_) (a_dt == 1 ? cplwgrad::launch_h<MD_, bf16_t>(a, coarse, g, gs, partial, N, C, H, s) : a_dt == 2 ? cplwgrad::launch_h<MD_, f16_t>(a, coarse, g, gs, partial, N, C, H, s) : cplwgrad::launch_h<MD_, float>(a, coarse, g, gs, partial, N, C, H, s))
    if (!coarse) return RCX_WC(2);
    return mode == 1 ? RCX_WC(1) : RCX_WC(0);
#undef RCX_WC
}

}  // namespace rcx
