// Channel-per-lane BACKWARD of the RecConv2d block (the adjoint of model/recnext.py:24-34) for the two blocks
// of RecNeXt at 224x224 whose planes fit one lane: 14x14 / level 2 (13 of M3's 21 blocks) and 7x7 / level 1.  One launch per
// block instead of ~30: a lane owns one (image, channel) plane -- x, the saved pyramid, the incoming gradient and every
// intermediate gradient live in its registers (rcx_cpl14_pieces.h: float32 pairs of horizontally adjacent pixels), nothing is
// exchanged between lanes, and the only HBM traffic is x, gy, the small saved planes, gx, and one row of weight-gradient partial
// sums per image (reduced over the batch by k_wgrad_reduce_jobs, the same fixed-order second stage as the per-step schedule).
//
//   forward (level 1 shown; level 2 nests it once more):  F = down(X); C = conv_a(F); T = X + R(C); Y = conv_b(T)
//   backward:  gT = conv_b^T(gY)            gW_b = <gY, T>       gb_b = sum gY
//              gC = R^T(gT)                 gW_a = <gC, F>       gb_a = sum gC
//              gF = conv_a^T(gC)            gW_d += <gF, X>_s2   gb_d += sum gF
//              gX = gT + down^T(gF)
// Every convolution is again v_pk_fma_f32 on operands that sit where the instruction wants them:
//   * conv^T: the forward's input-row-stationary stride-1 conv with the flipped taps (conv5_row);
//   * weight gradients pair TAPS: acc(v, v+1) += splat(g[r][c]) * T(c+v-2, c+v-1).  Even c reads T's aligned pairs for the tap
//     pairs (0,1)(2,3)(4,-), odd c for (-,0)(1,2)(3,4): two accumulator sets per conv, added at the end -- no shifted copies;
//   * down^T pairs output columns: gx(2j,2j+1) += g[j+1]*(w0,w1) + g[j]*(w2,w3) + (g[j-1]*w4, 0) -- the forward's tap pairs;
//   * the resize adjoint is scalar (a few hundred instructions per plane).
// Same arithmetic as the per-step backward (float32 throughout, one rounding at the gx store); the summation ORDER differs, so
// results agree with it to float32 rounding, not bit for bit -- both are checked against autograd (tests/test_backward_gpu.py).
#include "rcx_cplbwd_pieces.h"
#include "rcx_opts.h"

namespace rcx {
namespace cplbwd {

struct BwdArgs {
    const void* x;                 // N x W x W x C, the block's input (TIO)
    const void* gy;                // N x W x W x C of TGY: float32, or (round 6) the block's own bf16 -- dL/dy as autograd hands it over under autocast
    const float* wpack;            // (level+2, 25, C): the down conv's taps are read from here
    const float* wflip;            // the same pack with every 5x5 flipped: conv^T = conv with these
    const char* saved;             // the training forward's pyramid
    unsigned long long f_off[2], c_off[2];
    void* gx;
    float* part[4];                // partial rows: job 0 = down, 1 + j = convs[j]
    int N, C;
    int split;                     // 14x14 kernel: two waves per plane (see k_recconv_bwd_cpl14)
};

// The level-1 block's backward on resident planes.  X: in = the block input, out = its gradient.  GY: gradient of the block
// output.  The saved down(X) and conv_a(F) are read from the pyramid.  conv_a = pack 1, conv_b = pack 2; their weight gradients
// leave as soon as they are complete (part_a / part_b), the shared down conv's accumulate in wd.
// EARLY: every tap set is requested up front (a lone wave sees each load's full latency otherwise: five of them in a row);
// the 14x14 kernel has no registers to spare for that while gT0 is parked (it spills to scratch) and loads at the point of use.
template <int MODE, int CT, int NW, int NC, bool EARLY>
__device__ __forceinline__ void level1_bwd(f32x2 (&X)[NW][(NW + 1) / 2], const f32x2 (&GY)[NW][(NW + 1) / 2], const BwdArgs& A,
                                           unsigned long long f_off, unsigned long long c_off, int n, int C, int c, unsigned vow,
                                           float* part_a, float* part_b, DAcc& wd)
{
    constexpr int PW = (NW + 1) / 2, PC = (NC + 1) / 2;
    f32x2 Cc[NC][PC], F[NC][PC];
    load_plane<NC>(A.saved, c_off, n, C, c, Cc);
    Taps tfb, tfa, td;
    if constexpr (EARLY) {
        load_plane<NC>(A.saved, f_off, n, C, c, F);
        load_taps<CT>(tfb, A.wflip, nullptr, 2, C, vow, 0);
        load_taps<CT>(tfa, A.wflip, nullptr, 1, C, vow, 0);
        load_taps<CT>(td, A.wpack, nullptr, 0, C, vow, 0);
    }
    {
        // T = X + R(Cc) and conv_b's weight gradient
        f32x2 T[NW][PW];
        {
            f32x2 H[NC][PW];
#pragma unroll
            for (int i = 0; i < NC; ++i) resize_row<MODE, NC, NW>(Cc[i], H[i]);
#pragma unroll
            for (int r = 0; r < NW; ++r) {
#pragma unroll
                for (int j = 0; j < PW; ++j) T[r][j] = X[r][j];
                add_resized_row<MODE, NC, NW, PW>(T[r], H, r);
            }
            pin(T);
            RCX_FENCE;
        }
        WAcc wb;
        wb.zero();
#pragma unroll
        for (int t = 0; t < NW; ++t) {
            wgrad_row<NW>(GY[t], t, [&](int r) -> const f32x2(&)[PW] { return T[r]; }, wb);
            RCX_FENCE;
        }
        store_wacc(part_b, n, C, c, wb);
        RCX_FENCE;
    }
    f32x2 gT[NW][PW];
    if constexpr (!EARLY) load_taps<CT>(tfb, A.wflip, nullptr, 2, C, vow, 0);
    conv5_plane<NW>(GY, gT, tfb);                           // gT = conv_b^T(gY)
    f32x2 gF[NC][PC];
    {
        f32x2 gC[NC][PC];
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int j = 0; j < PC; ++j) gC[i][j] = f32x2{0.f, 0.f};
#pragma unroll
        for (int d = 0; d < NW; ++d) resizeT_row<MODE, NC, NW>(gT[d], d, gC);
        pin(gC);
        RCX_FENCE;
        if constexpr (!EARLY) load_plane<NC>(A.saved, f_off, n, C, c, F);
        {
            WAcc wa;
            wa.zero();
#pragma unroll
            for (int t = 0; t < NC; ++t) wgrad_row<NC>(gC[t], t, [&](int r) -> const f32x2(&)[PC] { return F[r]; }, wa);
            store_wacc(part_a, n, C, c, wa);
            RCX_FENCE;
        }
        if constexpr (!EARLY) load_taps<CT>(tfa, A.wflip, nullptr, 1, C, vow, 0);
        conv5_plane<NC>(gC, gF, tfa);                       // gF = conv_a^T(gC)
    }
    // the shared down conv: weight gradient against X, and gX = gT + down^T(gF)
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        wgrad2_row<NW, NC>(X[r], r, gF, wd);
        RCX_FENCE;
    }
    wgrad2_bias<NC>(gF, wd);
    if constexpr (!EARLY) load_taps<CT>(td, A.wpack, nullptr, 0, C, vow, 0);
#pragma unroll
    for (int r = 0; r < NW; ++r) {
#pragma unroll
        for (int j = 0; j < PW; ++j) X[r][j] = gT[r][j];
        downT_row<NW, NC>(gF, r, td, X[r]);
        if (NW & 1) X[r][PW - 1].y = 0.f;
        pin(X[r]);
        RCX_FENCE;
    }
}

// ---- 7x7 / level 1 ----
template <int MODE, int CT, typename TIO, typename TGY = float>
__global__ __launch_bounds__(64)
void k_recconv_bwd_cpl7(BwdArgs A)
{
    constexpr int W = 7, P = 4, W1 = 4;
    const int C = CT > 0 ? CT : A.C;
    const int nb = (C + 63) / 64;
    unsigned b = blockIdx.x;
    const unsigned G = gridDim.x;
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);                 // XCD-aware order (rcx_cpl14.hip)
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= A.N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;
    const size_t pix = (size_t)C * sizeof(TIO);
    const gcptr xb = (gcptr)A.x + (size_t)n * W * W * pix;
    const gcptr gxb = (gcptr)A.gx + (size_t)n * W * W * pix;
    const gcptr gyb = (gcptr)A.gy + (size_t)n * W * W * (size_t)C * sizeof(TGY);
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);
    const RowAddr<W, CT, TGY> rg((unsigned)c * (unsigned)sizeof(TGY), (size_t)C * sizeof(TGY));

    uint32_t raw[W][W], rawg[W][W];
    sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            raw[r][decltype(qc)::value] = SafeLd<TIO>::ld(base + decltype(immc)::value + voff);
        });
    });
    sfor<W>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        rg.row(gyb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            rawg[r][decltype(qc)::value] = SafeLd<TGY>::ld(base + decltype(immc)::value + voff);
        });
    });
    RCX_FENCE;
    f32x2 X[W][P], GY[W][P];
#pragma unroll
    for (int r = 0; r < W; ++r) {
        pin_raw(raw[r]);
#pragma unroll
        for (int j = 0; j < P; ++j) X[r][j] = f32x2{SafeLd<TIO>::cvt(raw[r][2 * j]), 2 * j + 1 < W ? SafeLd<TIO>::cvt(raw[r][2 * j + 1]) : 0.f};
    }
#pragma unroll
    for (int r = 0; r < W; ++r) {
        pin_raw(rawg[r]);
#pragma unroll
        for (int j = 0; j < P; ++j) GY[r][j] = f32x2{SafeLd<TGY>::cvt(rawg[r][2 * j]), 2 * j + 1 < W ? SafeLd<TGY>::cvt(rawg[r][2 * j + 1]) : 0.f};
    }
    RCX_FENCE;
    DAcc wd;
    wd.zero();
    level1_bwd<MODE, CT, W, W1, true>(X, GY, A, A.f_off[0], A.c_off[0], n, C, c, vow, A.part[1], A.part[2], wd);
    // gx
    sfor<W>([&](auto rc) {
        constexpr int o = decltype(rc)::value;
        typename PixSt<TIO>::packed pk[P];
#pragma unroll
        for (int j = 0; j < P; ++j) pk[j] = PixSt<TIO>::prep(X[o][j]);
        ra.row(gxb, o, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            constexpr int q = decltype(qc)::value;
            PixSt<TIO>::st(base + decltype(immc)::value + voff, pk[q >> 1], q & 1);
        });
    });
    store_wacc(A.part[0], n, C, c, wd);
}

// ---- 14x14 / level 2 ----
// Four sweeps.  (1) final conv's weight gradient: x rows and gy rows stream in, T0 = x + R(C1) is built one row at a time into a ring
// of the five rows a gradient row meets.  (2) gT0 = conv_2^T(gy): gy streams a second time (L2), input-row stationary as the forward's
// pass 2; a finished row is parked in the accumulator half of the register file and folded into gC1 = R^T(gT0).  (3) the level-1
// block on the resident 7x7 planes (F1, gC1 -> gF1).  (4) x streams once more: the down conv's weight gradient against gF1, and
// gx = gT0 + down^T(gF1) leaves row by row.
// A.split: sweep (1) is independent of the rest, and a batch of 128 x 256 channels is 512 planes on 1024 SIMDs -- two waves per plane
// then (workgroups b and b + 8, the same XCD: both read the plane's gy), one doing sweep (1), the other sweeps (2)-(4).
template <int MODE, int CT, typename TIO, typename TGY = float>
__global__ __launch_bounds__(64)
void k_recconv_bwd_cpl14(BwdArgs A)
{
    constexpr int W = 14, P = 7, W1 = 7, P1 = 4, W2 = 4;
    const int C = CT > 0 ? CT : A.C;
    const int nb = (C + 63) / 64;
    unsigned b = blockIdx.x;
    unsigned G = gridDim.x;
    int role = 0;                                                          // 0: everything, 1: sweep (1), 2: sweeps (2)-(4)
    if (A.split) {                                                         // uniform: a kernel argument
        const unsigned xcd = b & 7u, slot = b >> 3;
        role = 1 + (int)(slot & 1u);
        b = (slot >> 1) * 8u + xcd;
        G >>= 1;
    }
    if ((G & 7u) == 0) b = (b & 7u) * (G >> 3) + (b >> 3);                 // XCD-aware order (rcx_cpl14.hip)
    const int n = (int)(b / (unsigned)nb), cb = (int)(b - (unsigned)n * (unsigned)nb);
    if (n >= A.N) return;
    const int c = cb * 64 + (int)threadIdx.x;
    if (c >= C) return;
    const size_t pix = (size_t)C * sizeof(TIO);
    const gcptr xb = (gcptr)A.x + (size_t)n * W * W * pix;
    const gcptr gxb = (gcptr)A.gx + (size_t)n * W * W * pix;
    const gcptr gyb = (gcptr)A.gy + (size_t)n * W * W * (size_t)C * sizeof(TGY);
    const unsigned vo = (unsigned)c * (unsigned)sizeof(TIO), vow = (unsigned)c * 4u;
    const RowAddr<W, CT, TIO> ra(vo, pix);
    const RowAddr<W, CT, TGY> rg((unsigned)c * (unsigned)sizeof(TGY), (size_t)C * sizeof(TGY));
    uint32_t rx[W][W], rgy[W][W];                           // rows as loaded; only the rows in flight are live
    auto ld_x = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        ra.row(xb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            rx[r][decltype(qc)::value] = SafeLd<TIO>::ld(base + decltype(immc)::value + voff);
        });
    };
    auto ld_g = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        rg.row(gyb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
            rgy[r][decltype(qc)::value] = SafeLd<TGY>::ld(base + decltype(immc)::value + voff);
        });
    };

    // ---- (1) gW_2 = <gy, T0>, T0 = x + R(C1)
    if (role != 2) {
        sfor<3>([&](auto rc) { ld_x(rc); });
        ld_g(IC<0>{});
        f32x2 C1[W1][P1];
        load_plane<W1>(A.saved, A.c_off[0], n, C, c, C1);
        f32x2 H1[W1][P];                                    // C1 rows resized horizontally, each built just before its first use
        f32x2 T0[5][P];
        WAcc w2;
        w2.zero();
        auto build = [&](auto dc) {                         // T0 row d into its ring slot
            constexpr int d = decltype(dc)::value;
            constexpr VT vt = vtab(MODE, W1, W, d);
#pragma unroll
            for (int i = 0; i < W1; ++i) {
                const bool first_use = (i == vt.i0 || i == vt.i1) && (d == 0 || (i != vtab(MODE, W1, W, d - 1).i0 && i != vtab(MODE, W1, W, d - 1).i1));
                if (first_use) resize_row<MODE, W1, W>(C1[i], H1[i]);
            }
            pin_raw(rx[d]);
#pragma unroll
            for (int j = 0; j < P; ++j) T0[d % 5][j] = f32x2{SafeLd<TIO>::cvt(rx[d][2 * j]), SafeLd<TIO>::cvt(rx[d][2 * j + 1])};
            add_resized_row<MODE, W1, W, P>(T0[d % 5], H1, d);
        };
        sfor<W>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t == 0) { build(IC<0>{}); build(IC<1>{}); build(IC<2>{}); }
            if constexpr (t + 3 < W) ld_x(IC<t + 3>{});
            if constexpr (t + 1 < W) ld_g(IC<t + 1>{});
            if constexpr (t > 0 && t + 2 < W) build(IC<t + 2>{});
            pin_raw(rgy[t]);
            f32x2 g[P];
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = f32x2{SafeLd<TGY>::cvt(rgy[t][2 * j]), SafeLd<TGY>::cvt(rgy[t][2 * j + 1])};
            wgrad_row<W>(g, t, [&](int r) -> const f32x2(&)[P] { return T0[r % 5]; }, w2);
#pragma unroll
            for (int u = 0; u < 5; ++u) { pin(w2.E[u]); pin(w2.O[u]); }
            RCX_FENCE;
        });
        store_wacc(A.part[3], n, C, c, w2);
        RCX_FENCE;
    }
    if (role == 1) return;

    // ---- (2) gT0 = conv_2^T(gy) -> the stash, and gC1 = R^T(gT0)
    float S[W][W];
    f32x2 gC1[W1][P1];
#pragma unroll
    for (int i = 0; i < W1; ++i)
#pragma unroll
        for (int j = 0; j < P1; ++j) gC1[i][j] = f32x2{0.f, 0.f};
    {
        constexpr int AHEAD = 3;
        sfor<AHEAD>([&](auto rc) { ld_g(rc); });
        Taps tf;
        load_taps<CT>(tf, A.wflip, nullptr, 3, C, vow, 0);
        f32x2 acc[5][P];
        sfor<W>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int s_ = 0; s_ < 5; ++s_) {
                const bool enters = (t == 0) ? (s_ <= 2) : (s_ == (t + 2) % 5 && t + 2 < W);
                if (enters) {
#pragma unroll
                    for (int j = 0; j < P; ++j) acc[s_][j] = f32x2{0.f, 0.f};
                }
            }
            if constexpr (t + AHEAD < W) ld_g(IC<t + AHEAD>{});
            pin_raw(rgy[t]);
            f32x2 row[P];
#pragma unroll
            for (int j = 0; j < P; ++j) row[j] = f32x2{SafeLd<TGY>::cvt(rgy[t][2 * j]), SafeLd<TGY>::cvt(rgy[t][2 * j + 1])};
            conv5_row<W>(row, t, tf, [&](int o) -> f32x2(&)[P] { return acc[o % 5]; });
#pragma unroll
            for (int d = 2; d >= 0; --d) {
                const int o = t - d;
                if (o < 0 || (d < 2 && t != W - 1)) continue;
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    S[o][2 * j] = stash(acc[o % 5][j].x);
                    S[o][2 * j + 1] = stash(acc[o % 5][j].y);
                }
                resizeT_row<MODE, W1, W>(acc[o % 5], o, gC1);
            }
#pragma unroll
            for (int o = 0; o < W; ++o) if (o > t - 2 && o <= t + 2 && t != W - 1) pin(acc[o % 5]);
            pin(gC1);
            RCX_FENCE;
        });
    }

    // ---- (3) the level-1 block on the 7x7 planes: F1 -> gF1
    DAcc wd;
    wd.zero();
    f32x2 F1[W1][P1];
    load_plane<W1>(A.saved, A.f_off[0], n, C, c, F1);
    level1_bwd<MODE, CT, W1, W2, false>(F1, gC1, A, A.f_off[1], A.c_off[1], n, C, c, vow, A.part[1], A.part[2], wd);

    // ---- (4) gW_d += <gF1, x>_s2 ; gx = gT0 + down^T(gF1)
    {
        constexpr int AHEAD = 3;
        sfor<AHEAD>([&](auto rc) { ld_x(rc); });
        Taps td;
        load_taps<CT>(td, A.wpack, nullptr, 0, C, vow, 0);
        sfor<W>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r + AHEAD < W) ld_x(IC<r + AHEAD>{});
            pin_raw(rx[r]);
            f32x2 xr[P];
#pragma unroll
            for (int j = 0; j < P; ++j) xr[j] = f32x2{SafeLd<TIO>::cvt(rx[r][2 * j]), SafeLd<TIO>::cvt(rx[r][2 * j + 1])};
            wgrad2_row<W, W1>(xr, r, F1, wd);
            f32x2 out[P];
#pragma unroll
            for (int j = 0; j < P; ++j) out[j] = f32x2{unstash(S[r][2 * j]), unstash(S[r][2 * j + 1])};
            downT_row<W, W1>(F1, r, td, out);
            typename PixSt<TIO>::packed pk[P];
#pragma unroll
            for (int j = 0; j < P; ++j) pk[j] = PixSt<TIO>::prep(out[j]);
            ra.row(gxb, r, [&](auto qc, gcptr base, unsigned voff, auto immc) {
                constexpr int q = decltype(qc)::value;
                PixSt<TIO>::st(base + decltype(immc)::value + voff, pk[q >> 1], q & 1);
            });
#pragma unroll
            for (int u = 0; u < 5; ++u) pin(wd.a[u]);
            RCX_FENCE;
        });
        wgrad2_bias<W1>(F1, wd);
    }
    store_wacc(A.part[0], n, C, c, wd);
}

template <int MODE, int CT, typename TIO, typename TGY = float>
static hipError_t launch14(const BwdArgs& A, hipStream_t s)
{
    const unsigned planes = (unsigned)(A.N * ((A.C + 63) / 64));
    const char* v = rcx::opt::value(rcx::opt::BWD_SPLIT);                                // A/B switch: 0 = one wave per plane always
    // two waves per plane while that still fits the chip's 1024 SIMDs in one round (and the split grid keeps whole XCD rounds)
    BwdArgs B = A;
    B.split = !(v && *v == '0') && planes * 2 <= 1024 && planes % 8 == 0;
    hipLaunchKernelGGL((k_recconv_bwd_cpl14<MODE, CT, TIO, TGY>), dim3(B.split ? planes * 2 : planes), dim3(64), 0, s, B);
    return hipGetLastError();
}
template <int MODE, typename TIO, typename TGY = float>
static hipError_t launch14_c(const BwdArgs& A, hipStream_t s)
{
    if (A.C == 256) return launch14<MODE, 256, TIO, TGY>(A, s);
    return launch14<MODE, 0, TIO, TGY>(A, s);
}

template <int MODE, int CT, typename TIO, typename TGY = float>
static hipError_t launch7(const BwdArgs& A, hipStream_t s)
{
    const unsigned grid = (unsigned)(A.N * ((A.C + 63) / 64));
    hipLaunchKernelGGL((k_recconv_bwd_cpl7<MODE, CT, TIO, TGY>), dim3(grid), dim3(64), 0, s, A);
    return hipGetLastError();
}
template <int MODE, typename TIO, typename TGY = float>
static hipError_t launch7_c(const BwdArgs& A, hipStream_t s)
{
    if (A.C == 512) return launch7<MODE, 512, TIO, TGY>(A, s);
    return launch7<MODE, 0, TIO, TGY>(A, s);
}

}  // namespace cplbwd

// the fused backward applies where the fused training forward does, and needs one partial row per image in a 512-row slot
bool cplbwd_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    const char* v = rcx::opt::value(rcx::opt::BWD_FUSED);
    if (v && *v == '0') return false;
    if (N > 512) return false;
    return cpl7b_applicable(N, C, H, W, level, k, dtype) || cpl14_applicable(N, C, H, W, level, k, dtype);
}

// gy_dt: RCX_DTYPE_F32, or bfloat16 with a bfloat16 block (dL/dy as autograd hands it over under autocast: no float32 copy; round 6)
hipError_t cplbwd_recconv(const void* x, const void* gy, const float* wpack, const float* wflip, const void* saved,
                          const size_t* f_off, const size_t* c_off, void* gx, float* const* part,
                          int N, int C, int H, int level, int mode, int dtype, hipStream_t s, int gy_dt)
{
    if (gy_dt != 0 && !(gy_dt == 1 && dtype == 1)) return hipErrorInvalidValue;
    cplbwd::BwdArgs A{};
    A.x = x; A.gy = gy; A.wpack = wpack; A.wflip = wflip; A.saved = (const char*)saved; A.gx = gx; A.N = N; A.C = C;
    for (int l = 1; l <= level; ++l) { A.f_off[l - 1] = f_off[l]; A.c_off[l - 1] = c_off[l]; }
    for (int j = 0; j < level + 2; ++j) A.part[j] = part[j];
#define RCX_BW14(MD_) (gy_dt == 1 ? cplbwd::launch14_c<MD_, bf16_t, bf16_t>(A, s) : dtype == 1 ? cplbwd::launch14_c<MD_, bf16_t>(A, s) : dtype == 2 ? cplbwd::launch14_c<MD_, f16_t>(A, s) : cplbwd::launch14_c<MD_, float>(A, s))
    if (H == 14) return mode == 1 ? RCX_BW14(1) : RCX_BW14(0);
#undef RCX_BW14
#define RCX_BW7(MD_) (gy_dt == 1 ? cplbwd::launch7_c<MD_, bf16_t, bf16_t>(A, s) : dtype == 1 ? cplbwd::launch7_c<MD_, bf16_t>(A, s) : dtype == 2 ? cplbwd::launch7_c<MD_, f16_t>(A, s) : cplbwd::launch7_c<MD_, float>(A, s))
    return mode == 1 ? RCX_BW7(1) : RCX_BW7(0);
#undef RCX_BW7
}

}  // namespace rcx
