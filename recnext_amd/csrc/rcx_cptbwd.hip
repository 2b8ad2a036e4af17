// Backward of the FINE levels of the RecConv2d block on the 56x56 and 28x28 planes (the adjoint of model/recnext.py:31-34 and of the
// first steps of the down ladder, :27-29), channel per lane, tiled like the forward's single steps (rcx_upcpt.hip):
// a wave = 64 channels of one 14 x 14 tile of the plane, a lane = one channel, nothing shared between lanes, no LDS, no barrier.
//
// With g the gradient of a level's conv output (gy at level 0, gC_l below), K^ the conv with its taps flipped, D the shared stride-2
// conv, R the exact-2x resize (rcx_bwd.hip has the whole recursion):
//
//   k_bwd_gc   gC = R^T (K^ g)            the gradient handed DOWN to the next level, at half resolution: the full-resolution gT = K^ g
//                                         never exists in memory -- a tile forms it on its 14 x 14 pixels plus the one-pixel ring R^T
//                                         reads (16 x 16 outputs from a 20 x 20 window of g) and leaves its 7 x 7 coarse pixels;
//   k_bwd_gx   gF = K^ g + D^T G          the gradient handed UP: G = the total gradient of the next level's plane (known once the
//                                         coarser levels are done); at level 0 gF is gx, rounded once at the store.  K^ g is formed
//                                         a second time here instead of being kept: 25 packed FMAs per pixel pair against a float32
//                                         plane written and read back (at 56 x 56 x 64 channels x 128 images: 206 MB).
//
// The per-step schedule these replace ran K^ g (float32 in and out), R^T and D^T as three gathers of one thread per four channels
// of a pixel: 46 + 38 + 71 us at 128 x 64 x 56 x 56 where the two kernels here move 0.13 GB (profiles/r06_train_*).
//
// R^T of the exact 2x bilinear step: coarse pixel i collects fine pixels 2i-1 .. 2i+2 with weights 1/4, 3/4, 3/4, 1/4; at the border the
// forward clamps its source index, so the fine pixel that would lie outside gives its weight to the edge pixel (1/4 + 3/4 = 1):
// resolved per tile with wave-uniform weights, no branch.  Nearest: weights 0, 1, 1, 0.
// Arithmetic: float32 throughout, packed pairs of horizontally adjacent pixels (v_pk_fma_f32), input-row stationary with five
// accumulator rows in flight, exactly as the forward's pass 2; loads are ordered loads the compiler counts itself (SafeLd), requested
// AHEAD rows before they are used.
#include "rcx_cplbwd_pieces.h"
#include "rcx_opts.h"

#ifndef RCX_GX_AHEAD
#define RCX_GX_AHEAD 1
#endif
#ifndef RCX_GC_AHEAD
#define RCX_GC_AHEAD 1
#endif

namespace rcx {
namespace cptbwd {

using namespace cplbwd;

template <int H> struct Geo {
    static constexpr int W = H, NT = W / 14, NB = H / 14, Hc = H / 2, Wc = W / 2;
    static_assert(H % 14 == 0, "whole 14 x 14 tiles");
};

struct Unit {
    int n, cb, tr, tc, c;
    bool live;
    unsigned cl;
};
template <int H>
__device__ __forceinline__ bool decode_unit(Unit& u, int N, int C)
{
    using G = Geo<H>;
    const int lane = (int)(threadIdx.x & 63);
    const int nb = (C + 63) / 64;
    const unsigned total = (unsigned)N * nb * G::NB * G::NT;
    const unsigned unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (unit >= total) return false;
    u.tc = (int)(unit % (unsigned)G::NT);
    unsigned q = unit / (unsigned)G::NT;
    u.tr = (int)(q % (unsigned)G::NB);
    q /= (unsigned)G::NB;
    u.cb = (int)(q % (unsigned)nb);
    u.n = (int)(q / (unsigned)nb);
    u.c = u.cb * 64 + lane;
    u.live = u.c < C;
    u.cl = (unsigned)(u.live ? u.c : C - 1);          // ragged last block: the spare lanes shadow the last channel and store nothing
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_bwd_gx: out = K^ g + D^T G on a 14 x 14 tile.  g: N x H x H x C (TG), G: N x H/2 x H/2 x C float32 (HAS_D), out: TO.
// wf: this conv's 25 x C flipped taps; wd: the shared down conv's 25 x C taps (as the forward applies them).
// AH = g rows requested ahead of use, OCC = workgroups per CU the register budget is set for: <1, 2> (at most 256 registers: two waves per SIMD) where a
// launch fills the chip; <3, 1> where it has at most one wave per SIMD anyway (the float32 level-1 plane of the 56 x 56 block at batch 128: 512 waves) and
// only requests in flight hide the memory latency
template <typename TG, typename TO, int H, bool HAS_D, int AH = RCX_GX_AHEAD, int OCC = 2>
__global__ __launch_bounds__(256, OCC)
void k_bwd_gx(const TG* __restrict__ g, const float* __restrict__ Gc, TO* __restrict__ out, const float* __restrict__ wf,
              const float* __restrict__ wd, int N, int C)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, Hc = GE::Hc, Wc = GE::Wc, AHEAD = AH, NS = 18;
    Unit U;
    if (!decode_unit<H>(U, N, C)) return;
    const int r0 = 14 * U.tr, c0 = 14 * U.tc;
    const size_t pixg = (size_t)C * sizeof(TG), pixf = (size_t)C * 4, pixo = (size_t)C * sizeof(TO);
    const unsigned vog = U.cl * (unsigned)sizeof(TG), vof = U.cl * 4u, voo = U.cl * (unsigned)sizeof(TO);
    const gcptr gb = (gcptr)g + (size_t)U.n * H * W * pixg;
    const gcptr Gb = (gcptr)Gc + (size_t)U.n * Hc * Wc * pixf;
    const gcptr ob = (gcptr)out + (size_t)U.n * H * W * pixo;

    uint32_t rg[NS][18];                                   // g rows as loaded: local row s = image row r0 - 2 + s, local column q = image column c0 - 2 + q
    uint32_t rG[9][9];                                     // G rows: local row m = coarse row 7 tr - 1 + m, local column q = coarse column 7 tc - 1 + q
    auto ld_g = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        int r = r0 - 2 + s;
        r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);          // rows and columns outside the plane: a valid address, zeroed when the row is taken
        const gcptr rowp = gb + (size_t)r * W * pixg;
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            int col = c0 + q - 2;
            if (q < 2) col = col < 0 ? 0 : col;
            if (q >= 16) col = col > W - 1 ? W - 1 : col;
            rg[s][q] = SafeLd<TG>::ld(rowp + (size_t)col * pixg + vog);
        }
    };
    auto ld_G = [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        int r = 7 * U.tr - 1 + m;
        r = r < 0 ? 0 : (r > Hc - 1 ? Hc - 1 : r);
        const gcptr rowp = Gb + (size_t)r * Wc * pixf;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            int col = 7 * U.tc - 1 + q;
            if (q == 0) col = col < 0 ? 0 : col;
            if (q == 8) col = col > Wc - 1 ? Wc - 1 : col;
            rG[m][q] = SafeLd<float>::ld(rowp + (size_t)col * pixf + vof);
        }
    };
    // prologue: the first rows, then the taps (ordinary loads the compiler places)
    if constexpr (HAS_D) ld_G(IC<0>{});
    sfor<AHEAD>([&](auto sc) { ld_g(sc); });
    Taps tf, td;
    load_taps<0>(tf, wf, nullptr, 0, C, vof, 0);
    if constexpr (HAS_D) load_taps<0>(td, wd, nullptr, 0, C, vof, 0);
#pragma unroll
    for (int u = 0; u < 5; ++u) { pin(tf.p[u]); if constexpr (HAS_D) pin(td.p[u]); }

    const f32x2 keep_lo = splat(c0 == 0 ? 0.f : 1.f), keep_hi = splat(c0 + 14 == W ? 0.f : 1.f);
    const float Gkeep_lo = U.tc == 0 ? 0.f : 1.f, Gkeep_hi = 7 * U.tc + 7 == Wc ? 0.f : 1.f;
    f32x2 acc[5][7];

    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < NS) ld_g(IC<s + AHEAD>{});
        // D^T is input-row stationary too: coarse row i (local m = i + 1) feeds the fine rows o = 2i - 2 .. 2i + 2 (tap row u = o + 2 - 2i), exactly
        // the accumulator rows in flight in iteration s = 2i + 2 = 2m: G row m is taken there, and requested one even iteration earlier
        if constexpr (HAS_D && (s & 1) == 0 && s / 2 + 1 <= 8) ld_G(IC<(s / 2 + 1 <= 8 ? s / 2 + 1 : 8)>{});
        pin_raw(rg[s]);
        f32x2 row[9], odd[8];
        {
            const int r = r0 - 2 + s;
            const f32x2 keep = splat((s >= 2 && s < 16) || (r >= 0 && r < H) ? 1.f : 0.f);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                row[k] = f32x2{SafeLd<TG>::cvt(rg[s][2 * k]), SafeLd<TG>::cvt(rg[s][2 * k + 1])};
                if constexpr (s < 2 || s >= 16) row[k] = row[k] * keep;
            }
            row[0] = row[0] * keep_lo;
            row[8] = row[8] * keep_hi;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) odd[j] = shift1(row[j], row[j + 1]);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int o = s - u;
            if (o < 0 || o > 13) continue;
            f32x2(&a)[7] = acc[o % 5];
            if (u == 0) {
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = row[j] * splat(tf.at(0, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), a[j]);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
        }
        if constexpr (HAS_D && (s & 1) == 0 && s / 2 <= 8) {
            constexpr int m = s / 2 <= 8 ? s / 2 : 8;
            pin_raw(rG[m]);
            float Gm[9];                                    // zero outside the plane: the adjoint sums over existing coarse pixels only
            {
                const int r = 7 * U.tr - 1 + m;
                const float keep = (r >= 0 && r < Hc) ? 1.f : 0.f;
#pragma unroll
                for (int q = 0; q < 9; ++q) Gm[q] = (m == 0 || m == 8) ? __uint_as_float(rG[m][q]) * keep : __uint_as_float(rG[m][q]);
                Gm[0] *= Gkeep_lo;
                Gm[8] *= Gkeep_hi;
            }
            // out(2j, 2j+1) += G[i][j+1] (w0, w1) + G[i][j] (w2, w3) + (G[i][j-1] w4, 0); coarse column j of the tile = local q = j + 1
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int o = 2 * (m - 1) - 2 + u;
                if (o < 0 || o > 13) continue;
                f32x2(&a)[7] = acc[o % 5];
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(splat(Gm[j + 2]), td.p[u][0], a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = pfma(splat(Gm[j + 1]), td.p[u][1], a[j]);
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j].x = fmaf(Gm[j], td.p[u][2].x, a[j].x);
            }
        }
        if constexpr (s >= 4) {
            constexpr int o = s - 4;
            if (U.live) {
                const gcptr rowp = ob + ((size_t)(r0 + o) * W + c0) * pixo + voo;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const typename PixSt<TO>::packed pk = PixSt<TO>::prep(acc[o % 5][j]);
                    PixSt<TO>::st(rowp + (size_t)(2 * j) * pixo, pk, 0);
                    PixSt<TO>::st(rowp + (size_t)(2 * j + 1) * pixo, pk, 1);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 14; ++o) if (o > s - 4 && o <= s) pin(acc[o % 5]);
        RCX_FENCE;
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_bwd_gc: gC = R^T (K^ g) on the 7 x 7 coarse pixels of a 14 x 14 tile.  g: N x H x H x C (TG); gC: N x H/2 x H/2 x C float32.
// Local frames: g row s = image row r0 - 3 + s, column q = image column c0 - 3 + q (20 x 20); gT row o = image row r0 - 1 + o,
// column p = image column c0 - 1 + p (16 x 16: pairs start at an ODD image column); gT(o, p) = sum_{u,v} K^[u][v] g(o + u, p + v).
template <int MODE, typename TG, int H, int AH = RCX_GC_AHEAD, int OCC = 2>
__global__ __launch_bounds__(256, OCC)
void k_bwd_gc(const TG* __restrict__ g, float* __restrict__ gC, const float* __restrict__ wf, int N, int C)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, Hc = GE::Hc, Wc = GE::Wc, AHEAD = AH, NS = 20;
    constexpr float WQ = MODE == 1 ? 0.f : 0.25f, WT = MODE == 1 ? 1.f : 0.75f;
    Unit U;
    if (!decode_unit<H>(U, N, C)) return;
    const int r0 = 14 * U.tr, c0 = 14 * U.tc;
    const size_t pixg = (size_t)C * sizeof(TG), pixf = (size_t)C * 4;
    const unsigned vog = U.cl * (unsigned)sizeof(TG), vof = U.cl * 4u;
    const gcptr gb = (gcptr)g + (size_t)U.n * H * W * pixg;
    const gcptr cb = (gcptr)gC + (size_t)U.n * Hc * Wc * pixf;

    uint32_t rg[NS][20];
    auto ld_g = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        int r = r0 - 3 + s;
        r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        const gcptr rowp = gb + (size_t)r * W * pixg;
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            int col = c0 + q - 3;
            if (q < 3) col = col < 0 ? 0 : col;
            if (q >= 17) col = col > W - 1 ? W - 1 : col;
            rg[s][q] = SafeLd<TG>::ld(rowp + (size_t)col * pixg + vog);
        }
    };
    sfor<AHEAD>([&](auto sc) { ld_g(sc); });
    Taps tf;
    load_taps<0>(tf, wf, nullptr, 0, C, vof, 0);
#pragma unroll
    for (int u = 0; u < 5; ++u) pin(tf.p[u]);

    // columns of g outside the plane are zero padding of K^'s input: local q = 0, 1, 2 at the left edge, 17, 18, 19 at the right one
    const bool left = c0 == 0, right = c0 + 14 == W, top = r0 == 0, bottom = r0 + 14 == H;
    const f32x2 m0 = splat(left ? 0.f : 1.f), m1 = f32x2{left ? 0.f : 1.f, 1.f};                  // pairs (0,1), (2,3)
    const f32x2 m8 = f32x2{1.f, right ? 0.f : 1.f}, m9 = splat(right ? 0.f : 1.f);                // pairs (16,17), (18,19)
    // R^T, vertical: gT row o feeds coarse rows (o even) o/2 with WQ' and o/2 - 1 with WT', (o odd) (o-1)/2 with WT' and (o-3)/2 with WQ';
    // the clamped borders move the outside row's weight to the edge row: rows 0 / 15 get 0, rows 1 / 14 get WT + WQ there
    const float wv_o0 = top ? 0.f : WQ, wv_o1 = top ? WT + WQ : WT, wv_o14 = bottom ? WT + WQ : WT, wv_o15 = bottom ? 0.f : WQ;
    // horizontal: coarse column j = dot(A[j], (WQ, WT)) + dot(A[j+1], (WT, WQ)) over the pairs A[k] = (p = 2k, 2k + 1); same border rule
    const f32x2 hA0 = f32x2{left ? 0.f : WQ, left ? WT + WQ : WT}, hB6 = f32x2{right ? WT + WQ : WT, right ? 0.f : WQ};

    f32x2 acc[5][8];
    float V[2][7];
    const unsigned live = U.live ? 1u : 0u;
    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < NS) ld_g(IC<s + AHEAD>{});
        pin_raw(rg[s]);
        f32x2 row[10], odd[9];
        {
            const int r = r0 - 3 + s;
            const f32x2 keep = splat((s >= 3 && s < 17) || (r >= 0 && r < H) ? 1.f : 0.f);
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                row[k] = f32x2{SafeLd<TG>::cvt(rg[s][2 * k]), SafeLd<TG>::cvt(rg[s][2 * k + 1])};
                if constexpr (s < 3 || s >= 17) row[k] = row[k] * keep;
            }
            row[0] = row[0] * m0; row[1] = row[1] * m1; row[8] = row[8] * m8; row[9] = row[9] * m9;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) odd[j] = shift1(row[j], row[j + 1]);
        // g row s meets gT rows o = s - u
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int o = s - u;
            if (o < 0 || o > 15) continue;
            f32x2(&a)[8] = acc[o % 5];
            if (u == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = row[j] * splat(tf.at(0, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = pfma(row[j], splat(tf.at(u, 0)), a[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(odd[j], splat(tf.at(u, 1)), a[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(row[j + 1], splat(tf.at(u, 2)), a[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(odd[j + 1], splat(tf.at(u, 3)), a[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = pfma(row[j + 2], splat(tf.at(u, 4)), a[j]);
        }
        if constexpr (s >= 4) {
            constexpr int o = s - 4;                        // gT row o is complete: its horizontal adjoint first (7 values), then the vertical one on those
            const f32x2(&a)[8] = acc[o % 5];
            float h[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const f32x2 wa = j == 0 ? hA0 : f32x2{WQ, WT}, wb = j == 6 ? hB6 : f32x2{WT, WQ};
                const f32x2 t = pfma(a[j + 1], wb, a[j] * wa);
                h[j] = t.x + t.y;
            }
            // coarse row i collects o = 2i (first), 2i + 1, 2i + 2, 2i + 3 (last)
            if constexpr ((o & 1) == 0) {
                if constexpr (o / 2 <= 6) {                 // opens coarse row o / 2
                    const float w = o == 0 ? wv_o0 : WQ;
#pragma unroll
                    for (int j = 0; j < 7; ++j) V[(o / 2) & 1][j] = h[j] * w;
                }
                if constexpr (o / 2 - 1 >= 0) {
                    const float w = o == 14 ? wv_o14 : WT;
#pragma unroll
                    for (int j = 0; j < 7; ++j) V[(o / 2 - 1) & 1][j] = fmaf(h[j], w, V[(o / 2 - 1) & 1][j]);
                }
            } else {
                if constexpr ((o - 1) / 2 <= 6) {
                    const float w = o == 1 ? wv_o1 : WT;
#pragma unroll
                    for (int j = 0; j < 7; ++j) V[((o - 1) / 2) & 1][j] = fmaf(h[j], w, V[((o - 1) / 2) & 1][j]);
                }
                if constexpr (o >= 3) {
                    constexpr int i = (o - 3) / 2;          // closes coarse row i
                    const float w = o == 15 ? wv_o15 : WQ;
                    if (live) {
                        const gcptr rowp = cb + ((size_t)(7 * U.tr + i) * Wc + 7 * U.tc) * pixf + vof;
#pragma unroll
                        for (int j = 0; j < 7; ++j) gstore<float>(rowp + (size_t)j * pixf, fmaf(h[j], w, V[i & 1][j]));
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) if (o > s - 4 && o <= s) pin(acc[o % 5]);
#pragma unroll
        for (int j = 0; j < 7; ++j) { pin(V[0][j]); pin(V[1][j]); }
        RCX_FENCE;
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_wgrad_d: the shared stride-2 conv's weight gradient from one level, gW_d[u][v] += sum_{o,i} G[o][i] a[2o + u - 2][2i + v - 2]
// (model/recnext.py:21, :28), on the same tiles: a wave = 64 channels of a 14 x 14 tile of a = the 7 x 7 pixels of G it produced.
// rcx_cplwgrad.hip's k_wgrad2_cpl gives a wave a 14 x 14 tile of G (28 x 28 of a): 512 waves at 128 x 64 x 56 x 56, 330 registers, one
// wave on every second SIMD -- 50 us for 0.1 GB.  Here: four times the waves, G's 49 values resident, a-row stationary with the taps paired
// against a's aligned pairs (rcx_cplbwd_pieces.h), rows requested AHEAD before use.  The NT column tiles of a band share a workgroup and add
// their 26 sums per channel through LDS in a fixed order: one partial row per (image, 14-row band), reduced over the batch by k_wgrad_reduce_jobs.
template <typename TA, int H>
__global__ __launch_bounds__(64 * (H / 14))
void k_wgrad_d(const TA* __restrict__ a, const float* __restrict__ Gc, float* __restrict__ partial, int N, int C)
{
    using GE = Geo<H>;
    constexpr int W = GE::W, NT = GE::NT, NB = GE::NB, Hc = GE::Hc, Wc = GE::Wc, AHEAD = 2, NS = 17;
    __shared__ float red[NT][26][64];
    const int tile = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int nb = (C + 63) / 64;
    const unsigned unit = blockIdx.x;
    const int cb = (int)(unit % (unsigned)nb), band = (int)((unit / (unsigned)nb) % (unsigned)NB), n = (int)(unit / (unsigned)(nb * NB));
    const int c = cb * 64 + lane;
    const bool live = c < C;
    const unsigned cl = (unsigned)(live ? c : C - 1);
    const int r0 = 14 * band, c0 = 14 * tile;
    const size_t pixa = (size_t)C * sizeof(TA), pixf = (size_t)C * 4;
    const unsigned voa = cl * (unsigned)sizeof(TA), vof = cl * 4u;
    const gcptr ab = (gcptr)a + (size_t)n * H * W * pixa;
    const gcptr Gb = (gcptr)Gc + ((size_t)n * Hc * Wc + (size_t)(7 * band) * Wc + 7 * tile) * pixf;

    uint32_t ra[NS][18], rG[7][7];
    auto ld_a = [&](auto sc) {                                // local row s = image row r0 - 2 + s, local column q = image column c0 - 2 + q (clamped; zeroed when taken)
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
    sfor<7>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
#pragma unroll
        for (int i = 0; i < 7; ++i) rG[o][i] = SafeLd<float>::ld(Gb + ((size_t)o * Wc + i) * pixf + vof);
    });
    sfor<AHEAD>([&](auto sc) { ld_a(sc); });
    const f32x2 keep_lo = splat(c0 == 0 ? 0.f : 1.f), keep_hi = splat(c0 + 14 == W ? 0.f : 1.f);
    DAcc acc;
    acc.zero();
    float G[7][7];
    sfor<7>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
        pin_raw(rG[o]);
#pragma unroll
        for (int i = 0; i < 7; ++i) { G[o][i] = __uint_as_float(rG[o][i]); if (i & 1) acc.bs.y += G[o][i]; else acc.bs.x += G[o][i]; }
    });
    sfor<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < NS) ld_a(IC<s + AHEAD>{});
        pin_raw(ra[s]);
        f32x2 ar[9];
        {
            const int r = r0 - 2 + s;
            const f32x2 keep = splat((s >= 2 && s < 16) || (r >= 0 && r < H) ? 1.f : 0.f);
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                ar[m] = f32x2{SafeLd<TA>::cvt(ra[s][2 * m]), SafeLd<TA>::cvt(ra[s][2 * m + 1])};
                if constexpr (s < 2 || s >= 16) ar[m] = ar[m] * keep;
            }
            ar[0] = ar[0] * keep_lo;
            ar[8] = ar[8] * keep_hi;
        }
        // a row s = 2o + u: the G rows o whose window covers it, tap row u; G column i against the pairs ar[i + k] = a columns 2i + 2k, 2i + 2k + 1
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int o = 0; o < 7; ++o) {
                const int u = s - 2 * o;
                if (u < 0 || u > 4) continue;
                const f32x2 gv = splat(G[o][i]);
#pragma unroll
                for (int k = 0; k < 3; ++k) acc.a[u][k] = pfma(gv, ar[i + k], acc.a[u][k]);
            }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) pin(acc.a[u]);
        RCX_FENCE;
    });
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

#ifndef RCX_CPTBWD_KERNELS_ONLY            // (a tuning harness instantiates single kernels)
// tile-waves up to which a launch leaves at most one wave per SIMD (256 CUs x 4): the deep-prefetch instantiations (RCX_BWD_CPT=shallow: never)
static long long few_units()
{
    const char* v = rcx::opt::value(rcx::opt::BWD_CPT);
    return (v && *v == 's') ? 0 : 1024;
}
template <typename TG, typename TO, int H>
static hipError_t launch_gx(const void* g, const float* G, void* out, const float* wf, const float* wd, int N, int C, hipStream_t s)
{
    const long long units = (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    if (!G) hipLaunchKernelGGL((k_bwd_gx<TG, TO, H, false>), grid, block, 0, s, (const TG*)g, G, (TO*)out, wf, wd, N, C);
    else if (units <= few_units()) hipLaunchKernelGGL((k_bwd_gx<TG, TO, H, true, 3, 1>), grid, block, 0, s, (const TG*)g, G, (TO*)out, wf, wd, N, C);
    else hipLaunchKernelGGL((k_bwd_gx<TG, TO, H, true>), grid, block, 0, s, (const TG*)g, G, (TO*)out, wf, wd, N, C);
    return hipGetLastError();
}

template <typename TG, int H>
static hipError_t launch_gc(const void* g, float* gC, const float* wf, int N, int C, int mode, hipStream_t s)
{
    const long long units = (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14);
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    if (units <= few_units()) {
        if (mode == 1) hipLaunchKernelGGL((k_bwd_gc<1, TG, H, 3, 1>), grid, block, 0, s, (const TG*)g, gC, wf, N, C);
        else hipLaunchKernelGGL((k_bwd_gc<0, TG, H, 3, 1>), grid, block, 0, s, (const TG*)g, gC, wf, N, C);
    } else {
        if (mode == 1) hipLaunchKernelGGL((k_bwd_gc<1, TG, H>), grid, block, 0, s, (const TG*)g, gC, wf, N, C);
        else hipLaunchKernelGGL((k_bwd_gc<0, TG, H>), grid, block, 0, s, (const TG*)g, gC, wf, N, C);
    }
    return hipGetLastError();
}

template <typename TA, int H>
static hipError_t launch_wd(const void* a, const float* G, float* partial, int N, int C, hipStream_t s)
{
    const unsigned grid = (unsigned)(N * (H / 14) * ((C + 63) / 64));
    hipLaunchKernelGGL((k_wgrad_d<TA, H>), dim3(grid), dim3(64 * (H / 14)), 0, s, (const TA*)a, G, partial, N, C);
    return hipGetLastError();
}
#endif
}  // namespace cptbwd

#ifndef RCX_CPTBWD_KERNELS_ONLY
// the shared down conv's weight gradient from (a: H x H of a_dt, G: H/2 x H/2 float32): one partial row of 26 C sums per (image, 14-row band of a)
hipError_t bwd_wgrad_d_cpt(const void* a, int a_dt, const float* G, float* partial, int N, int C, int H, hipStream_t s, int* rows_out)
{
    if (rows_out) *rows_out = N * (H / 14);
#define RCX_WD(TA_) (H == 56 ? cptbwd::launch_wd<TA_, 56>(a, G, partial, N, C, s) : cptbwd::launch_wd<TA_, 28>(a, G, partial, N, C, s))
    if (a_dt == 1) return RCX_WD(bf16_t);
    if (a_dt == 2) return RCX_WD(f16_t);
    return RCX_WD(float);
#undef RCX_WD
}

bool bwd_cpt_applicable(int N, int C, int H, int W, int k)
{
    if (rcx::opt::is_zero(rcx::opt::BWD_CPT)) return false;
    return k == 5 && H == W && (H == 56 || H == 28) && N >= 1 && C >= 1 && (long long)N * ((C + 63) / 64) * (H / 14) * (H / 14) < (1LL << 31);
}

// out (out_dt) = K^ g + D^T G   (G == nullptr: the conv adjoint alone)
hipError_t bwd_gx_cpt(const void* g, int g_dt, const float* G, void* out, int out_dt, const float* wf, const float* wd, int N, int C, int H, hipStream_t s)
{
#define RCX_GX(TG_, TO_) (H == 56 ? cptbwd::launch_gx<TG_, TO_, 56>(g, G, out, wf, wd, N, C, s) : cptbwd::launch_gx<TG_, TO_, 28>(g, G, out, wf, wd, N, C, s))
    if (g_dt == 0 && out_dt == 0) return RCX_GX(float, float);
    if (g_dt == 1 && out_dt == 1) return RCX_GX(bf16_t, bf16_t);
    if (g_dt == 2 && out_dt == 2) return RCX_GX(f16_t, f16_t);
    if (g_dt == 0 && out_dt == 1) return RCX_GX(float, bf16_t);
    if (g_dt == 0 && out_dt == 2) return RCX_GX(float, f16_t);
#undef RCX_GX
    return hipErrorInvalidValue;
}

// gC (float32, H/2 x H/2) = R^T (K^ g)
hipError_t bwd_gc_cpt(const void* g, int g_dt, float* gC, const float* wf, int N, int C, int H, int mode, hipStream_t s)
{
#define RCX_GC(TG_) (H == 56 ? cptbwd::launch_gc<TG_, 56>(g, gC, wf, N, C, mode, s) : cptbwd::launch_gc<TG_, 28>(g, gC, wf, N, C, mode, s))
    if (g_dt == 1) return RCX_GC(bf16_t);
    if (g_dt == 2) return RCX_GC(f16_t);
    return RCX_GC(float);
#undef RCX_GC
}
#endif

}  // namespace rcx
