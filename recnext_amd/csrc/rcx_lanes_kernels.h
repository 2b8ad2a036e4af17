// Kernels and launchers of the register-resident RecConv2d schedule (included by rcx_lanes.hip: 7 * 2^k planes, and by
// rcx_lanes16.hip: 16 * 2^k planes -- two translation units so that they compile in parallel).
#pragma once
#include "rcx_lanes.h"
#include "rcx_opts.h"
#include "rcx_launch.h"

#ifndef RCX_LSTAMP
#define RCX_LSTAMP(id) do { } while (0)
#endif
#ifdef RCX_STAMPS
#define RCX_LABLATE(a, bit) ((a).ablate & (bit))
#else
#define RCX_LABLATE(a, bit) 0
#endif

namespace rcx {
namespace lanes {

struct LanesArgs {
    int ablate;        // diagnostic build only (RCX_LANES_ABLATE): 1 = skip the arithmetic, 2 = skip the global loads, 4 = skip the stores
    int N, C;
    int nblk;          // channel blocks per image (C / CBW)
    int ni;            // images per workgroup (consecutive)
    int has_bias;
};

// One workgroup of NW waves = (block of CBW = NW * 64/LPC channels, group of `ni` consecutive images).  Taps are staged
// once per workgroup; the next image's 16-byte chunks are prefetched into registers while the current one is computed.
// Every LDS offset is a compile-time constant (immediate offsets, no address registers).
template <int W0, int LEVEL, int LPC, int MODE, int NW, typename TIO>
__global__ __launch_bounds__(NW * 64)
void k_recconv_lanes(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack, LanesArgs a)
{
    constexpr int LA = lanes_active(W0, LPC);
    constexpr int B0 = W0 / LA;
    static_assert(B0 * LA == W0, "plane width must be LA * B0");
    constexpr int HW = W0 * W0;
    constexpr int CPW = 64 / LPC;                 // channels per wave
    constexpr int CBW = NW * CPW;                 // channels per workgroup
    constexpr int NT = NW * 64;
    constexpr int NCONV = LEVEL + 2;
    constexpr int ESZ = (int)sizeof(TIO);
    constexpr int PITCH = CBW * ESZ + 16;         // bytes per pixel of the raw LDS image
    constexpr int CPP = CBW * ESZ / 16;           // 16-byte chunks per pixel
    static_assert(CPP >= 1 && (CPP & (CPP - 1)) == 0, "channel block must be a power-of-two number of 16-byte chunks");
    constexpr int NCHUNKS = HW * CPP;
    constexpr int STAGE = (NCHUNKS + NT - 1) / NT;   // chunks per thread and image
    constexpr int TAPS_BYTES = NCONV * 26 * CBW * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* taps = reinterpret_cast<float*>(smem);                 // [NCONV][26][CBW]
    unsigned char* img = smem + TAPS_BYTES;                        // [HW][PITCH]
    float* xarea = reinterpret_cast<float*>(img + HW * PITCH);     // [CBW][xch_stride]: neighbour-exchange lines (rcx_lanes.h)
    constexpr int XFLOATS = RCX_XCH_LDS ? CBW * xch_stride(LPC, B0) : 0;

    const int tid = threadIdx.x;
    // workgroup -> (channel block, image group): the channel blocks of one image group get ids that are equal mod 8,
    // i.e. land on the same XCD (round-robin dispatch) close in time and share its L2 lines
    const int cb = (blockIdx.x % (8 * a.nblk)) / 8;
    const int n0 = ((blockIdx.x / (8 * a.nblk)) * 8 + blockIdx.x % 8) * a.ni;
    const int n1 = n0 + a.ni < a.N ? n0 + a.ni : a.N;
    const int c0 = cb * CBW;
    const size_t img_stride = (size_t)HW * a.C;               // elements per image

    // chunk <-> (pixel, part) of this thread, fixed for all images
    int g_off[STAGE], l_off[STAGE];
    bool have[STAGE];
    sfor<STAGE>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        have[i] = (i + 1) * NT <= NCHUNKS || cidx < NCHUNKS;
        cidx = have[i] ? cidx : NCHUNKS - 1;
        const int p = cidx / CPP, part = cidx % CPP;
        g_off[i] = p * a.C * ESZ + part * 16;
        l_off[i] = lds_slot<W0, B0, LA>(p) * PITCH + part * 16;
    });
    RCX_LSTAMP(32);
    u32x4 v[STAGE];
    auto prefetch = [&](int n) RCX_INL {
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(x + (size_t)n * img_stride + c0);
        if (RCX_LABLATE(a, 2)) { sfor<STAGE>([&](auto I) RCX_INL { v[decltype(I)::value] = u32x4{0u, 0u, 0u, 0u}; }); return; }
        sfor<STAGE>([&](auto I) RCX_INL { v[decltype(I)::value] = *reinterpret_cast<const u32x4*>(xg + g_off[decltype(I)::value]); });
    };
    if (n0 < n1) prefetch(n0);

    // taps + bias rows -> LDS: rows of CBW floats, float4 per thread, all loads of a thread in flight together
    {
        constexpr int Q4 = CBW / 4;
        constexpr int TOTAL = NCONV * 26 * Q4;
        constexpr int TB = (TOTAL + NT - 1) / NT;
        float4 t[TB];
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            int idx = tid + i * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int row = idx / Q4, part = idx % Q4;
            const int conv = row / 26, tap = row - conv * 26;
            if (tap < 25) t[i] = *reinterpret_cast<const float4*>(wpack + ((size_t)conv * 25 + tap) * a.C + c0 + part * 4);
            else if (a.has_bias) t[i] = *reinterpret_cast<const float4*>(bpack + (size_t)conv * a.C + c0 + part * 4);
            else t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        });
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const int idx = tid + i * NT;
            if (idx < TOTAL) *reinterpret_cast<float4*>(taps + (size_t)idx * 4) = t[i];
        });
    }

    for (int i = tid; i < XFLOATS; i += NT) xarea[i] = 0.f;           // the zeros are the horizontal padding; ordered by the first barrier below

    const int lane = tid & 63, wave = tid >> 6;
    Ctx c;
    c.lane_in_group = lane % LPC;
    c.mode = MODE;
    const int ch = wave * CPW + lane / LPC;
    xch_setup<LPC>(c, xarea, ch, B0);
    const bool active = c.lane_in_group < LA;
    unsigned char* mine = img + c.lane_in_group * PITCH + ch * ESZ;   // + (row * W0 + j * LA) * PITCH for column j of the lane
    const float* my_taps = taps + ch;

    float xr[2][B0];                                                  // x rows in flight (lower halves stay zero, Raw::ld_hi)
#pragma unroll
    for (int j = 0; j < B0; ++j) { xr[0][j] = 0.f; xr[1][j] = 0.f; asm volatile("" : "+v"(xr[0][j]), "+v"(xr[1][j])); }

    for (int n = n0; n < n1; ++n) {
        // ---- this image's chunks (already in registers) -> raw LDS image
        sfor<STAGE>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (have[i]) *reinterpret_cast<u32x4*>(img + l_off[i]) = v[i];
        });
        if (n == n0) RCX_LSTAMP(33);
        __syncthreads();
        if (n == n0) RCX_LSTAMP(34);
        if (n + 1 < n1) prefetch(n + 1);
        // ---- the whole block in registers; every output row overwrites the lane's own (already consumed) x bytes
        if (active && !RCX_LABLATE(a, 1)) {
            // x rows are read one row ahead into two alternating register sets (even / odd rows), so that the LDS round trip
            // of row r+1 runs under the FMAs of row r.  bf16 lands in the upper half of a register whose lower half stays zero
            // (Raw::ld_hi), which is the f32 value.  Output rows trail the input by two rows, so the bytes are still x.
            Level<LPC, MODE, 0, LEVEL, W0, B0, 1, CBW>::run_io(
                [&](auto R, float (&row)[B0]) RCX_INL {
                    constexpr int r = decltype(R)::value;
                    if constexpr (r == 0) {
                        sfor<B0>([&](auto J) RCX_INL { Raw<TIO>::template ld_hi<(decltype(J)::value * LA) * PITCH>(mine, xr[0][decltype(J)::value]); });
                    }
                    Raw<TIO>::settle(xr[r & 1]);                             // row r was requested one row ago
                    if constexpr (r + 1 < W0) {
                        sfor<B0>([&](auto J) RCX_INL {
                            Raw<TIO>::template ld_hi<((r + 1) * W0 + decltype(J)::value * LA) * PITCH>(mine, xr[(r + 1) & 1][decltype(J)::value]);
                        });
                    }
#pragma unroll
                    for (int j = 0; j < B0; ++j) row[j] = xr[r & 1][j];
                },
                [&](auto O, const float (&acc)[B0]) RCX_INL {
#pragma unroll
                    for (int j = 0; j < B0; ++j) Raw<TIO>::st(mine + (decltype(O)::value * W0 + j * LA) * PITCH, acc[j]);
                },
                my_taps, c);
        }
        if (n == n0) RCX_LSTAMP(35);
        __syncthreads();
        if (n == n0) RCX_LSTAMP(36);
        // ---- y: raw LDS image -> coalesced 16-byte stores (same thread <-> chunk mapping as the loads, so the next
        //      image's LDS writes need no barrier after these reads)
        unsigned char* yg = reinterpret_cast<unsigned char*>(y + (size_t)n * img_stride + c0);
        if (!RCX_LABLATE(a, 4)) sfor<STAGE>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (have[i]) *reinterpret_cast<u32x4*>(yg + g_off[i]) = *reinterpret_cast<const u32x4*>(img + l_off[i]);
        });
        if (n == n0) RCX_LSTAMP(37);
    }
}

// ------------------------------------------------------------------------------------------------
// Sectioned level 0 for the planes whose rows do not fit the register file at once (28x28, 56x56): x streams through a
// 3-slot LDS ring in bands of SR rows, twice (pass 1: stride-2 conv -> F_1 plane in registers; pass 2: T_0 = x + resize(C_1),
// final conv, y rows written back into the ring and stored band by band).  Levels >= 1 run in registers between the passes.
// One __syncthreads per band; the partial sums that straddle a band boundary are carried in registers; the plane P is a
// register array indexed through a uniform switch on the band number (only the taken case executes).
#ifndef RCX_BANDED_WPE
#define RCX_BANDED_WPE
#endif
template <int W0, int LEVEL, int LPC, int MODE, int NW, int SR, typename TIO>
__global__ __launch_bounds__(NW * 64) RCX_BANDED_WPE
void k_recconv_lanes_banded(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bpack, LanesArgs a)
{
    constexpr int LA = lanes_active(W0, LPC);
    constexpr int B0 = W0 / LA, H0 = W0, W1 = W0 / 2, H1 = W1, B1 = B0 / 2;
    static_assert(B0 * LA == W0 && B0 >= 2 && (B0 % 2) == 0, "plane width must be LA * B0, B0 even");
    static_assert(SR >= 4 && (SR % 2) == 0 && (H0 % SR) == 0, "band height");
    constexpr int NS = H0 / SR;                   // bands per pass
    constexpr int HS = SR / 2;                    // coarse rows per band
    constexpr int CPW = 64 / LPC, CBW = NW * CPW, NT = NW * 64, NCONV = LEVEL + 2, ESZ = (int)sizeof(TIO);
    constexpr int PITCH = CBW * ESZ + 16, CPP = CBW * ESZ / 16;
    static_assert(CPP >= 1 && (CPP & (CPP - 1)) == 0, "channel block must be a power-of-two number of 16-byte chunks");
    constexpr int BAND_PX = SR * W0, BAND_BYTES = BAND_PX * PITCH, NCHUNKS = BAND_PX * CPP;
    constexpr int STAGE = (NCHUNKS + NT - 1) / NT;
    constexpr int TAPS_BYTES = NCONV * 26 * CBW * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* taps = reinterpret_cast<float*>(smem);
    unsigned char* ring = smem + TAPS_BYTES;      // [3][BAND_PX][PITCH]
    float* xarea = reinterpret_cast<float*>(ring + 3 * BAND_BYTES);   // [CBW][xch_stride]: neighbour-exchange lines (rcx_lanes.h)
    constexpr int XFLOATS = RCX_XCH_LDS ? CBW * xch_stride(LPC, B0) : 0;

    const int tid = threadIdx.x;
    // workgroup -> (channel block, image group): the channel blocks of one image group get ids that are equal mod 8,
    // i.e. land on the same XCD (round-robin dispatch) close in time and share its L2 lines
    const int cb = (blockIdx.x % (8 * a.nblk)) / 8;
    const int n0 = ((blockIdx.x / (8 * a.nblk)) * 8 + blockIdx.x % 8) * a.ni;
    const int n1 = n0 + a.ni < a.N ? n0 + a.ni : a.N;
    const int c0 = cb * CBW;
    const size_t img_stride = (size_t)H0 * W0 * a.C;
    const int band_stride = BAND_PX * a.C * ESZ;  // bytes between bands in global memory

    int g_off[STAGE], l_off[STAGE];
    bool have[STAGE];
    sfor<STAGE>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        have[i] = (i + 1) * NT <= NCHUNKS || cidx < NCHUNKS;
        cidx = have[i] ? cidx : NCHUNKS - 1;
        const int p = cidx / CPP, part = cidx % CPP;
        g_off[i] = p * a.C * ESZ + part * 16;
        l_off[i] = lds_slot<W0, B0, LA>(p) * PITCH + part * 16;
    });
    u32x4 v[STAGE];
    auto prefetch = [&](int n, int band) RCX_INL {
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(x + (size_t)n * img_stride + c0) + (size_t)band * band_stride;
        sfor<STAGE>([&](auto I) RCX_INL { v[decltype(I)::value] = *reinterpret_cast<const u32x4*>(xg + g_off[decltype(I)::value]); });
    };
    auto stage_in = [&](unsigned char* slot) RCX_INL {
        sfor<STAGE>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (have[i]) *reinterpret_cast<u32x4*>(slot + l_off[i]) = v[i];
        });
    };
    // a finished y band leaves in two halves so that neither round trip is waited for: LDS -> registers right after the
    // barrier, registers -> global memory after the band's arithmetic
    u32x4 yv[STAGE];
    auto lift_band = [&](const unsigned char* slot) RCX_INL {
        sfor<STAGE>([&](auto I) RCX_INL { yv[decltype(I)::value] = *reinterpret_cast<const u32x4*>(slot + l_off[decltype(I)::value]); });
    };
    auto drop_band = [&](int n, int band) RCX_INL {
        unsigned char* yg = reinterpret_cast<unsigned char*>(y + (size_t)n * img_stride + c0) + (size_t)band * band_stride;
        sfor<STAGE>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (have[i]) *reinterpret_cast<u32x4*>(yg + g_off[i]) = yv[i];
        });
    };
    auto store_band = [&](int n, int band, const unsigned char* slot) RCX_INL {
        unsigned char* yg = reinterpret_cast<unsigned char*>(y + (size_t)n * img_stride + c0) + (size_t)band * band_stride;
        sfor<STAGE>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (have[i]) *reinterpret_cast<u32x4*>(yg + g_off[i]) = *reinterpret_cast<const u32x4*>(slot + l_off[i]);
        });
    };
    if (n0 < n1) prefetch(n0, 0);

    {   // taps + bias rows -> LDS
        constexpr int Q4 = CBW / 4, TOTAL = NCONV * 26 * Q4, TB = (TOTAL + NT - 1) / NT;
        float4 t[TB];
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            int idx = tid + i * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int row = idx / Q4, part = idx % Q4;
            const int conv = row / 26, tap = row - conv * 26;
            if (tap < 25) t[i] = *reinterpret_cast<const float4*>(wpack + ((size_t)conv * 25 + tap) * a.C + c0 + part * 4);
            else if (a.has_bias) t[i] = *reinterpret_cast<const float4*>(bpack + (size_t)conv * a.C + c0 + part * 4);
            else t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        });
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const int idx = tid + i * NT;
            if (idx < TOTAL) *reinterpret_cast<float4*>(taps + (size_t)idx * 4) = t[i];
        });
    }

    for (int i = tid; i < XFLOATS; i += NT) xarea[i] = 0.f;           // the zeros are the horizontal padding; ordered by the barrier below

    const int lane = tid & 63, wave = tid >> 6;
    Ctx c;
    c.lane_in_group = lane % LPC;
    c.mode = MODE;
    const int ch = wave * CPW + lane / LPC;
    xch_setup<LPC>(c, xarea, ch, B0);
    const bool active = c.lane_in_group < LA;
    const int mine = c.lane_in_group * PITCH + ch * ESZ;              // + (row * W0 + j * LA) * PITCH for column j of the lane
    const float* my_taps = taps + ch;
    constexpr VT te = vtab(MODE, H1, H0, 2), to = vtab(MODE, H1, H0, 3);   // vertical weights of an interior even / odd row
    // packed FMAs in pass 2 only where the register pairs still leave two waves per SIMD (float32 I/O stages twice the bytes)
    constexpr bool PK2 = RCX_PK_FMA && sizeof(TIO) == 2;

    RCX_LSTAMP(0);
    __syncthreads();                               // taps are read (pass 1 keeps them in registers) before the first band barrier
    int slot = 0;
    auto slot_ptr = [&](int k) RCX_INL { return ring + ((slot + k) % 3) * BAND_BYTES; };

    for (int n = n0; n < n1; ++n) {
        float P[H1][B1];                                   // F_1, later C_1
        // ================= pass 1: F_1 = down(x) =================
        {
            float A[2][B1];
            Taps w;
            if (active) load_taps<CBW>(my_taps, w);                  // stays in registers for the whole pass
            const float bias = w.bias();
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                unsigned char* cur = slot_ptr(0);
                stage_in(cur);
                if (n == n0 && s < 4) RCX_LSTAMP(8 + 3 * s);
                __syncthreads();
                if (n == n0 && s < 4) RCX_LSTAMP(9 + 3 * s);
                prefetch(n, s + 1 < NS ? s + 1 : 0);
                if (active) {
                    float L[HS + 2][B1];
#pragma unroll
                    for (int cidx = 0; cidx < B1; ++cidx) { L[0][cidx] = s == 0 ? bias : A[0][cidx]; L[1][cidx] = s == 0 ? bias : A[1][cidx]; }
                    const unsigned char* xb = cur + mine;
                    float nxt[B0];                                      // next row's x, loaded one row ahead of its use
#pragma unroll
                    for (int j = 0; j < B0; ++j) nxt[j] = Raw<TIO>::ld(xb + (j * LA) * PITCH);
                    sfor<SR>([&](auto I) RCX_INL {
                        constexpr int i = decltype(I)::value;
                        float row[B0], ext[B0 + 4];
#pragma unroll
                        for (int j = 0; j < B0; ++j) row[j] = nxt[j];
                        if constexpr (i + 1 < SR) {
#pragma unroll
                            for (int j = 0; j < B0; ++j) nxt[j] = Raw<TIO>::ld(xb + ((i + 1) * W0 + j * LA) * PITCH);
                        }
                        make_ext<LPC, B0, 1>(row, ext, c);
                        sfor<5>([&](auto U) RCX_INL {
                            constexpr int u = decltype(U)::value;
                            constexpr int t = i + 2 - u;
                            if constexpr (((t % 2) + 2) % 2 == 0) {
                                constexpr int orel = (t + 2) / 2 - 1;                   // floor(t / 2), t >= -2
                                constexpr bool is_first = orel >= 1 && u == 0;           // rows that start inside this band
#pragma unroll
                                for (int q = 0; q < B1; ++q) {
                                    float acc = is_first ? bias : L[orel + 1][q];
#pragma unroll
                                    for (int vv = 0; vv < 5; ++vv) acc = fmaf(ext[2 * q + vv], w.get(u, vv), acc);
                                    L[orel + 1][q] = acc;
                                }
                            }
                        });
                        RCX_ROW_FENCE;
                    });
                    // rows HS*s - 1 .. HS*s + HS - 2 are complete
                    sfor<NS>([&](auto S) RCX_INL {
                        constexpr int sv = decltype(S)::value;
                        if (s == sv) {
                            sfor<HS>([&](auto K) RCX_INL {
                                constexpr int o = HS * sv + decltype(K)::value - 1;
                                if constexpr (o >= 0) {
#pragma unroll
                                    for (int q = 0; q < B1; ++q) P[o][q] = L[decltype(K)::value][q];
                                }
                            });
                        }
                    });
#pragma unroll
                    for (int q = 0; q < B1; ++q) { A[0][q] = L[HS][q]; A[1][q] = L[HS + 1][q]; }
                }
                if (n == n0 && s < 4) RCX_LSTAMP(10 + 3 * s);
                slot = (slot + 1) % 3;
            }
            if (active) {
#pragma unroll
                for (int q = 0; q < B1; ++q) P[H1 - 1][q] = A[0][q];
            }
        }
        // ================= levels >= 1 in registers: P <- C_1 =================
        if (n == n0) RCX_LSTAMP(1);
        float wt[B0][2];
        if (active) {
            float Q[H1][B1];
            Level<LPC, MODE, 1, LEVEL, W1, B1, 1, CBW, PK2>::run(P, Q, my_taps, c);
            sfor<H1>([&](auto R) RCX_INL {
#pragma unroll
                for (int q = 0; q < B1; ++q) P[decltype(R)::value][q] = Q[decltype(R)::value][q];
            });
            hweights_2x<B1, B0>(c, W1, W0, wt);
        }
        // ================= pass 2: y = conv_L(x + resize(C_1)) =================
        if (n == n0) RCX_LSTAMP(2);
        {
            f32x2 Cy[4][B0 / 2];
            Taps w;
            if (active) load_taps<CBW>(my_taps + (1 + LEVEL) * 26 * CBW, w);
            const float bias = w.bias();
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                unsigned char* cur = slot_ptr(0);
                unsigned char* prev = slot_ptr(2);
                stage_in(cur);
                if (n == n0 && s < 4) RCX_LSTAMP(24 + 4 * s);
                __syncthreads();
                if (n == n0 && s < 4) RCX_LSTAMP(25 + 4 * s);
                if (s + 1 < NS) prefetch(n, s + 1);
                else if (n + 1 < n1) prefetch(n + 1, 0);
                if (s >= 2) lift_band(slot_ptr(1));
                if (n == n0 && s < 4) RCX_LSTAMP(26 + 4 * s);
                if (active) {
                    // coarse rows HS*s - 1 .. HS*s + HS (clamped) out of the register plane, then resized horizontally.
                    // The empty asm keeps the resize inside the band loop (hoisted, it would hold every band's rows live).
                    float cw[HS + 2][B1], hw[HS + 2][B0];
                    sfor<NS>([&](auto S) RCX_INL {
                        constexpr int sv = decltype(S)::value;
                        if (s == sv) {
                            sfor<HS + 2>([&](auto K) RCX_INL {
                                constexpr int raw = HS * sv - 1 + decltype(K)::value;
                                constexpr int cr = raw < 0 ? 0 : (raw > H1 - 1 ? H1 - 1 : raw);
#pragma unroll
                                for (int q = 0; q < B1; ++q) cw[decltype(K)::value][q] = P[cr][q];
                            });
                        }
                    });
                    sfor<HS + 2>([&](auto K) RCX_INL {
#pragma unroll
                        for (int q = 0; q < B1; ++q) asm volatile("" : "+v"(cw[decltype(K)::value][q]));
                        hresize_row<LPC, B1, B0>(cw[decltype(K)::value], wt, hw[decltype(K)::value], c);
                    });
                    // partial sums of the band's rows as column pairs (v_pk_fma_f32: two columns per instruction, the tap
                    // splat through op_sel; same products and order of summation as the scalar form)
                    f32x2 L[SR + 4][B0 / 2];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < B0 / 2; ++q) L[k][q] = s == 0 ? f32x2{bias, bias} : Cy[k][q];
                    unsigned char* xb = cur + mine;
                    unsigned char* pb = prev + mine;
                    float nxt[B0];                                      // next row's x, loaded one row ahead of its use
#pragma unroll
                    for (int j = 0; j < B0; ++j) nxt[j] = Raw<TIO>::ld(xb + (j * LA) * PITCH);
                    sfor<SR>([&](auto I) RCX_INL {
                        constexpr int i = decltype(I)::value;
                        constexpr int mr = i / 2;
                        float row[B0], ext[B0 + 4];
#pragma unroll
                        for (int j = 0; j < B0; ++j) {
                            const float xv = nxt[j];
                            if constexpr (MODE == 1) row[j] = xv + hw[mr + 1][j];
                            else if constexpr ((i & 1) == 0) row[j] = xv + fmaf(te.l, hw[mr + 1][j], (1.f - te.l) * hw[mr][j]);
                            else row[j] = xv + fmaf(to.l, hw[mr + 2][j], (1.f - to.l) * hw[mr + 1][j]);
                        }
                        if constexpr (i + 1 < SR) {
#pragma unroll
                            for (int j = 0; j < B0; ++j) nxt[j] = Raw<TIO>::ld(xb + ((i + 1) * W0 + j * LA) * PITCH);
                        }
                        make_ext<LPC, B0, 1>(row, ext, c);
                        // the row as aligned pairs E[k] = (ext[2k], ext[2k+1]) and the odd ones O[k] = (ext[2k+1], ext[2k+2])
                        f32x2 E[B0 / 2 + 2], O[B0 / 2 + 1];
                        if constexpr (PK2) {
#pragma unroll
                            for (int k = 0; k < B0 / 2 + 2; ++k) E[k] = f32x2{ext[2 * k], ext[2 * k + 1]};
#pragma unroll
                            for (int k = 0; k < B0 / 2 + 1; ++k) O[k] = f32x2{ext[2 * k + 1], ext[2 * k + 2]};
                        }
                        sfor<5>([&](auto U) RCX_INL {
                            constexpr int u = decltype(U)::value;
                            constexpr int idx = i + 2 - u + 2;                           // output row (i + 2 - u) relative to the band, + 2
#pragma unroll
                            for (int q = 0; q < B0 / 2; ++q) {
                                f32x2 acc = u == 0 ? f32x2{bias, bias} : L[idx][q];
                                if constexpr (PK2) {
                                    acc = __builtin_elementwise_fma(E[q], w.splat(u, 0), acc);
                                    acc = __builtin_elementwise_fma(O[q], w.splat(u, 1), acc);
                                    acc = __builtin_elementwise_fma(E[q + 1], w.splat(u, 2), acc);
                                    acc = __builtin_elementwise_fma(O[q + 1], w.splat(u, 3), acc);
                                    acc = __builtin_elementwise_fma(E[q + 2], w.splat(u, 4), acc);
                                } else {
#pragma unroll
                                    for (int vv = 0; vv < 5; ++vv) {
                                        acc.x = fmaf(ext[2 * q + vv], w.get(u, vv), acc.x);
                                        acc.y = fmaf(ext[2 * q + vv + 1], w.get(u, vv), acc.y);
                                    }
                                }
                                L[idx][q] = acc;
                            }
                        });
                        // output row i - 2 (relative) is complete
                        if constexpr (i < 2) {
                            if (s > 0) {
#pragma unroll
                                for (int j = 0; j < B0; ++j) Raw<TIO>::st(pb + ((SR + i - 2) * W0 + j * LA) * PITCH, (j & 1) ? L[i][j / 2].y : L[i][j / 2].x);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < B0; ++j) Raw<TIO>::st(xb + ((i - 2) * W0 + j * LA) * PITCH, (j & 1) ? L[i][j / 2].y : L[i][j / 2].x);
                        }
                        RCX_ROW_FENCE;
                    });
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < B0 / 2; ++q) Cy[k][q] = L[SR + k][q];
                }
                if (s >= 2) drop_band(n, s - 2);
                if (n == n0 && s < 4) RCX_LSTAMP(27 + 4 * s);
                slot = (slot + 1) % 3;
            }
            // rows H0-2, H0-1 into the last band's slot
            if (active) {
                unsigned char* pb = slot_ptr(2) + mine;
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < B0; ++j) Raw<TIO>::st(pb + ((SR - 2 + k) * W0 + j * LA) * PITCH, (j & 1) ? Cy[k][j / 2].y : Cy[k][j / 2].x);
            }
            __syncthreads();
            if (NS >= 2) store_band(n, NS - 2, slot_ptr(1));
            store_band(n, NS - 1, slot_ptr(2));
            if (n == n0) RCX_LSTAMP(3);
        }
    }
}

struct LanesPlan {
    bool ok;
    bool banded;
    int w0, level, lpc, waves, sr;
    size_t lds;
    LanesArgs args;
};

static inline int env_int(rcx::opt::Id id, int dflt)
{
    const char* v = rcx::opt::value(id);
    return v && *v ? atoi(v) : dflt;
}

static inline LanesPlan plan(int N, int C, int H, int W, int level, int k, int dtype)
{
    LanesPlan p{};
    if (env_int(rcx::opt::LANES, 1) == 0) return p;
    if (k != 5 || H != W) return p;
    if (dtype > 1) return p;                       // float16 I/O: the channel-per-lane kernels and the generic schedule (rcx_api.hip)
    int natural = -1, lpc = 8;
    if (W == 7) natural = 1;
    else if (W == 14) natural = 2;
    else if (W == 28) natural = 3;
    else if (W == 56) { natural = 4; lpc = 16; }
    else if (W == 16) { natural = 1; lpc = 16; }          // 16 * 2^k planes: all 16 lanes of a group are active, the DPP row
    else if (W == 32) { natural = 2; lpc = 16; }          // boundary itself is the zero padding
    else if (W == 64) { natural = 3; lpc = 16; }
    if (natural < 0 || level != natural) return p;
    const int esz = dtype == 1 ? 2 : 4;
    const int cpw = 64 / lpc;
    const bool banded = W >= 28 && W != 16;
    const int sr = 4;
    int waves = env_int(rcx::opt::LANES_WAVES, W == 7 || lpc == 16 ? 8 : 4);
    if (waves != 8 && waves != 4 && waves != 2 && waves != 1) waves = 8;
    while (waves > 1 && C % (waves * cpw) != 0) waves >>= 1;
    if (W % 16 == 0 && waves < 4) return p;                  // the 16-family is instantiated for 8 and 4 waves only
    const int cbw = waves * cpw;
    if (C % cbw != 0 || (cbw * esz) % 16 != 0) return p;
    p.lds = (size_t)(level + 2) * 26 * cbw * 4 + (size_t)(banded ? 3 * sr : H) * W * (cbw * esz + 16);
    if (RCX_XCH_LDS) p.lds += (size_t)cbw * xch_stride(lpc, W / lanes_active(W, lpc)) * 4;
    if (p.lds > 160 * 1024) return p;
    p.banded = banded; p.sr = sr;
    p.w0 = W; p.level = level; p.lpc = lpc; p.waves = waves;
    p.args.N = N; p.args.C = C; p.args.nblk = C / cbw;
    // enough workgroups to fill 256 CUs a few times over, the rest of the batch looped inside (taps staged once)
    int ni = env_int(rcx::opt::LANES_NI, 0);
    if (ni <= 0) {
        ni = 1;
        while ((long)p.args.nblk * ((N + 2 * ni - 1) / (2 * ni)) >= 2048 && ni < 4) ni *= 2;
    }
    p.args.ni = ni;
    p.ok = true;
    return p;
}

template <int W0, int LEVEL, int LPC, int MODE, int NW, typename TIO>
static hipError_t launch_w(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    auto kfn = k_recconv_lanes<W0, LEVEL, LPC, MODE, NW, TIO>;
    RCX_SET_LDS_ONCE(kfn, p.lds);
    LanesArgs a = p.args;
    a.has_bias = bpack != nullptr;
    a.ablate = env_int(rcx::opt::LANES_ABLATE, 0);
    const unsigned grid = (unsigned)(a.nblk * (((a.N + a.ni - 1) / a.ni + 7) / 8 * 8));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), p.lds, s, (const TIO*)x, (TIO*)y, wpack, bpack, a);
    return hipGetLastError();
}

template <int W0, int LEVEL, int LPC, int MODE, int NW, int SR, typename TIO>
static hipError_t launch_bw(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    auto kfn = k_recconv_lanes_banded<W0, LEVEL, LPC, MODE, NW, SR, TIO>;
    RCX_SET_LDS_ONCE(kfn, p.lds);
    LanesArgs a = p.args;
    a.has_bias = bpack != nullptr;
    const unsigned grid = (unsigned)(a.nblk * (((a.N + a.ni - 1) / a.ni + 7) / 8 * 8));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), p.lds, s, (const TIO*)x, (TIO*)y, wpack, bpack, a);
    return hipGetLastError();
}

template <int W0, int LEVEL, int LPC, int MODE, int SR, typename TIO>
static hipError_t launch_b(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    switch (p.waves) {
    case 8: return launch_bw<W0, LEVEL, LPC, MODE, 8, SR, TIO>(x, y, wpack, bpack, p, s);
    case 4: return launch_bw<W0, LEVEL, LPC, MODE, 4, SR, TIO>(x, y, wpack, bpack, p, s);
    case 2:
        if constexpr (sizeof(TIO) * 2 * (64 / LPC) >= 16) return launch_bw<W0, LEVEL, LPC, MODE, 2, SR, TIO>(x, y, wpack, bpack, p, s);
        return hipErrorInvalidConfiguration;
    default:
        if constexpr (sizeof(TIO) * (64 / LPC) >= 16) return launch_bw<W0, LEVEL, LPC, MODE, 1, SR, TIO>(x, y, wpack, bpack, p, s);
        return hipErrorInvalidConfiguration;
    }
}

template <int W0, int LEVEL, int LPC, int MODE, typename TIO>
static hipError_t launch_t(const void* x, void* y, const float* wpack, const float* bpack, const LanesPlan& p, hipStream_t s)
{
    switch (p.waves) {
    case 8: return launch_w<W0, LEVEL, LPC, MODE, 8, TIO>(x, y, wpack, bpack, p, s);
    case 4: return launch_w<W0, LEVEL, LPC, MODE, 4, TIO>(x, y, wpack, bpack, p, s);
    case 2: return launch_w<W0, LEVEL, LPC, MODE, 2, TIO>(x, y, wpack, bpack, p, s);
    default:
        if constexpr (sizeof(TIO) * (64 / LPC) >= 16) return launch_w<W0, LEVEL, LPC, MODE, 1, TIO>(x, y, wpack, bpack, p, s);
        return hipErrorInvalidConfiguration;
    }
}

}  // namespace lanes
}  // namespace rcx
