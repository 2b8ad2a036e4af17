// Register-resident single-step kernels on the 7 * 2^k planes (lane layout of rcx_lanes.h):
//
//   k_upadd_lanes:  y = dwconv5(x + resize(coarse -> (H, W), mode))   -- the tail of RecAttn2d.forward (model/recattn.py:67) and
//                   one up-recursion step of RecConv2d (model/recnext.py:33-34); coarse is exactly half the size.
//   k_down5_lanes:  y = dwconv5_stride2(x)                            -- RecAttn2d.down[0] (model/recattn.py:61) and one rung of
//                   RecConv2d's ladder (model/recnext.py:28).
//
// Both stream x through an LDS ring in 4-row bands (coalesced 16-byte global accesses, one barrier per band) and keep the
// five live window rows in registers; the up-add kernel also rings the four coarse rows a band needs.
#include "rcx_lanes.h"
#include "rcx_opts.h"
#include "rcx_launch.h"

namespace rcx {
namespace lanes {

struct StepArgs {
    int N, C;
    int nblk;          // channel blocks per image
    int ni;            // images per workgroup
    int has_bias;
};

// ---------------------------------------------------------------------------------------------------------------------
template <int W0, int LPC, int MODE, int NW, bool HAS_COARSE, typename TX, typename TC>
__global__ __launch_bounds__(NW * 64)
void k_upadd_lanes(const TX* __restrict__ x, const TC* __restrict__ coarse, TX* __restrict__ y,
                   const float* __restrict__ wk, const float* __restrict__ bk, StepArgs a)
{
    constexpr int LA = lanes_active(W0, LPC);
    constexpr int B0 = W0 / LA, H0 = W0, W1 = W0 / 2, H1 = W1, B1 = B0 / 2;
    static_assert(B0 * LA == W0 && B0 >= 2 && (B0 % 2) == 0, "plane width must be LA * B0, B0 even");
    constexpr int SR = 4, NS = (H0 + SR - 1) / SR, HS = SR / 2;
    constexpr int CPW = 64 / LPC, CBW = NW * CPW, NT = NW * 64, XE = (int)sizeof(TX), CE = (int)sizeof(TC);
    constexpr int XPITCH = CBW * XE + 16, CPITCH = CBW * CE + 16, XCPP = CBW * XE / 16, CCPP = CBW * CE / 16;
    static_assert(XCPP >= 1 && (XCPP & (XCPP - 1)) == 0, "channel block must be a power-of-two number of 16-byte chunks");
    constexpr int XBAND = SR * W0 * XPITCH, CBAND = (HS + 2) * W1 * CPITCH;
    constexpr int XN = SR * W0 * XCPP, CN = (HS + 2) * W1 * CCPP;
    constexpr int XST = (XN + NT - 1) / NT, CST = (CN + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* taps = reinterpret_cast<float*>(smem);                 // [26][CBW]
    unsigned char* xring = smem + 26 * CBW * 4;                    // [3][XBAND]
    unsigned char* cring = xring + 3 * XBAND;                      // [2][CBAND]

    const int tid = threadIdx.x;
    const int cb = (blockIdx.x % (8 * a.nblk)) / 8;
    const int n0 = ((blockIdx.x / (8 * a.nblk)) * 8 + blockIdx.x % 8) * a.ni;
    const int n1 = n0 + a.ni < a.N ? n0 + a.ni : a.N;
    const int c0 = cb * CBW;
    const size_t ximg = (size_t)H0 * W0 * a.C, cimg = (size_t)H1 * W1 * a.C;

    int xg[XST], xl[XST], xrow[XST], cg[CST], cl[CST], crow[CST];
    bool xhave[XST], chave[CST];
    sfor<XST>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        xhave[i] = (i + 1) * NT <= XN || cidx < XN;
        cidx = xhave[i] ? cidx : XN - 1;
        const int p = cidx / XCPP, part = cidx % XCPP;
        xg[i] = p * a.C * XE + part * 16;
        xrow[i] = p / W0;
        xl[i] = lds_slot<W0, B0, LA>(p) * XPITCH + part * 16;
    });
    sfor<CST>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        chave[i] = (i + 1) * NT <= CN || cidx < CN;
        cidx = chave[i] ? cidx : CN - 1;
        const int p = cidx / CCPP, part = cidx % CCPP;
        crow[i] = p / W1;
        cg[i] = (p % W1) * a.C * CE + part * 16;                  // + coarse row * W1 * C * CE, clamped per band
        cl[i] = lds_slot<W1, B1, LA>(p) * CPITCH + part * 16;
    });
    u32x4 xv[XST], cv[CST], yv[XST];
    auto prefetch = [&](int n, int band) RCX_INL {
        const unsigned char* xp = reinterpret_cast<const unsigned char*>(x + (size_t)n * ximg + c0) + (size_t)band * (SR * W0) * a.C * XE;
        sfor<XST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if ((H0 % SR) == 0 || band * SR + xrow[i] < H0) xv[i] = *reinterpret_cast<const u32x4*>(xp + xg[i]);
            else xv[i] = u32x4{0u, 0u, 0u, 0u};
        });
        const unsigned char* cp = reinterpret_cast<const unsigned char*>(coarse + (size_t)n * cimg + c0);
        if constexpr (HAS_COARSE) sfor<CST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            int r = HS * band - 1 + crow[i];                        // coarse rows HS*band - 1 .. HS*band + HS, clamped
            r = r < 0 ? 0 : (r > H1 - 1 ? H1 - 1 : r);
            cv[i] = *reinterpret_cast<const u32x4*>(cp + (size_t)r * W1 * a.C * CE + cg[i]);
        });
    };
    auto stage_in = [&](unsigned char* xs, unsigned char* cs) RCX_INL {
        sfor<XST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (xhave[i]) *reinterpret_cast<u32x4*>(xs + xl[i]) = xv[i];
        });
        if constexpr (HAS_COARSE) sfor<CST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (chave[i]) *reinterpret_cast<u32x4*>(cs + cl[i]) = cv[i];
        });
    };
    auto lift_band = [&](const unsigned char* slot) RCX_INL {
        sfor<XST>([&](auto I) RCX_INL { yv[decltype(I)::value] = *reinterpret_cast<const u32x4*>(slot + xl[decltype(I)::value]); });
    };
    auto drop_band = [&](int n, int band) RCX_INL {
        unsigned char* yp = reinterpret_cast<unsigned char*>(y + (size_t)n * ximg + c0) + (size_t)band * (SR * W0) * a.C * XE;
        sfor<XST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (xhave[i] && ((H0 % SR) == 0 || band * SR + xrow[i] < H0)) *reinterpret_cast<u32x4*>(yp + xg[i]) = yv[i];
        });
    };
    if (n0 < n1) prefetch(n0, 0);

    {   // taps + bias row -> LDS
        constexpr int Q4 = CBW / 4, TOTAL = 26 * Q4, TB = (TOTAL + NT - 1) / NT;
        float4 t[TB];
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            int idx = tid + i * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int row = idx / Q4, part = idx % Q4;
            if (row < 25) t[i] = *reinterpret_cast<const float4*>(wk + (size_t)row * a.C + c0 + part * 4);
            else if (a.has_bias) t[i] = *reinterpret_cast<const float4*>(bk + c0 + part * 4);
            else t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        });
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const int idx = tid + i * NT;
            if (idx < TOTAL) *reinterpret_cast<float4*>(taps + (size_t)idx * 4) = t[i];
        });
    }

    const int lane = tid & 63, wave = tid >> 6;
    Ctx c;
    c.lane_in_group = lane % LPC;
    c.mode = MODE;
    const int ch = wave * CPW + lane / LPC;
    const bool active = c.lane_in_group < LA;
    const int xmine = c.lane_in_group * XPITCH + ch * XE;
    const int cmine = c.lane_in_group * CPITCH + ch * CE;
    constexpr VT te = vtab(MODE, H1, H0, 2), to = vtab(MODE, H1, H0, 3);
    __syncthreads();

    int slot = 0;
    for (int n = n0; n < n1; ++n) {
        float Cy[4][B0];
        float w[25], bias = 0.f, wt[B0][2];
        if (active) {
            load_taps<CBW>(taps + ch, w, bias);
            if constexpr (HAS_COARSE) hweights_2x<B1, B0>(c, W1, W0, wt);
        }
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            unsigned char* cur = xring + (slot % 3) * XBAND;
            unsigned char* prev = xring + ((slot + 2) % 3) * XBAND;
            unsigned char* ccur = cring + (s & 1) * CBAND;
            stage_in(cur, ccur);
            __syncthreads();
            if (s + 1 < NS) prefetch(n, s + 1);
            else if (n + 1 < n1) prefetch(n + 1, 0);
            if (s >= 2) lift_band(xring + ((slot + 1) % 3) * XBAND);
            if (active) {
                float hw[HS + 2][B0];
                const unsigned char* cbp = ccur + cmine;
                if constexpr (HAS_COARSE) sfor<HS + 2>([&](auto K) RCX_INL {
                    float cw[B1];
#pragma unroll
                    for (int q = 0; q < B1; ++q) cw[q] = Raw<TC>::ld(cbp + (decltype(K)::value * W1 + q * LA) * CPITCH);
                    hresize_row<LPC, B1, B0>(cw, wt, hw[decltype(K)::value]);
                });
                float L[SR + 4][B0];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < B0; ++j) L[k][j] = s == 0 ? bias : Cy[k][j];
                unsigned char* xb = cur + xmine;
                unsigned char* pb = prev + xmine;
                float nxt[B0];
#pragma unroll
                for (int j = 0; j < B0; ++j) nxt[j] = Raw<TX>::ld(xb + (j * LA) * XPITCH);
                sfor<SR>([&](auto I) RCX_INL {
                    constexpr int i = decltype(I)::value;
                    constexpr int mr = i / 2;
                    float row[B0], ext[B0 + 4];
#pragma unroll
                    for (int j = 0; j < B0; ++j) {
                        const float xval = nxt[j];
                        if constexpr (!HAS_COARSE) row[j] = xval;
                        else if constexpr (MODE == 1) row[j] = xval + hw[mr + 1][j];
                        else if constexpr ((i & 1) == 0) row[j] = xval + fmaf(te.l, hw[mr + 1][j], (1.f - te.l) * hw[mr][j]);
                        else row[j] = xval + fmaf(to.l, hw[mr + 2][j], (1.f - to.l) * hw[mr + 1][j]);
                    }
                    if constexpr ((H0 % SR) != 0) {                    // rows past the plane (14 = 3.5 bands) must be zero, not resize(...)
                        if (s * SR + i >= H0) {
#pragma unroll
                            for (int j = 0; j < B0; ++j) row[j] = 0.f;
                        }
                    }
                    if constexpr (i + 1 < SR) {
#pragma unroll
                        for (int j = 0; j < B0; ++j) nxt[j] = Raw<TX>::ld(xb + ((i + 1) * W0 + j * LA) * XPITCH);
                    }
                    make_ext<LPC, B0, 1>(row, ext);
                    sfor<5>([&](auto U) RCX_INL {
                        constexpr int u = decltype(U)::value;
                        constexpr int idx = i + 2 - u + 2;
                        if constexpr (RCX_PK_FMA && sizeof(TX) == 2) {           // two columns per v_pk_fma_f32 (rcx_lanes.h, conv5_s1)
#pragma unroll
                            for (int q = 0; q < B0 / 2; ++q) {
                                f32x2 acc = u == 0 ? f32x2{bias, bias} : f32x2{L[idx][2 * q], L[idx][2 * q + 1]};
#pragma unroll
                                for (int vv = 0; vv < 5; ++vv)
                                    acc = __builtin_elementwise_fma(f32x2{ext[2 * q + vv], ext[2 * q + vv + 1]}, f32x2{w[u * 5 + vv], w[u * 5 + vv]}, acc);
                                L[idx][2 * q] = acc.x;
                                L[idx][2 * q + 1] = acc.y;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < B0; ++j) {
                                float acc = u == 0 ? bias : L[idx][j];
#pragma unroll
                                for (int vv = 0; vv < 5; ++vv) acc = fmaf(ext[j + vv], w[u * 5 + vv], acc);
                                L[idx][j] = acc;
                            }
                        }
                    });
                    if constexpr (i < 2) {
                        if (s > 0) {
#pragma unroll
                            for (int j = 0; j < B0; ++j) Raw<TX>::st(pb + ((SR + i - 2) * W0 + j * LA) * XPITCH, L[i][j]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < B0; ++j) Raw<TX>::st(xb + ((i - 2) * W0 + j * LA) * XPITCH, L[i][j]);
                    }
                    RCX_ROW_FENCE;
                });
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < B0; ++j) Cy[k][j] = L[SR + k][j];
            }
            if (s >= 2) drop_band(n, s - 2);
            slot = (slot + 1) % 3;
        }
        // rows 4*NS-2, 4*NS-1 (the last two of the padded plane) into the last band's slot
        if (active) {
            unsigned char* pb = xring + ((slot + 2) % 3) * XBAND + xmine;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < B0; ++j) Raw<TX>::st(pb + ((SR - 2 + k) * W0 + j * LA) * XPITCH, Cy[k][j]);
        }
        __syncthreads();
        if (NS >= 2) { lift_band(xring + ((slot + 1) % 3) * XBAND); drop_band(n, NS - 2); }
        lift_band(xring + ((slot + 2) % 3) * XBAND);
        drop_band(n, NS - 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// y = dwconv5_stride2(x): (H0, W0) -> (H0/2, W0/2); TO may differ from TX (the training forward keeps F_l in float32)
template <int W0, int LPC, int NW, typename TX, typename TO>
__global__ __launch_bounds__(NW * 64)
void k_down5_lanes(const TX* __restrict__ x, TO* __restrict__ y, const float* __restrict__ wk, const float* __restrict__ bk, StepArgs a)
{
    constexpr int LA = lanes_active(W0, LPC);
    constexpr int B0 = W0 / LA, BO = B0 / 2, H0 = W0, W1 = W0 / 2, H1 = W1;
    static_assert(B0 * LA == W0 && B0 >= 2 && (B0 % 2) == 0, "plane width must be LA * B0, B0 even");
    constexpr int SR = 4, NS = (H0 + SR - 1) / SR, HS = SR / 2;
    constexpr int CPW = 64 / LPC, CBW = NW * CPW, NT = NW * 64, XE = (int)sizeof(TX), OE = (int)sizeof(TO);
    constexpr int XPITCH = CBW * XE + 16, OPITCH = CBW * OE + 16, XCPP = CBW * XE / 16, OCPP = CBW * OE / 16;
    static_assert(XCPP >= 1 && (XCPP & (XCPP - 1)) == 0, "channel block must be a power-of-two number of 16-byte chunks");
    constexpr int XBAND = SR * W0 * XPITCH, OBAND = HS * W1 * OPITCH;
    constexpr int XN = SR * W0 * XCPP, ON = HS * W1 * OCPP;
    constexpr int XST = (XN + NT - 1) / NT, OST = (ON + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* taps = reinterpret_cast<float*>(smem);
    unsigned char* xring = smem + 26 * CBW * 4;                    // [2][XBAND]
    unsigned char* oring = xring + 2 * XBAND;                      // [2][OBAND]: output rows 2s-1, 2s of band s

    const int tid = threadIdx.x;
    const int cb = (blockIdx.x % (8 * a.nblk)) / 8;
    const int n0 = ((blockIdx.x / (8 * a.nblk)) * 8 + blockIdx.x % 8) * a.ni;
    const int n1 = n0 + a.ni < a.N ? n0 + a.ni : a.N;
    const int c0 = cb * CBW;
    const size_t ximg = (size_t)H0 * W0 * a.C, yimg = (size_t)H1 * W1 * a.C;

    int xg[XST], xl[XST], xrow[XST], og[OST], ol[OST], opix[OST];
    bool xhave[XST], ohave[OST];
    sfor<XST>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        xhave[i] = (i + 1) * NT <= XN || cidx < XN;
        cidx = xhave[i] ? cidx : XN - 1;
        const int p = cidx / XCPP, part = cidx % XCPP;
        xg[i] = p * a.C * XE + part * 16;
        xrow[i] = p / W0;
        xl[i] = lds_slot<W0, B0, LA>(p) * XPITCH + part * 16;
    });
    sfor<OST>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        ohave[i] = (i + 1) * NT <= ON || cidx < ON;
        cidx = ohave[i] ? cidx : ON - 1;
        const int p = cidx / OCPP, part = cidx % OCPP;
        opix[i] = p;
        og[i] = p * a.C * OE + part * 16;
        ol[i] = lds_slot<W1, BO, LA>(p) * OPITCH + part * 16;
    });
    u32x4 xv[XST], yv[OST];
    auto prefetch = [&](int n, int band) RCX_INL {
        const unsigned char* xp = reinterpret_cast<const unsigned char*>(x + (size_t)n * ximg + c0) + (size_t)band * (SR * W0) * a.C * XE;
        sfor<XST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if ((H0 % SR) == 0 || band * SR + xrow[i] < H0) xv[i] = *reinterpret_cast<const u32x4*>(xp + xg[i]);
            else xv[i] = u32x4{0u, 0u, 0u, 0u};
        });
    };
    auto stage_in = [&](unsigned char* slot) RCX_INL {
        sfor<XST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            if (xhave[i]) *reinterpret_cast<u32x4*>(slot + xl[i]) = xv[i];
        });
    };
    // output rows leave in pairs: pair s = rows (2s - 1, 2s), s = 0 .. NS
    auto lift_pair = [&](const unsigned char* slot) RCX_INL {
        sfor<OST>([&](auto I) RCX_INL { yv[decltype(I)::value] = *reinterpret_cast<const u32x4*>(slot + ol[decltype(I)::value]); });
    };
    auto drop_pair = [&](int n, int s) RCX_INL {
        unsigned char* yp = reinterpret_cast<unsigned char*>(y + (size_t)n * yimg + c0) + ((ptrdiff_t)(2 * s - 1) * W1) * a.C * OE;
        sfor<OST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const bool row_ok = opix[i] < W1 ? (s > 0 && 2 * s - 1 < H1) : 2 * s < H1;
            if (ohave[i] && row_ok) *reinterpret_cast<u32x4*>(yp + og[i]) = yv[i];
        });
    };
    if (n0 < n1) prefetch(n0, 0);
    {
        constexpr int Q4 = CBW / 4, TOTAL = 26 * Q4, TB = (TOTAL + NT - 1) / NT;
        float4 t[TB];
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            int idx = tid + i * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int row = idx / Q4, part = idx % Q4;
            if (row < 25) t[i] = *reinterpret_cast<const float4*>(wk + (size_t)row * a.C + c0 + part * 4);
            else if (a.has_bias) t[i] = *reinterpret_cast<const float4*>(bk + c0 + part * 4);
            else t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        });
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const int idx = tid + i * NT;
            if (idx < TOTAL) *reinterpret_cast<float4*>(taps + (size_t)idx * 4) = t[i];
        });
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int lig = lane % LPC;
    const int ch = wave * CPW + lane / LPC;
    const bool active = lig < LA;
    const int xmine = lig * XPITCH + ch * XE;
    const int omine = lig * OPITCH + ch * OE;
    __syncthreads();

    for (int n = n0; n < n1; ++n) {
        float A[2][BO];
        float w[25], bias = 0.f;
        if (active) load_taps<CBW>(taps + ch, w, bias);
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            unsigned char* xs = xring + (s & 1) * XBAND;
            unsigned char* os = oring + (s & 1) * OBAND;
            stage_in(xs);
            __syncthreads();
            if (s + 1 < NS) prefetch(n, s + 1);
            else if (n + 1 < n1) prefetch(n + 1, 0);
            if (s >= 1) lift_pair(oring + ((s - 1) & 1) * OBAND);
            if (active) {
                float L[HS + 2][BO];                                   // output rows 2s-1 .. 2s+2
#pragma unroll
                for (int q = 0; q < BO; ++q) { L[0][q] = s == 0 ? bias : A[0][q]; L[1][q] = s == 0 ? bias : A[1][q]; }
                const unsigned char* xb = xs + xmine;
                unsigned char* ob = os + omine;
                float nxt[B0];
#pragma unroll
                for (int j = 0; j < B0; ++j) nxt[j] = Raw<TX>::ld(xb + (j * LA) * XPITCH);
                sfor<SR>([&](auto I) RCX_INL {
                    constexpr int i = decltype(I)::value;
                    float row[B0], ext[B0 + 4];
#pragma unroll
                    for (int j = 0; j < B0; ++j) row[j] = nxt[j];
                    if constexpr (i + 1 < SR) {
#pragma unroll
                        for (int j = 0; j < B0; ++j) nxt[j] = Raw<TX>::ld(xb + ((i + 1) * W0 + j * LA) * XPITCH);
                    }
                    make_ext<LPC, B0, 1>(row, ext);
                    sfor<5>([&](auto U) RCX_INL {
                        constexpr int u = decltype(U)::value;
                        constexpr int t = i + 2 - u;
                        if constexpr (((t % 2) + 2) % 2 == 0) {
                            constexpr int orel = (t + 2) / 2 - 1;
                            constexpr bool is_first = orel >= 1 && u == 0;
#pragma unroll
                            for (int q = 0; q < BO; ++q) {
                                float acc = is_first ? bias : L[orel + 1][q];
#pragma unroll
                                for (int vv = 0; vv < 5; ++vv) acc = fmaf(ext[2 * q + vv], w[u * 5 + vv], acc);
                                L[orel + 1][q] = acc;
                            }
                        }
                    });
                    if constexpr (i == 0) {                               // output row 2s - 1 got its last window row
                        if (s > 0) {
#pragma unroll
                            for (int q = 0; q < BO; ++q) Raw<TO>::st(ob + (q * LA) * OPITCH, L[0][q]);
                        }
                    }
                    if constexpr (i == 2) {                               // output row 2s
#pragma unroll
                        for (int q = 0; q < BO; ++q) Raw<TO>::st(ob + (W1 + q * LA) * OPITCH, L[1][q]);
                    }
                    RCX_ROW_FENCE;
                });
#pragma unroll
                for (int q = 0; q < BO; ++q) { A[0][q] = L[HS][q]; A[1][q] = L[HS + 1][q]; }
            }
            if (s >= 1) drop_pair(n, s - 1);
        }
        __syncthreads();                                                   // the slot below was lifted during the last band
        if constexpr (2 * NS - 1 < H1) {
            if (active) {
                unsigned char* ob = oring + (NS & 1) * OBAND + omine;
#pragma unroll
                for (int q = 0; q < BO; ++q) Raw<TO>::st(ob + (q * LA) * OPITCH, A[0][q]);
            }
        }
        __syncthreads();
        lift_pair(oring + ((NS - 1) & 1) * OBAND);
        drop_pair(n, NS - 1);
        if constexpr (2 * NS - 1 < H1) {
            lift_pair(oring + (NS & 1) * OBAND);
            drop_pair(n, NS);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct StepPlan {
    bool ok;
    int w0, lpc, waves;
    StepArgs args;
};

static StepPlan plan_step(int N, int C, int H, int W, int min_bytes_per_channel)
{
    StepPlan p{};
    const char* off = rcx::opt::value(rcx::opt::LANES);
    if (off && *off == '0') return p;
    if (H != W) return p;
    int lpc;
    if (W == 56 || W == 128 || W == 64 || W == 32) lpc = 16;
    else if (W == 28 || W == 14) lpc = 8;
    else return p;
    const int cpw = 64 / lpc;
    int waves = lpc == 16 ? (W == 128 ? 4 : 8) : 4;          // 128-wide bands: 16-channel blocks so that the rings fit 160 KB
    while (waves > 1 && C % (waves * cpw) != 0) waves >>= 1;
    const int cbw = waves * cpw;
    if (C % cbw != 0 || (cbw * min_bytes_per_channel) % 16 != 0) return p;
    if (!((lpc == 16 && (waves == 8 || waves == 4)) || (lpc == 8 && (waves == 4 || waves == 2)))) return p;
    p.w0 = W; p.lpc = lpc; p.waves = waves;
    p.args.N = N; p.args.C = C; p.args.nblk = C / cbw;
    int ni = 1;
    while ((long)p.args.nblk * ((N + 2 * ni - 1) / (2 * ni)) >= 2048 && ni < 4) ni *= 2;
    p.args.ni = ni;
    p.ok = true;
    return p;
}

template <class K, class... Args>
static hipError_t launch_step(K kfn, size_t lds, const StepPlan& p, bool has_bias, hipStream_t s, Args... args)
{
    RCX_SET_LDS_ONCE(kfn, lds);
    StepArgs a = p.args;
    a.has_bias = has_bias;
    const unsigned grid = (unsigned)(a.nblk * (((a.N + a.ni - 1) / a.ni + 7) / 8 * 8));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(p.waves * 64), lds, s, args..., a);
    return hipGetLastError();
}

template <int W0, int LPC, int MODE, int NW, typename TX, typename TC>
static hipError_t upadd_w(const void* x, const void* coarse, void* y, const float* w, const float* b, const StepPlan& p, hipStream_t s)
{
    constexpr int CBW = NW * 64 / LPC;
    const size_t lds = 26 * CBW * 4 + 3 * (size_t)4 * W0 * (CBW * sizeof(TX) + 16) + 2 * (size_t)4 * (W0 / 2) * (CBW * sizeof(TC) + 16);
    if (coarse) return launch_step(k_upadd_lanes<W0, LPC, MODE, NW, true, TX, TC>, lds, p, b != nullptr, s, (const TX*)x, (const TC*)coarse, (TX*)y, w, b);
    if constexpr (MODE == 0 && sizeof(TC) == 4)      // plain conv5: one instantiation per x type is enough
        return launch_step(k_upadd_lanes<W0, LPC, 0, NW, false, TX, TC>, lds, p, b != nullptr, s, (const TX*)x, (const TC*)x, (TX*)y, w, b);
    return hipErrorInvalidConfiguration;
}

template <int W0, int LPC, int MODE, typename TX, typename TC>
static hipError_t upadd_t(const void* x, const void* coarse, void* y, const float* w, const float* b, const StepPlan& p, hipStream_t s)
{
    constexpr int WA = LPC == 16 ? 8 : 4;
    if (p.waves == WA) return upadd_w<W0, LPC, MODE, WA, TX, TC>(x, coarse, y, w, b, p, s);
    return upadd_w<W0, LPC, MODE, WA / 2, TX, TC>(x, coarse, y, w, b, p, s);
}

template <int MODE, typename TX, typename TC>
static hipError_t upadd_m(const void* x, const void* coarse, void* y, const float* w, const float* b, const StepPlan& p, hipStream_t s)
{
    if (p.w0 == 56) return upadd_t<56, 16, MODE, TX, TC>(x, coarse, y, w, b, p, s);
    if (p.w0 == 64) return upadd_t<64, 16, MODE, TX, TC>(x, coarse, y, w, b, p, s);
    if (p.w0 == 128) {
        if (p.waves == 4) return upadd_w<128, 16, MODE, 4, TX, TC>(x, coarse, y, w, b, p, s);
        return upadd_w<128, 16, MODE, 2, TX, TC>(x, coarse, y, w, b, p, s);
    }
    if (p.w0 == 32) return upadd_t<32, 16, MODE, TX, TC>(x, coarse, y, w, b, p, s);
    if (p.w0 == 28) return upadd_t<28, 8, MODE, TX, TC>(x, coarse, y, w, b, p, s);
    return upadd_t<14, 8, MODE, TX, TC>(x, coarse, y, w, b, p, s);
}

template <int W0, int LPC, int NW, typename TX, typename TO>
static hipError_t down5_w(const void* x, void* y, const float* w, const float* b, const StepPlan& p, hipStream_t s)
{
    constexpr int CBW = NW * 64 / LPC;
    const size_t lds = 26 * CBW * 4 + 2 * (size_t)4 * W0 * (CBW * sizeof(TX) + 16) + 2 * (size_t)2 * (W0 / 2) * (CBW * sizeof(TO) + 16);
    return launch_step(k_down5_lanes<W0, LPC, NW, TX, TO>, lds, p, b != nullptr, s, (const TX*)x, (TO*)y, w, b);
}

template <int W0, int LPC, typename TX, typename TO>
static hipError_t down5_t(const void* x, void* y, const float* w, const float* b, const StepPlan& p, hipStream_t s)
{
    constexpr int WA = LPC == 16 ? 8 : 4;
    if (p.waves == WA) return down5_w<W0, LPC, WA, TX, TO>(x, y, w, b, p, s);
    return down5_w<W0, LPC, WA / 2, TX, TO>(x, y, w, b, p, s);
}

template <typename TX, typename TO>
static hipError_t down5_m(const void* x, void* y, const float* w, const float* b, const StepPlan& p, hipStream_t s)
{
    if (p.w0 == 56) return down5_t<56, 16, TX, TO>(x, y, w, b, p, s);
    if (p.w0 == 64) return down5_t<64, 16, TX, TO>(x, y, w, b, p, s);
    if (p.w0 == 128) return down5_w<128, 16, 4, TX, TO>(x, y, w, b, p, s);
    if (p.w0 == 32) return down5_t<32, 16, TX, TO>(x, y, w, b, p, s);
    if (p.w0 == 28) return down5_t<28, 8, TX, TO>(x, y, w, b, p, s);
    return down5_t<14, 8, TX, TO>(x, y, w, b, p, s);
}

// LDS footprint of k_upadd_lanes; a 128-wide plane with float32 operands only fits with 8-channel blocks (2 waves)
static size_t upadd_lds(int W, int cbw, int xb, int cbytes)
{
    return (size_t)26 * cbw * 4 + 3 * (size_t)4 * W * (cbw * xb + 16) + 2 * (size_t)4 * (W / 2) * (cbw * cbytes + 16);
}

static StepPlan plan_upadd(int N, int C, int H, int W, int xb, int cbytes)
{
    StepPlan p = plan_step(N, C, H, W, xb < cbytes ? xb : cbytes);
    if (!p.ok) return p;
    const int cpw = 64 / p.lpc;
    if (upadd_lds(W, p.waves * cpw, xb, cbytes) > 160 * 1024) {
        if (W != 128 || p.waves != 4 || C % (2 * cpw) != 0) { p.ok = false; return p; }
        p.waves = 2;
        p.args.nblk = C / (2 * cpw);
        if (upadd_lds(W, 2 * cpw, xb, cbytes) > 160 * 1024 || (2 * cpw * (xb < cbytes ? xb : cbytes)) % 16 != 0) p.ok = false;
    }
    return p;
}

}  // namespace lanes

// x and y share a dtype; coarse may be float32 or the same as x
// plain stride-1 conv5 (no coarse operand) in the same kernel
bool conv5_lanes_applicable(int N, int C, int H, int W, int k, int x_dt, int out_dt)
{
    if (x_dt > 1 || out_dt > 1) return false;                       // float16 I/O: the channel-per-lane kernels and the generic schedule (rcx_api.hip)
    if (k != 5 || out_dt != x_dt) return false;
    return lanes::plan_upadd(N, C, H, W, x_dt == 1 ? 2 : 4, 4).ok;
}

hipError_t conv5_lanes(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int x_dt, hipStream_t s)
{
    const lanes::StepPlan p = lanes::plan_upadd(N, C, H, W, x_dt == 1 ? 2 : 4, 4);
    if (!p.ok) return hipErrorInvalidConfiguration;
    if (x_dt == 1) return lanes::upadd_m<0, bf16_t, float>(x, nullptr, y, w, b, p, s);
    return lanes::upadd_m<0, float, float>(x, nullptr, y, w, b, p, s);
}

bool upadd_lanes_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt)
{
    if (x_dt > 1 || c_dt > 1 || out_dt > 1) return false;                       // float16 I/O: the channel-per-lane kernels and the generic schedule (rcx_api.hip)
    if (k != 5 || out_dt != x_dt || Hc * 2 != H || Wc * 2 != W) return false;
    if (!(c_dt == x_dt || c_dt == 0)) return false;
    const int xb = x_dt == 1 ? 2 : 4, cbytes = c_dt == 1 ? 2 : 4;
    return lanes::plan_upadd(N, C, H, W, xb, cbytes).ok;
}

hipError_t upadd_lanes(const void* x, const void* coarse, void* y, const float* w, const float* b,
                       int N, int C, int H, int W, int mode, int x_dt, int c_dt, hipStream_t s)
{
    const int xb = x_dt == 1 ? 2 : 4, cbytes = c_dt == 1 ? 2 : 4;
    const lanes::StepPlan p = lanes::plan_upadd(N, C, H, W, xb, cbytes);
    if (!p.ok) return hipErrorInvalidConfiguration;
#define RCX_UP(TX, TC) (mode == 1 ? lanes::upadd_m<1, TX, TC>(x, coarse, y, w, b, p, s) : lanes::upadd_m<0, TX, TC>(x, coarse, y, w, b, p, s))
    if (x_dt == 1 && c_dt == 1) return RCX_UP(bf16_t, bf16_t);
    if (x_dt == 1) return RCX_UP(bf16_t, float);
    return RCX_UP(float, float);
#undef RCX_UP
}

bool down5_lanes_applicable(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt)
{
    if (in_dt > 1 || out_dt > 1) return false;                       // float16 I/O: the channel-per-lane kernels and the generic schedule (rcx_api.hip)
    if (k != 5 || stride != 2) return false;
    if (!(out_dt == in_dt || out_dt == 0)) return false;
    return lanes::plan_step(N, C, H, W, in_dt == 1 ? 2 : 4).ok;
}

hipError_t down5_lanes(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int in_dt, int out_dt, hipStream_t s)
{
    const lanes::StepPlan p = lanes::plan_step(N, C, H, W, in_dt == 1 ? 2 : 4);
    if (!p.ok) return hipErrorInvalidConfiguration;
    if (in_dt == 1 && out_dt == 1) return lanes::down5_m<bf16_t, bf16_t>(x, y, w, b, p, s);
    if (in_dt == 1) return lanes::down5_m<bf16_t, float>(x, y, w, b, p, s);
    return lanes::down5_m<float, float>(x, y, w, b, p, s);
}

}  // namespace rcx
