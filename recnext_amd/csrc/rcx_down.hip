// Register-resident kernel for the stage-transition depthwise conv of the model skeleton:
// nn.Conv2d(C, 2C, kernel_size=7, stride=2, padding=3, groups=C) + folded eval BatchNorm (model/recnext.py:165-166, :170),
// on the 7 * 2^k planes.  Same lane layout as rcx_lanes.hip: LPC consecutive lanes own one OUTPUT channel (output
// channel o reads input channel o/2, so two lane groups read the same x bytes from LDS -- a broadcast), lane j of a group
// holds B0 adjacent input columns and produces B0/2 output columns; horizontal taps come from DPP lane shifts, the
// seven window rows stream through registers.  x enters in 4-row bands through a 2-slot LDS ring (coalesced 16-byte
// global accesses), finished output row pairs leave through a second small ring.  One barrier per band.
#include "rcx_lanes.h"
#include "rcx_opts.h"
#include "rcx_launch.h"

namespace rcx {
namespace lanes {

struct DownArgs {
    int N, Cin;
    int nblk;          // input-channel blocks per image
    int ni;            // images per workgroup
    int has_bias;
};

constexpr int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// ext[e] = column (e - PADL) of the row, columns outside the lane's own B0 fetched from the neighbouring lanes
template <int LPC, int B0, int PADL, int PADR>
__device__ __forceinline__ void make_ext_wide(const float (&row)[B0], float (&ext)[B0 + PADL + PADR])
{
    sfor<B0 + PADL + PADR>([&](auto E) RCX_INL {
        constexpr int e = decltype(E)::value;
        constexpr int col = e - PADL;
        constexpr int off = floor_div(col, B0);
        constexpr int sub = col - off * B0;
        if constexpr (off == 0) ext[e] = row[sub];
        else if constexpr (off < 0) ext[e] = from_left<-off, LPC>(row[sub]);
        else ext[e] = from_right<off, LPC>(row[sub]);
    });
}

template <int W0, int LPC, int NW, int K, typename TIO>
__global__ __launch_bounds__(NW * 64)
void k_down_lanes(const TIO* __restrict__ x, TIO* __restrict__ y, const float* __restrict__ wpack, const float* __restrict__ bias_pack, DownArgs a)
{
    static_assert(K == 7, "window rows per band are laid out for k = 7");
    constexpr int LA = lanes_active(W0, LPC);
    constexpr int B0 = W0 / LA, BO = B0 / 2, H0 = W0, W1 = W0 / 2, H1 = W1, PAD = K / 2;
    static_assert(B0 * LA == W0 && B0 >= 2 && (B0 % 2) == 0, "plane width must be LA * B0, B0 even");
    constexpr int SR = 4, NS = (H0 + SR - 1) / SR;   // rows past the plane (14 = 3.5 bands) are staged as zeros
    constexpr int OPW = 64 / LPC, OCB = NW * OPW, ICB = OCB / 2, NT = NW * 64, ESZ = (int)sizeof(TIO);
    constexpr int XPITCH = ICB * ESZ + 16, OPITCH = OCB * ESZ + 16, XCPP = ICB * ESZ / 16, OCPP = OCB * ESZ / 16;
    static_assert(XCPP >= 1 && (XCPP & (XCPP - 1)) == 0, "input channel block must be a power-of-two number of 16-byte chunks");
    constexpr int XBAND = SR * W0 * XPITCH, OBAND = 2 * W1 * OPITCH;
    constexpr int XN = SR * W0 * XCPP, ON = 2 * W1 * OCPP;
    constexpr int XST = (XN + NT - 1) / NT, OST = (ON + NT - 1) / NT;
    constexpr int NTAP = K * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* taps = reinterpret_cast<float*>(smem);                         // [NTAP + 1][OCB]
    unsigned char* xring = smem + (NTAP + 1) * OCB * 4;                    // [2][XBAND]
    unsigned char* oring = xring + 2 * XBAND;                              // [2][OBAND]

    const int tid = threadIdx.x;
    // workgroup -> (channel block, image group): the channel blocks of one image group get ids that are equal mod 8,
    // i.e. land on the same XCD (round-robin dispatch) close in time and share its L2 lines
    const int cb = (blockIdx.x % (8 * a.nblk)) / 8;
    const int n0 = ((blockIdx.x / (8 * a.nblk)) * 8 + blockIdx.x % 8) * a.ni;
    const int n1 = n0 + a.ni < a.N ? n0 + a.ni : a.N;
    const int ci0 = cb * ICB, co0 = 2 * ci0, Cout = 2 * a.Cin;
    const size_t ximg = (size_t)H0 * W0 * a.Cin, yimg = (size_t)H1 * W1 * Cout;

    int xg[XST], xl[XST], xrow[XST], og[OST], ol[OST], opix[OST];
    bool xhave[XST], ohave[OST];
    sfor<XST>([&](auto I) RCX_INL {
        constexpr int i = decltype(I)::value;
        int cidx = tid + i * NT;
        xhave[i] = (i + 1) * NT <= XN || cidx < XN;
        cidx = xhave[i] ? cidx : XN - 1;
        const int p = cidx / XCPP, part = cidx % XCPP;
        xg[i] = p * a.Cin * ESZ + part * 16;
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
        og[i] = p * Cout * ESZ + part * 16;
        ol[i] = lds_slot<W1, BO, LA>(p) * OPITCH + part * 16;
    });
    u32x4 xv[XST], yv[OST];
    auto prefetch = [&](int n, int band) RCX_INL {
        const unsigned char* xp = reinterpret_cast<const unsigned char*>(x + (size_t)n * ximg + ci0) + (size_t)band * (SR * W0) * a.Cin * ESZ;
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
    // output rows leave in pairs: pair s = rows (2s - 1, 2s), s = 0 .. NS (the first and last pair hold one row)
    auto lift_pair = [&](const unsigned char* slot) RCX_INL {
        sfor<OST>([&](auto I) RCX_INL { yv[decltype(I)::value] = *reinterpret_cast<const u32x4*>(slot + ol[decltype(I)::value]); });
    };
    auto drop_pair = [&](int n, int s) RCX_INL {
        unsigned char* yp = reinterpret_cast<unsigned char*>(y + (size_t)n * yimg + co0) + ((ptrdiff_t)(2 * s - 1) * W1) * Cout * ESZ;
        sfor<OST>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const bool row_ok = opix[i] < W1 ? (s > 0 && 2 * s - 1 < H1) : 2 * s < H1;
            if (ohave[i] && row_ok) *reinterpret_cast<u32x4*>(yp + og[i]) = yv[i];
        });
    };
    if (n0 < n1) prefetch(n0, 0);

    {   // taps (k, k, 2C) and the bias row -> LDS
        constexpr int Q4 = OCB / 4, TOTAL = (NTAP + 1) * Q4, TB = (TOTAL + NT - 1) / NT;
        float4 t[TB];
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            int idx = tid + i * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int row = idx / Q4, part = idx % Q4;
            if (row < NTAP) t[i] = *reinterpret_cast<const float4*>(wpack + (size_t)row * Cout + co0 + part * 4);
            else if (a.has_bias) t[i] = *reinterpret_cast<const float4*>(bias_pack + co0 + part * 4);
            else t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        });
        sfor<TB>([&](auto I) RCX_INL {
            constexpr int i = decltype(I)::value;
            const int idx = tid + i * NT;
            if (idx < TOTAL) *reinterpret_cast<float4*>(taps + (size_t)idx * 4) = t[i];
        });
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int lig = lane % LPC;                     // lane inside its channel group
    const int oc = wave * OPW + lane / LPC;         // output channel inside the block
    const bool active = lig < LA;
    const int xmine = lig * XPITCH + (oc >> 1) * ESZ;
    const int omine = lig * OPITCH + oc * ESZ;

    for (int n = n0; n < n1; ++n) {
        // Every output keeps two partial sums (taps 0,2,4,6 / 1,3,5 of each window row): two taps per v_pk_fma_f32, added
        // when the output row leaves (DPP moves keep the SIMD in its slow issue mode, where a packed FMA costs as much
        // as a scalar one; rcx_lanes.h).
        f32x2 A[3][BO];                             // partial sums of the three output rows that straddle a band boundary
        f32x2 wp[K][K / 2];                         // tap pairs (2k, 2k+1) of every window row
        float wl[K], bias = 0.f;                    // the last tap of every window row
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
                if (s == 0) {
#pragma unroll
                    for (int u = 0; u < K; ++u) {
#pragma unroll
                        for (int k = 0; k < K / 2; ++k) wp[u][k] = f32x2{taps[(u * K + 2 * k) * OCB + oc], taps[(u * K + 2 * k + 1) * OCB + oc]};
                        wl[u] = taps[(u * K + K - 1) * OCB + oc];
                    }
                    bias = taps[NTAP * OCB + oc];
#pragma unroll
                    for (int u = 0; u < K; ++u) {
#pragma unroll
                        for (int k = 0; k < K / 2; ++k) asm volatile("" : "+v"(wp[u][k]));
                        asm volatile("" : "+v"(wl[u]));
                    }
                }
                f32x2 L[5][BO];                     // output rows 2s-1 .. 2s+3
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < BO; ++q) L[k][q] = s == 0 ? f32x2{bias, 0.f} : A[k][q];
                const unsigned char* xb = xs + xmine;
                unsigned char* ob = os + omine;
                float nxt[B0];
#pragma unroll
                for (int j = 0; j < B0; ++j) nxt[j] = Raw<TIO>::ld(xb + (j * LA) * XPITCH);
                sfor<SR>([&](auto I) RCX_INL {
                    constexpr int i = decltype(I)::value;
                    float row[B0], ext[B0 + PAD + 2];
#pragma unroll
                    for (int j = 0; j < B0; ++j) row[j] = nxt[j];
                    if constexpr (i + 1 < SR) {
#pragma unroll
                        for (int j = 0; j < B0; ++j) nxt[j] = Raw<TIO>::ld(xb + ((i + 1) * W0 + j * LA) * XPITCH);
                    }
                    make_ext_wide<LPC, B0, PAD, 2>(row, ext);
                    sfor<5>([&](auto R) RCX_INL {
                        constexpr int orel = decltype(R)::value - 1;             // output row 2s + orel
                        constexpr int u = i - 2 * orel + PAD;                   // window row of input row 4s + i
                        if constexpr (u >= 0 && u < K) {
#pragma unroll
                            for (int q = 0; q < BO; ++q) {
                                f32x2 acc = (u == 0) ? f32x2{bias, 0.f} : L[orel + 1][q];    // u == 0: the row starts inside this band
#pragma unroll
                                for (int k = 0; k < K / 2; ++k)
                                    acc = __builtin_elementwise_fma(f32x2{ext[2 * q + 2 * k], ext[2 * q + 2 * k + 1]}, wp[u][k], acc);
                                acc.x = fmaf(ext[2 * q + K - 1], wl[u], acc.x);
                                L[orel + 1][q] = acc;
                            }
                        }
                    });
                    if constexpr (i == 1) {                                      // output row 2s - 1 is complete
                        if (s > 0) {
#pragma unroll
                            for (int q = 0; q < BO; ++q) Raw<TIO>::st(ob + (q * LA) * OPITCH, L[0][q].x + L[0][q].y);
                        }
                    }
                    if constexpr (i == 3) {                                      // output row 2s is complete
#pragma unroll
                        for (int q = 0; q < BO; ++q) Raw<TIO>::st(ob + (W1 + q * LA) * OPITCH, L[1][q].x + L[1][q].y);
                    }
                    RCX_ROW_FENCE;
                });
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < BO; ++q) A[k][q] = L[2 + k][q];
            }
            if (s >= 1) drop_pair(n, s - 1);
        }
        // the last output row (its remaining window rows are padding) goes into the pair after the last band
        __syncthreads();                            // its slot was read (lift_pair) during the last band
        if constexpr (2 * NS - 1 < H1) {
            if (active) {
                unsigned char* ob = oring + (NS & 1) * OBAND + omine;
#pragma unroll
                for (int q = 0; q < BO; ++q) Raw<TIO>::st(ob + (q * LA) * OPITCH, A[0][q].x + A[0][q].y);
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

struct DownPlan {
    bool ok;
    int w0, lpc, waves;
    size_t lds;
    DownArgs args;
};

static int env_int_d(rcx::opt::Id id, int dflt)
{
    const char* v = rcx::opt::value(id);
    return v && *v ? atoi(v) : dflt;
}

static DownPlan plan_down(int N, int Cin, int H, int W, int k, int stride, int dtype)
{
    DownPlan p{};
    if (env_int_d(rcx::opt::LANES, 1) == 0) return p;
    if (k != 7 || stride != 2 || H != W) return p;
    int lpc;
    if (W == 56 || W == 128 || W == 64 || W == 32) lpc = 16;
    else if (W == 28 || W == 14) lpc = 8;
    else return p;
    const int esz = dtype == 1 ? 2 : 4;
    const int opw = 64 / lpc;
    int waves = lpc == 16 ? 16 : 8;
    while (waves > 1 && (Cin % (waves * opw / 2) != 0)) waves >>= 1;
    const int icb = waves * opw / 2;
    if (icb < 1 || Cin % icb != 0 || (icb * esz) % 16 != 0) return p;
    const int ocb = 2 * icb;
    p.lds = (size_t)50 * ocb * 4 + (size_t)2 * 4 * W * (icb * esz + 16) + (size_t)2 * 2 * (W / 2) * (ocb * esz + 16);
    if (p.lds > 160 * 1024) return p;
    p.w0 = W; p.lpc = lpc; p.waves = waves;
    p.args.N = N; p.args.Cin = Cin; p.args.nblk = Cin / icb;
    int ni = 1;
    while ((long)p.args.nblk * ((N + 2 * ni - 1) / (2 * ni)) >= 1024 && ni < 4) ni *= 2;
    p.args.ni = ni;
    p.ok = true;
    return p;
}

template <int W0, int LPC, int NW, typename TIO>
static hipError_t launch_down_w(const void* x, void* y, const float* w, const float* b, const DownPlan& p, hipStream_t s)
{
    auto kfn = k_down_lanes<W0, LPC, NW, 7, TIO>;
    RCX_SET_LDS_ONCE(kfn, p.lds);
    DownArgs a = p.args;
    a.has_bias = b != nullptr;
    const unsigned grid = (unsigned)(a.nblk * (((a.N + a.ni - 1) / a.ni + 7) / 8 * 8));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), p.lds, s, (const TIO*)x, (TIO*)y, w, b, a);
    return hipGetLastError();
}

template <int W0, int LPC, typename TIO>
static hipError_t launch_down_t(const void* x, void* y, const float* w, const float* b, const DownPlan& p, hipStream_t s)
{
    constexpr int OPW = 64 / LPC;
    switch (p.waves) {
    case 16:
        if constexpr (LPC == 16) return launch_down_w<W0, LPC, 16, TIO>(x, y, w, b, p, s);
        return hipErrorInvalidConfiguration;
    case 8: return launch_down_w<W0, LPC, 8, TIO>(x, y, w, b, p, s);
    case 4:
        if constexpr (4 * OPW / 2 * sizeof(TIO) >= 16) return launch_down_w<W0, LPC, 4, TIO>(x, y, w, b, p, s);
        return hipErrorInvalidConfiguration;
    default:
        return hipErrorInvalidConfiguration;
    }
}

template <typename TIO>
static hipError_t launch_down(const void* x, void* y, const float* w, const float* b, const DownPlan& p, hipStream_t s)
{
    if (p.w0 == 56) return launch_down_t<56, 16, TIO>(x, y, w, b, p, s);
    if (p.w0 == 128) return launch_down_t<128, 16, TIO>(x, y, w, b, p, s);
    if (p.w0 == 64) return launch_down_t<64, 16, TIO>(x, y, w, b, p, s);
    if (p.w0 == 32) return launch_down_t<32, 16, TIO>(x, y, w, b, p, s);
    if (p.w0 == 28) return launch_down_t<28, 8, TIO>(x, y, w, b, p, s);
    if (p.w0 == 14) return launch_down_t<14, 8, TIO>(x, y, w, b, p, s);
    return hipErrorInvalidConfiguration;
}

}  // namespace lanes

bool down_lanes_applicable(int N, int Cin, int H, int W, int k, int stride, int dtype)
{
    if (dtype > 1) return false;                       // float16 I/O: the channel-per-lane kernels and the generic schedule (rcx_api.hip)
    const lanes::DownPlan p = lanes::plan_down(N, Cin, H, W, k, stride, dtype);
    return p.ok && (p.waves == 16 || p.waves == 8 || p.waves == 4);
}

hipError_t down_lanes(const void* x, void* y, const float* w, const float* b, int N, int Cin, int H, int W, int k, int stride, int dtype, hipStream_t s)
{
    const lanes::DownPlan p = lanes::plan_down(N, Cin, H, W, k, stride, dtype);
    if (!p.ok) return hipErrorInvalidConfiguration;
    if (dtype == 1) return lanes::launch_down<bf16_t>(x, y, w, b, p, s);
    return lanes::launch_down<float>(x, y, w, b, p, s);
}

}  // namespace rcx
