// Generic (any shape, any odd k, any C) depthwise kernels for gfx950.
//
//   k_dwconv        y = dwconv_{k,stride}(x)                         model/recnext.py:21-22,28
//   k_upadd_dwconv  y = dwconv_{k,1}(x + resize(coarse -> size(x)))  model/recnext.py:33-34, model/recattn.py:67
//
// These are the schedule of last resort: one launch per ladder step, inputs read straight from
// global memory (L1/L2 absorb the k x k window reuse), every thread producing R horizontally
// adjacent outputs for a vector of V channels so the window is read once per row instead of once
// per tap.  NHWC puts channels on consecutive lanes: every wave-load is a run of whole pixels.
// The fused single-launch schedules live in rcx_plane.hip; rcx_api.hip picks between them.
#include <type_traits>
#include "rcx_common.h"
#include "rcx_launch.h"

namespace rcx {

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int id = 0; };
template <> struct DT<bf16_t> { static constexpr int id = 1; };
template <> struct DT<f16_t> { static constexpr int id = 2; };

struct ConvGeom {
    int N, C, H, W;       // input extent (of x); C = OUTPUT channels (= weights' channel count)
    int Cin;              // input channels: C / MULT (channel-multiplier depthwise conv), else == C
    int Ho, Wo;           // output extent
    int Hc, Wc;           // coarse extent (upadd only)
    int k, stride;
    int strips;           // ceil(Wo / R)
    int cvecs;            // C / V
    float sy, sx;         // Hc/H, Wc/W resize scales (upadd only)
};

// T(n, iy, ix, c..c+V) = x + resize(coarse), zero outside the plane (the conv's zero padding
// applies to the *sum*, as in conv(f + x) at model/recnext.py:33).
template <typename TX, typename TC, int V, int MODE, bool HAS_COARSE, int MULT = 1>
__device__ __forceinline__ void fetch_sum(const TX* __restrict__ xn, const TC* __restrict__ cn, const ConvGeom& g,
                                          int iy, int ix, int c, float (&out)[V])
{
    if constexpr (MULT == 1) {
        load_vec<V>(xn + ((size_t)iy * g.W + ix) * g.C + c, out);
    } else {
        // grouped conv with groups = Cin: output channel o reads input channel o / MULT (nn.Conv2d semantics)
        static_assert(V % MULT == 0 && !HAS_COARSE, "channel multiplier");
        float t[V / MULT];
        load_vec<V / MULT>(xn + ((size_t)iy * g.W + ix) * g.Cin + c / MULT, t);
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = t[i / MULT];
    }
    if constexpr (HAS_COARSE) {
        if constexpr (MODE == 1) {
            int cy = nearest_src(iy, g.Hc, g.sy), cx = nearest_src(ix, g.Wc, g.sx);
            float t[V];
            load_vec<V>(cn + ((size_t)cy * g.Wc + cx) * g.C + c, t);
#pragma unroll
            for (int i = 0; i < V; ++i) out[i] += t[i];
        } else {
            Lerp ly = bilinear_src(iy, g.Hc, g.sy), lx = bilinear_src(ix, g.Wc, g.sx);
            float a[V], b[V], d[V], e[V];
            load_vec<V>(cn + ((size_t)ly.i0 * g.Wc + lx.i0) * g.C + c, a);
            load_vec<V>(cn + ((size_t)ly.i0 * g.Wc + lx.i1) * g.C + c, b);
            load_vec<V>(cn + ((size_t)ly.i1 * g.Wc + lx.i0) * g.C + c, d);
            load_vec<V>(cn + ((size_t)ly.i1 * g.Wc + lx.i1) * g.C + c, e);
            const float wy1 = ly.lam, wy0 = 1.f - ly.lam, wx1 = lx.lam, wx0 = 1.f - lx.lam;
#pragma unroll
            for (int i = 0; i < V; ++i)
                out[i] += wy0 * (wx0 * a[i] + wx1 * b[i]) + wy1 * (wx0 * d[i] + wx1 * e[i]);
        }
    }
}

// One thread: V channels x R adjacent output columns of one output row.
// K == 0 selects the runtime-k body.
template <typename TX, typename TC, typename TO, int K, int STRIDE, int V, int R, int MODE, bool HAS_COARSE, int MULT = 1>
__global__ void __launch_bounds__(256)
k_conv_generic(const TX* __restrict__ x, const TC* __restrict__ coarse, TO* __restrict__ y,
               const float* __restrict__ w, const float* __restrict__ bias, ConvGeom g)
{
    const int k = K ? K : g.k;
    const int p = k / 2;
    const long long total = (long long)g.N * g.Ho * g.strips * g.cvecs;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        long long r = t;
        const int cv = (int)(r % g.cvecs); r /= g.cvecs;
        const int st = (int)(r % g.strips); r /= g.strips;
        const int oy = (int)(r % g.Ho);
        const int n = (int)(r / g.Ho);
        const int c = cv * V;
        const int ox0 = st * R;
        const TX* xn = x + (size_t)n * g.H * g.W * g.Cin;
        const TC* cn = HAS_COARSE ? coarse + (size_t)n * g.Hc * g.Wc * g.C : nullptr;

        float acc[R][V];
        {
            float b[V];
#pragma unroll
            for (int i = 0; i < V; ++i) b[i] = 0.f;
            if (bias) load_vec<V>(bias + c, b);
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) acc[j][i] = b[i];
        }
        const int span = (R - 1) * STRIDE + k;           // input columns touched by the strip
        for (int u = 0; u < k; ++u) {
            const int iy = oy * STRIDE + u - p;
            if (iy < 0 || iy >= g.H) continue;
            for (int s = 0; s < span; ++s) {
                const int ix = ox0 * STRIDE + s - p;
                if (ix < 0 || ix >= g.W) continue;
                float in[V];
                fetch_sum<TX, TC, V, MODE, HAS_COARSE, MULT>(xn, cn, g, iy, ix, c, in);
                // input column s feeds output j through tap v = s - j*STRIDE
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int v = s - j * STRIDE;
                    if (v >= 0 && v < k) {
                        float wv[V];
                        load_vec<V>(w + ((size_t)u * k + v) * g.C + c, wv);
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[j][i] = fmaf(wv[i], in[i], acc[j][i]);
                    }
                }
            }
        }
        TO* yo = y + (((size_t)n * g.Ho + oy) * g.Wo + ox0) * g.C + c;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (ox0 + j < g.Wo) store_vec<V>(yo + (size_t)j * g.C, acc[j]);
    }
}

template <typename TX, typename TC, typename TO, int K, int STRIDE, int V, int MODE, bool HAS_COARSE, int MULT = 1>
static hipError_t launch_rv(const void* x, const void* coarse, void* y, const float* w, const float* b,
                            ConvGeom g, hipStream_t stream)
{
    constexpr int R = 4;
    g.strips = (g.Wo + R - 1) / R;
    g.cvecs = g.C / V;
    const long long total = (long long)g.N * g.Ho * g.strips * g.cvecs;
    long long blocks = (total + 255) / 256;
    if (blocks > 256LL * 64) blocks = 256LL * 64;         // grid-stride beyond 64 blocks per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_conv_generic<TX, TC, TO, K, STRIDE, V, R, MODE, HAS_COARSE, MULT>), dim3((unsigned)blocks), dim3(256), 0, stream,
                       (const TX*)x, (const TC*)coarse, (TO*)y, w, b, g);
    return hipGetLastError();
}

template <typename TX, typename TC, typename TO, int K, int STRIDE, int MODE, bool HAS_COARSE>
static hipError_t launch_v(const void* x, const void* coarse, void* y, const float* w, const float* b,
                           const ConvGeom& g, hipStream_t s)
{
    // widest channel vector that divides C and keeps 16-byte loads on the narrowest operand
    if (g.C % 8 == 0 && sizeof(TX) == 2 && sizeof(TO) == 2)
        return launch_rv<TX, TC, TO, K, STRIDE, 8, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    if (g.C % 4 == 0) return launch_rv<TX, TC, TO, K, STRIDE, 4, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    if (g.C % 2 == 0) return launch_rv<TX, TC, TO, K, STRIDE, 2, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    return launch_rv<TX, TC, TO, K, STRIDE, 1, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
}

template <typename TX, typename TC, typename TO, int STRIDE, int MODE, bool HAS_COARSE>
static hipError_t launch_k(const void* x, const void* coarse, void* y, const float* w, const float* b,
                           const ConvGeom& g, hipStream_t s)
{
    switch (g.k) {
    case 3: return launch_v<TX, TC, TO, 3, STRIDE, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    case 5: return launch_v<TX, TC, TO, 5, STRIDE, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    case 7: return launch_v<TX, TC, TO, 7, STRIDE, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    default: return launch_v<TX, TC, TO, 0, STRIDE, MODE, HAS_COARSE>(x, coarse, y, w, b, g, s);
    }
}

template <typename TX, typename TO>
static hipError_t dwconv_io(const void* x, void* y, const float* w, const float* b, const ConvGeom& g, hipStream_t s)
{
    if (g.stride == 2) return launch_k<TX, float, TO, 2, 0, false>(x, nullptr, y, w, b, g, s);
    return launch_k<TX, float, TO, 1, 0, false>(x, nullptr, y, w, b, g, s);
}

hipError_t generic_dwconv(const void* x, void* y, const float* w, const float* b,
                          int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt, hipStream_t s)
{
    ConvGeom g{};
    g.N = N; g.C = C; g.Cin = C; g.H = H; g.W = W; g.k = k; g.stride = stride;
    const int p = k / 2;
    g.Ho = (H + 2 * p - k) / stride + 1;
    g.Wo = (W + 2 * p - k) / stride + 1;
    if (in_dt == 0 && out_dt == 0) return dwconv_io<float, float>(x, y, w, b, g, s);
    if (in_dt == 1 && out_dt == 0) return dwconv_io<bf16_t, float>(x, y, w, b, g, s);
    if (in_dt == 0 && out_dt == 1) return dwconv_io<float, bf16_t>(x, y, w, b, g, s);
    if (in_dt == 1 && out_dt == 1) return dwconv_io<bf16_t, bf16_t>(x, y, w, b, g, s);
    // float16 pairs with itself and with the float32 intermediates
    if (in_dt == 2 && out_dt == 0) return dwconv_io<f16_t, float>(x, y, w, b, g, s);
    if (in_dt == 0 && out_dt == 2) return dwconv_io<float, f16_t>(x, y, w, b, g, s);
    if (in_dt == 2 && out_dt == 2) return dwconv_io<f16_t, f16_t>(x, y, w, b, g, s);
    return hipErrorInvalidValue;
}

template <typename TX, typename TC, typename TO>
static hipError_t upadd_io(const void* x, const void* c, void* y, const float* w, const float* b,
                           const ConvGeom& g, int mode, hipStream_t s)
{
    if (mode == 0) return launch_k<TX, TC, TO, 1, 0, true>(x, c, y, w, b, g, s);
    return launch_k<TX, TC, TO, 1, 1, true>(x, c, y, w, b, g, s);
}

hipError_t generic_upadd_dwconv(const void* x, const void* coarse, void* y, const float* w, const float* b,
                                int N, int C, int H, int W, int Hc, int Wc, int k, int mode,
                                int x_dt, int c_dt, int out_dt, hipStream_t s)
{
    if (!coarse) return generic_dwconv(x, y, w, b, N, C, H, W, k, 1, x_dt, out_dt, s);
    ConvGeom g{};
    g.N = N; g.C = C; g.Cin = C; g.H = H; g.W = W; g.Ho = H; g.Wo = W; g.Hc = Hc; g.Wc = Wc; g.k = k; g.stride = 1;
    g.sy = (float)Hc / (float)H;
    g.sx = (float)Wc / (float)W;
    if (x_dt == 2 || c_dt == 2 || out_dt == 2) {
        // float16: x and y float16 with a float32 (RecConv2d) or float16 (RecAttn2d) coarse operand; float32 x with float16 y
        if (x_dt == 2 && c_dt == 0 && out_dt == 2) return upadd_io<f16_t, float, f16_t>(x, coarse, y, w, b, g, mode, s);
        if (x_dt == 2 && c_dt == 2 && out_dt == 2) return upadd_io<f16_t, f16_t, f16_t>(x, coarse, y, w, b, g, mode, s);
        if (x_dt == 2 && c_dt == 0 && out_dt == 0) return upadd_io<f16_t, float, float>(x, coarse, y, w, b, g, mode, s);
        if (x_dt == 0 && c_dt == 0 && out_dt == 2) return upadd_io<float, float, f16_t>(x, coarse, y, w, b, g, mode, s);
        return hipErrorInvalidValue;
    }
    const int key = x_dt * 4 + c_dt * 2 + out_dt;
    switch (key) {
    case 0: return upadd_io<float, float, float>(x, coarse, y, w, b, g, mode, s);
    case 1: return upadd_io<float, float, bf16_t>(x, coarse, y, w, b, g, mode, s);
    case 2: return upadd_io<float, bf16_t, float>(x, coarse, y, w, b, g, mode, s);
    case 3: return upadd_io<float, bf16_t, bf16_t>(x, coarse, y, w, b, g, mode, s);
    case 4: return upadd_io<bf16_t, float, float>(x, coarse, y, w, b, g, mode, s);
    case 5: return upadd_io<bf16_t, float, bf16_t>(x, coarse, y, w, b, g, mode, s);
    case 6: return upadd_io<bf16_t, bf16_t, float>(x, coarse, y, w, b, g, mode, s);
    default: return upadd_io<bf16_t, bf16_t, bf16_t>(x, coarse, y, w, b, g, mode, s);
    }
}

// ---- depthwise conv with channel multiplier 2 (groups = Cin, Cout = 2*Cin): Downsample.token_mixer ----
template <typename TX, typename TO, int K, int STRIDE>
static hipError_t mult2_v(const void* x, void* y, const float* w, const float* b, const ConvGeom& g, hipStream_t s)
{
    if (g.C % 8 == 0) return launch_rv<TX, float, TO, K, STRIDE, 8, 0, false, 2>(x, nullptr, y, w, b, g, s);
    if (g.C % 4 == 0) return launch_rv<TX, float, TO, K, STRIDE, 4, 0, false, 2>(x, nullptr, y, w, b, g, s);
    return launch_rv<TX, float, TO, K, STRIDE, 2, 0, false, 2>(x, nullptr, y, w, b, g, s);
}

template <typename TX, typename TO>
static hipError_t mult2_ks(const void* x, void* y, const float* w, const float* b, const ConvGeom& g, hipStream_t s)
{
    if (g.stride == 2) {
        switch (g.k) {
        case 7: return mult2_v<TX, TO, 7, 2>(x, y, w, b, g, s);
        case 5: return mult2_v<TX, TO, 5, 2>(x, y, w, b, g, s);
        case 3: return mult2_v<TX, TO, 3, 2>(x, y, w, b, g, s);
        default: return mult2_v<TX, TO, 0, 2>(x, y, w, b, g, s);
        }
    }
    switch (g.k) {
    case 7: return mult2_v<TX, TO, 7, 1>(x, y, w, b, g, s);
    case 5: return mult2_v<TX, TO, 5, 1>(x, y, w, b, g, s);
    case 3: return mult2_v<TX, TO, 3, 1>(x, y, w, b, g, s);
    default: return mult2_v<TX, TO, 0, 1>(x, y, w, b, g, s);
    }
}

// x: N x H x W x Cin ; y: N x Ho x Wo x (2*Cin) ; w packed (k,k,2*Cin) ; same dtype in and out
hipError_t generic_dwconv_mult2(const void* x, void* y, const float* w, const float* b,
                                int N, int Cin, int H, int W, int k, int stride, int dt, hipStream_t s)
{
    ConvGeom g{};
    g.N = N; g.C = 2 * Cin; g.Cin = Cin; g.H = H; g.W = W; g.k = k; g.stride = stride;
    const int p = k / 2;
    g.Ho = (H + 2 * p - k) / stride + 1;
    g.Wo = (W + 2 * p - k) / stride + 1;
    if (dt == 0) return mult2_ks<float, float>(x, y, w, b, g, s);
    if (dt == 2) return mult2_ks<f16_t, f16_t>(x, y, w, b, g, s);
    return mult2_ks<bf16_t, bf16_t>(x, y, w, b, g, s);
}

// ---- parameter packing ----
template <typename T>
__global__ void k_pack_dw_weight(const T* __restrict__ w, float* __restrict__ dst, int C, int kk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // index into dst (tap-major)
    if (i >= C * kk) return;
    const int tap = i / C, c = i % C;
    float v;
    v = elem_to_f32(w[(size_t)c * kk + tap]);
    dst[i] = v;
}

hipError_t pack_dw_weight(const void* w, float* dst, int C, int k, int dt, hipStream_t s)
{
    const int n = C * k * k;
    if (dt == 0) hipLaunchKernelGGL(k_pack_dw_weight<float>, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)w, dst, C, k * k);
    else if (dt == 2) hipLaunchKernelGGL(k_pack_dw_weight<f16_t>, dim3((n + 255) / 256), dim3(256), 0, s, (const f16_t*)w, dst, C, k * k);
    else hipLaunchKernelGGL(k_pack_dw_weight<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16_t*)w, dst, C, k * k);
    return hipGetLastError();
}

// every parameter of a RecConv2d block in ONE launch (a training step repacks them after each optimizer step): blockIdx.y = conv;
// also writes the pack with every k x k flipped (the backward's conv^T taps) and the float32 biases.  wflip / bpack may be null.
template <typename T>
__global__ void k_pack_params(PackPtrs P, float* __restrict__ wpack, float* __restrict__ wflip, float* __restrict__ bpack, int C, int kk)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * kk) return;
    const int tap = i / C, c = i % C;
    const float v = elem_to_f32(reinterpret_cast<const T*>(P.w[j])[(size_t)c * kk + tap]);
    wpack[(size_t)j * kk * C + i] = v;
    if (wflip) wflip[(size_t)j * kk * C + (size_t)(kk - 1 - tap) * C + c] = v;
    if (bpack && tap == 0) bpack[(size_t)j * C + c] = P.b[j] ? elem_to_f32(reinterpret_cast<const T*>(P.b[j])[c]) : 0.f;
}

hipError_t pack_params(const PackPtrs& P, float* wpack, float* wflip, float* bpack, int count, int C, int k, int dt, hipStream_t s)
{
    const int n = C * k * k;
    dim3 grid((n + 255) / 256, count), block(256);
    if (dt == 0) hipLaunchKernelGGL(k_pack_params<float>, grid, block, 0, s, P, wpack, wflip, bpack, C, k * k);
    else if (dt == 2) hipLaunchKernelGGL(k_pack_params<f16_t>, grid, block, 0, s, P, wpack, wflip, bpack, C, k * k);
    else hipLaunchKernelGGL(k_pack_params<bf16_t>, grid, block, 0, s, P, wpack, wflip, bpack, C, k * k);
    return hipGetLastError();
}


// the packed weight gradients (count, k*k, C) back into the parameters' layout, each (C, 1, k, k) float32 contiguous: one launch
__global__ void k_unpack_grads(const float* __restrict__ gwpack, PackPtrs P, int C, int kk)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // index into the destination (channel-major)
    if (i >= C * kk) return;
    const int c = i / kk, tap = i % kk;
    reinterpret_cast<float*>(const_cast<void*>(P.w[j]))[i] = gwpack[(size_t)j * kk * C + (size_t)tap * C + c];
}

hipError_t unpack_grads(const float* gwpack, const PackPtrs& P, int count, int C, int k, hipStream_t s)
{
    const int n = C * k * k;
    hipLaunchKernelGGL(k_unpack_grads, dim3((n + 255) / 256, count), dim3(256), 0, s, gwpack, P, C, k * k);
    return hipGetLastError();
}

hipError_t pack_bias(const void* b, float* dst, int C, int dt, hipStream_t s)
{
    // a bias is a (C,1,1,1) weight with one tap
    if (dt == 0) hipLaunchKernelGGL(k_pack_dw_weight<float>, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)b, dst, C, 1);
    else if (dt == 2) hipLaunchKernelGGL(k_pack_dw_weight<f16_t>, dim3((C + 255) / 256), dim3(256), 0, s, (const f16_t*)b, dst, C, 1);
    else hipLaunchKernelGGL(k_pack_dw_weight<bf16_t>, dim3((C + 255) / 256), dim3(256), 0, s, (const bf16_t*)b, dst, C, 1);
    return hipGetLastError();
}

// ---- start-up self-test (rcx_selftest_d16): the D16 "hi" loads must zero the other half of their destination on this device.  Registers
// preset to patterns; every lane compares what the three load forms leave with (element << 16) and ORs its verdict into *flag.
__global__ void k_selftest_d16(const uint16_t* __restrict__ src, unsigned* __restrict__ flag)
{
    __shared__ uint16_t sm[64];
    const int lane = threadIdx.x;
    sm[lane] = src[lane];
    __syncthreads();
    uint32_t a = 0xAAAAAAAAu, b = 0xBBBBBBBBu, c = 0xCCCCCCCCu;
    const uint16_t* p = src + lane;
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)(sm + lane);
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    i32x4_ rs;
    const unsigned long long base = (unsigned long long)src;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
    rs.z = 128;
    rs.w = 0x00020000;
    const unsigned voff = (unsigned)lane * 2u;
    asm volatile("s_nop 4\n\t"
                 "global_load_short_d16_hi %0, %3, off\n\t"
                 "buffer_load_short_d16_hi %1, %4, %5, 0 offen\n\t"
                 "ds_read_u16_d16_hi %2, %6\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(a), "+v"(b), "+v"(c) : "v"(p), "v"(voff), "s"(rs), "v"(la) : "memory");
    const uint32_t want = (uint32_t)sm[lane] << 16;
    const unsigned bad = (a != want ? 1u : 0u) | (b != want ? 2u : 0u) | (c != want ? 4u : 0u);
    if (bad) atomicOr(flag, bad);
}

hipError_t selftest_d16(const void* src, void* flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_d16, dim3(1), dim3(64), 0, s, (const uint16_t*)src, (unsigned*)flag);
    return hipGetLastError();
}

}  // namespace rcx
