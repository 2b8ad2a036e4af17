// RecNextStem in ONE launch (bf16 inference): y = conv3x3_s2(gelu(conv3x3_s2(x) + b1)) + b2, both convs BN-folded (model/recnext.py:134-146 `RecNextStem`,
// :75-97 ConvNorm.fuse).  As library calls the stem is two convs, two bias adds and a GELU (0.39 ms of RecNeXt-M3's 2.9 ms step): the intermediate
// 112 x 112 x C/2 tensor -- twice the size of the stem's output -- is written once and read three times.  Here it exists only as a 17 x 17 tile in LDS:
//   a workgroup (4 waves) owns an 8 x 8 tile of output pixels;
//   A. conv1 + bias + GELU for the 17 x 17 pixels of the intermediate the tile needs, ALSO on the matrix cores (K = 27 taps padded to 32; the im2col operand gathered
//      from a 35 x 35 x 3 tile of x in LDS: a pixel's inputs are three runs of nine contiguous bf16), rounded to bf16 into LDS (zeros outside the intermediate's plane:
//      the second conv's padding).  (A first version ran it on the vector pipe, a pixel per thread with the weights as scalar operands: 488 us, no faster than the library.)
//   B. conv2 as an implicit GEMM on the matrix cores: D (32 out channels x 32 pixels) = sum over (tap, 16-channel group) of W2 fragment x h1 fragment, the h1
//      fragment read straight from the LDS tile at the tap's offset (8 consecutive channels of one pixel = 16 bytes), the W2 fragments packed on the host
//      (ops.pack_stem); + b2 -> bf16 -> y.
// Numerics: float32 accumulation, the intermediate rounded to bf16 once (as the library path does), exact GELU (rcx_gelu.h).
#include "rcx_common.h"
#include "rcx_launch.h"
#include "rcx_gelu.h"

namespace rcx {
namespace stem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
typedef unsigned u32x2q __attribute__((ext_vector_type(2)));

// CM = channels of the intermediate, KC = CM rounded up to a multiple of 16 (the k-steps of a tap), M1 = ceil(CM / 32) tiles of the first product, MT = output tiles
template <int CM, int KC, int MT>
__global__ void __launch_bounds__(256)
k_stem(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const u32x4q* __restrict__ w1frag, const float* __restrict__ b1, const u32x4q* __restrict__ w2frag,
       const float* __restrict__ b2, int N, int H, int W, int H1, int W1, int H2, int W2, int CO, int tiles_x, int tiles_y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int KS = 9 * (KC / 16), NF = MT * KS, PIX = 2 * KC + 16, R = 17;      // k-steps, W2 fragments, bytes per pixel of the h1 tile (16 of padding: banks)
    constexpr int M1 = (CM + 31) / 32, XR = 35, XE = 105, XP = 216;                  // x tile: 35 rows of 35 pixels x 3 channels = 105 bf16 (210 bytes), pitch 216
    constexpr int NPT = (R * R + 31) / 32;                                           // pixel tiles of the first product; the h1 tile holds 32 NPT pixels, the x tile 37 rows: no guarded LDS writes
    const u32x4q* const Lf = reinterpret_cast<const u32x4q*>(lds_raw);               // [NF] W2 fragments, then [2 M1] W1 fragments
    const u32x4q* const Lf1 = Lf + NF * 64;
    unsigned char* const Lh = lds_raw + (size_t)(NF + 2 * M1) * 1024;                // h1 tile [17 x 17][PIX]
    unsigned char* const Lx = Lh + (size_t)32 * NPT * PIX;                           // x tile [37][XP]
    float* const Lb1 = reinterpret_cast<float*>(Lx + (size_t)(XR + 2) * XP);        // [32 M1] then b2 [32 MT]
    float* const Lb2 = Lb1 + 32 * M1;
    {                                                                                // the weights: once per workgroup (it walks tiles blockIdx.x, + gridDim.x, ...)
        u32x4q* Lw = reinterpret_cast<u32x4q*>(lds_raw);
        for (int i = threadIdx.x; i < NF * 64; i += 256) Lw[i] = w2frag[i];
        for (int i = threadIdx.x; i < 2 * M1 * 64; i += 256) Lw[NF * 64 + i] = w1frag[i];
        for (int i = threadIdx.x; i < 32 * M1; i += 256) Lb1[i] = b1[i];
        for (int i = threadIdx.x; i < 32 * MT; i += 256) Lb2[i] = b2[i];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned xbytes = (unsigned)H * (unsigned)W * 6u;                          // one image of x (3 channels, bf16); checked < 2^31 by the launcher
    // the x tile of a tile: rows 32 ty - 3 + i, pixels 32 tx - 3 + j of the image (zeros outside it: the first conv's padding); consecutive lanes = consecutive
    // elements; a thread's XN elements are requested a tile ahead (during the previous tile's second product) and written to LDS at the top of the tile
    constexpr int XN = (XR * XE + 255) / 256;
    static_assert(XN * 256 <= (XR + 2) * XE, "the two spare x-tile rows take the last round's overshoot");
    unsigned xgo[XN], xlo[XN], xij[XN];               // element u of this thread: offset within the image relative to the tile's first element, LDS offset, (row, pixel) of the tile
#pragma unroll
    for (int u = 0; u < XN; ++u) {
        const int e = threadIdx.x + 256 * u, i = e / XE, q = e - i * XE, px = q / 3;
        xgo[u] = ((unsigned)(i * W + px) * 3u + (unsigned)(q - 3 * px)) * 2u;
        xlo[u] = (unsigned)(i * XP + 2 * q);
        xij[u] = (unsigned)i | ((unsigned)px << 8) | (e < XR * XE ? 0u : 0x10000u);
    }
    bf16_t xq[XN];
    auto request_x = [&](int t) {
        const int nn = t / (tiles_x * tiles_y), trr = t - nn * tiles_x * tiles_y, tyy = trr / tiles_x, txx = trr - tyy * tiles_x;
        const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)nn * H * W * 3), 0, xbytes, 0x00020000);
        const int y0 = 32 * tyy - 3, x0 = 32 * txx - 3;
        const unsigned base = (unsigned)((y0 * W + x0) * 6);                       // (may wrap below zero: only added to offsets of elements inside the image)
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            const int yy = y0 + (int)(xij[u] & 0xff), xx = x0 + (int)((xij[u] >> 8) & 0xff);
            const bool ok = xij[u] < 0x10000u && yy >= 0 && yy < H && xx >= 0 && xx < W;
            xq[u] = (bf16_t)__builtin_amdgcn_raw_buffer_load_b16(xsrc, (int)(ok ? base + xgo[u] : 0x80000000u), 0, 0);      // outside the image: past the buffer, reads 0
        }
    };
    const int ntiles = N * tiles_x * tiles_y;
    if ((int)blockIdx.x < ntiles) request_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / (tiles_x * tiles_y), tr = tile - n * tiles_x * tiles_y, ty = tr / tiles_x, tx = tr - ty * tiles_x;
#pragma unroll
    for (int u = 0; u < XN; ++u) *reinterpret_cast<bf16_t*>(Lx + xlo[u]) = xq[u];
    __syncthreads();                                   // the x tile (and, the first time, the weights) is in LDS; every wave has left the previous tile's second product

    // ---- A. the intermediate's 17 x 17 pixels (rows 16 ty - 1 + i, columns 16 tx - 1 + j) on the matrix cores: D (32 channels x 32 pixels) = W1 (channel x 27 taps,
    // padded to 32) x im2col.  Pixel (i, j) reads x-tile rows 2 i + dy, elements 6 j .. 6 j + 8: its 27 inputs are three runs of nine contiguous bf16, k = 9 dy + e.
    unsigned koff[2][8];                                                            // byte offset of input k = 16 ks + 8 h + jj from the pixel's first element
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int kk = 16 * ks + 8 * h + jj, dy = (kk * 57) >> 9;
            koff[ks][jj] = kk < 27 ? (unsigned)(dy * XP + 2 * (kk - 9 * dy)) : 0u;         // (k = 27 .. 31: zero weights; any input of the pixel's own window will do)
        }
    for (int pt = wave; pt < NPT; pt += 4) {
        const int p = 32 * pt + r, pc = p < R * R ? p : R * R - 1, i = pc / R, j = pc - i * R;
        const int r1 = 16 * ty - 1 + i, c1 = 16 * tx - 1 + j;
        const bool inside = p < R * R && r1 >= 0 && r1 < H1 && c1 >= 0 && c1 < W1;
        const unsigned char* const xp = Lx + (2 * i) * XP + 12 * j;
        bf16x8 bfr[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned short v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = *reinterpret_cast<const unsigned short*>(xp + koff[ks][jj]);
            const u32x4q pk = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
            bfr[ks] = __builtin_bit_cast(bf16x8, pk);
        }
#pragma unroll
        for (int m1 = 0; m1 < M1; ++m1) {
            f32x16 d;
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] = 0.f;
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Lf1[(2 * m1 + 0) * 64 + lane]), bfr[0], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Lf1[(2 * m1 + 1) * 64 + lane]), bfr[1], d, 0, 0, 0);
            // + b1, gelu, bf16 -> the h1 tile (pixels outside the intermediate's plane: zeros, the second conv's padding; channels past CM: zero weights and bias)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * m1 + 8 * g + 4 * h;
                if (32 * m1 + 8 * g < KC) {
                    const f32x4q bb = *reinterpret_cast<const f32x4q*>(Lb1 + c0);
                    const gelu_f32x2 a = gelu2(gelu_f32x2{d[4 * g] + bb.x, d[4 * g + 1] + bb.y}), b = gelu2(gelu_f32x2{d[4 * g + 2] + bb.z, d[4 * g + 3] + bb.w});
                    bf16x4 o;
                    o[0] = (__bf16)(inside ? a.x : 0.f); o[1] = (__bf16)(inside ? a.y : 0.f); o[2] = (__bf16)(inside ? b.x : 0.f); o[3] = (__bf16)(inside ? b.y : 0.f);
                    *reinterpret_cast<u32x2q*>(Lh + (size_t)p * PIX + 2 * c0) = __builtin_bit_cast(u32x2q, o);
                }
            }
        }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) request_x(tile + gridDim.x);

    // ---- B. conv2: wave w takes the (output tile mt, pixel tile nt) pairs w, w + 4, ...; pixel q of the 8 x 8 tile = (q / 8, q % 8)
    for (int pr = wave; pr < 2 * MT; pr += 4) {
        const int mt = pr >> 1, nt = pr & 1;
        const int q = 32 * nt + r, py = q >> 3, px = q & 7;
        const unsigned char* const hp = Lh + (size_t)((2 * py) * R + 2 * px) * PIX + 16 * h;
        f32x16 d0, d1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            constexpr int KG = KC / 16;
            const int t = ks / KG, cg = ks - t * KG, dy = t / 3, dx = t - 3 * dy;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(hp + (size_t)(dy * R + dx) * PIX + 32 * cg);
            const bf16x8 a = __builtin_bit_cast(bf16x8, Lf[(mt * KS + ks) * 64 + lane]);
            if (ks & 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
            else d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
        }
        const int oy = 8 * ty + py, ox = 8 * tx + px;
        if (oy < H2 && ox < W2) {
            bf16_t* const yp = y + ((size_t)(n * H2 + oy) * W2 + ox) * CO;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * mt + 8 * g + 4 * h;
                if (c0 < CO) {
                    const f32x4q bb = *reinterpret_cast<const f32x4q*>(Lb2 + c0);
                    bf16x4 o;
                    o[0] = (__bf16)(d0[4 * g + 0] + d1[4 * g + 0] + bb.x); o[1] = (__bf16)(d0[4 * g + 1] + d1[4 * g + 1] + bb.y);
                    o[2] = (__bf16)(d0[4 * g + 2] + d1[4 * g + 2] + bb.z); o[3] = (__bf16)(d0[4 * g + 3] + d1[4 * g + 3] + bb.w);
                    *reinterpret_cast<u32x2q*>(yp + c0) = __builtin_bit_cast(u32x2q, o);
                }
            }
        }
    }
    }
}

}  // namespace stem

// CM (even, <= 40), CO (multiple of 4, <= 96), bf16; one image of x below 2^31 bytes
static bool stem_shape(int CM, int CO, int* kc, int* mt)
{
    if (CM <= 0 || CM > 40 || (CM & 1) || CO <= 0 || CO > 96 || CO % 4) return false;
    *kc = (CM + 15) / 16 * 16;
    *mt = (CO + 31) / 32;
    return CM == 20 || CM == 24 || CM == 28 || CM == 32 || CM == 40;
}

bool stem_applicable(int N, int H, int W, int CM, int CO, int dtype)
{
    int kc, mt;
    return dtype == 1 && N > 0 && H > 0 && W > 0 && (unsigned long long)H * W * 6 < (1ull << 31) && stem_shape(CM, CO, &kc, &mt);
}

size_t stem_pack_bytes(int CM, int CO)
{
    int kc, mt;
    if (!stem_shape(CM, CO, &kc, &mt)) return 0;
    return (size_t)mt * 9 * (kc / 16) * 1024;
}

template <int CM, int KC>
static hipError_t launch_stem(const void* x, void* y, const void* w1, const float* b1, const void* w2frag, const float* b2, int N, int H, int W, int CO, int MT, int ncu, hipStream_t s)
{
    const int H1 = (H + 1) / 2, W1 = (W + 1) / 2, H2 = (H1 + 1) / 2, W2 = (W1 + 1) / 2, tx = (W2 + 7) / 8, ty = (H2 + 7) / 8;
    constexpr int KS = 9 * (KC / 16), PIX = 2 * KC + 16;
    constexpr int M1 = (CM + 31) / 32;
    const size_t lds = (size_t)(MT * KS + 2 * M1) * 1024 + (size_t)320 * PIX + (size_t)37 * 216 + sizeof(float) * 32 * (M1 + MT);
    const long long ntiles = (long long)N * tx * ty;
    if (ntiles > 0x7fffffffLL || lds > 160 * 1024) return hipErrorInvalidConfiguration;
    const long long per_cu = (long long)(160 * 1024 / lds) > 0 ? (long long)(160 * 1024 / lds) : 1;
    const long long grid = ntiles < ncu * per_cu ? ntiles : ncu * per_cu;          // persistent: the weights go to LDS once per workgroup
#define RCX_STEM_GO(MT_)                                                                                                                   \
    {                                                                                                                                      \
        auto kfn = stem::k_stem<CM, KC, MT_>;                                                                                              \
        RCX_SET_LDS_ONCE(kfn, lds);                                                                                                        \
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, (const stem::u32x4q*)w1, b1, (const stem::u32x4q*)w2frag, b2, \
                           N, H, W, H1, W1, H2, W2, CO, tx, ty);                                                                          \
        return hipGetLastError();                                                                                                          \
    }
    if (MT == 1) RCX_STEM_GO(1)
    if (MT == 2) RCX_STEM_GO(2)
    if (MT == 3) RCX_STEM_GO(3)
#undef RCX_STEM_GO
    return hipErrorInvalidConfiguration;
}

hipError_t stem_fwd(const void* x, void* y, const void* w1, const float* b1, const void* w2frag, const float* b2, int N, int H, int W, int CM, int CO, int dtype, hipStream_t s)
{
    int kc, mt;
    if (!stem_applicable(N, H, W, CM, CO, dtype) || !stem_shape(CM, CO, &kc, &mt)) return hipErrorInvalidConfiguration;
    static std::atomic<int> cus[RCX_MAX_DEVICES];                     // compute units of each device, asked once
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < RCX_MAX_DEVICES) {
        ncu = cus[dev].load(std::memory_order_relaxed);
        if (ncu <= 0) {
            int v = 0;
            ncu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
            cus[dev].store(ncu, std::memory_order_relaxed);
        }
    }
    switch (CM) {
        case 20: return launch_stem<20, 32>(x, y, w1, b1, w2frag, b2, N, H, W, CO, mt, ncu, s);
        case 24: return launch_stem<24, 32>(x, y, w1, b1, w2frag, b2, N, H, W, CO, mt, ncu, s);
        case 28: return launch_stem<28, 32>(x, y, w1, b1, w2frag, b2, N, H, W, CO, mt, ncu, s);
        case 32: return launch_stem<32, 32>(x, y, w1, b1, w2frag, b2, N, H, W, CO, mt, ncu, s);
        case 40: return launch_stem<40, 48>(x, y, w1, b1, w2frag, b2, N, H, W, CO, mt, ncu, s);
        default: return hipErrorInvalidConfiguration;
    }
}

}  // namespace rcx
