// The channel mixer of a MetaNeXtBlock and the residual add around it in ONE launch (inference, bf16 activations):
//     y = x + W2 gelu(W1 z + b1) + b2          (model/recnext.py:125-132 `mlp`, :157-158 `x + drop_path(channel_mixer(norm(token_mixer(x))))`, :169-171 Downsample;
//                                               model/recattn.py:171, :184)
// z = the token mixer's output (its BatchNorm already folded into the mixer's last conv), x = the block's input, W1 (H x C) / W2 (C x H) the two BN-folded 1x1
// convs.  As four library launches (GEMM + bias, GELU, GEMM + bias, add) the hidden tensor -- 2 x the size of x -- is written and read twice and y once more:
// 7 S of traffic where x, z in and y out are 3 S; at the 56 x 56 and 28 x 28 stages (C = 64 / 128) these GEMMs are memory-bound, so that is most of the cost.
//
// ONE WAVE = 32 tokens at a time, the hidden layer streamed through its registers 32 units at a time, never in memory:
//   D1  (32 hidden units x 32 tokens) = W1[32 ht ..][:] z^T      v_mfma_f32_32x32x16_bf16: A = a W1 fragment (LDS), B = the tokens' channels (registers, loaded
//                                                               once per tile straight from memory: 16 bytes per lane and k-step)
//   h   = 2 gelu(D1 + b1) in float32, rounded to bf16 (W2 is packed halved: exact)   the accumulator layout (token on the lane, units in the registers) IS the B operand of ...
//   D2 += W2[:][32 ht ..] h                                      ... the second product, with W2's columns stored in the order the registers imply
//   y   = D2 + b2 + x                                            8-byte loads / stores of four channels per lane (token on the lane)
// The weights are ready-made fragments (one conflict-free 16-byte LDS read per lane and product), packed once on the host (ops.pack_channel_mlp), hidden tile by
// hidden tile: resident in LDS for C <= 128 (k_channel_mlp), streamed through a two-slot LDS ring for C = 128 .. 320 (k_channel_mlp_stream, further down).  Global
// accesses are whole-wave contiguous kilobytes, transposed to / from the token-on-lane layouts in per-wave LDS images (measurements: profiles/r05_channel_mlp.txt).
// GELU is the exact form 0.5 v (1 + erf(v / sqrt 2)) with erf as an odd degree-15 polynomial of the argument clamped to +-2.8: |error| < 7.7e-5 in erf, i.e. 4e-5 |v|
// in gelu -- a fiftieth of a bf16 ulp; the library's erff would be most of this kernel's vector work.
#include "rcx_common.h"
#include "rcx_launch.h"
#include "rcx_opts.h"
#include "rcx_gelu.h"

namespace rcx {
namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef float f32x2q __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
typedef unsigned u32x2q __attribute__((ext_vector_type(2)));

// LDS operations of one wave execute in order; this only keeps the compiler from moving accesses across a hand-off within the wave
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (row of accumulator register i in lane half h of a 32 x 32 tile: (i & 3) + 8 (i >> 2) + 4 h -- ops.py::_mlp_acc_unit orders W2's columns by it)

// One 32-unit tile of the hidden layer from its chunk of fragments in LDS (Lc: KS1 W1 fragments, then 2 CT W2 fragments in (ct, q) order):
//   D1 = W1 tile x z^T, h = gelu(D1 + b1) -> bf16 (already the second product's B operand), D2 += W2 columns x h.
// The fragments go through register rings RD deep, requested RD products ahead -- and the second product's first RD before the GELU: a read issued right in
// front of its product costs the LDS latency per product (measured: 13 % of the matrix-core peak whatever the shape).
template <int KS1, int CT, int RD>
__device__ __forceinline__ void hidden_tile_ring(const u32x4q* Lc, const float* b1t, int lane, int h, const bf16x8 (&zb)[KS1], f32x16 (&d2)[CT])
{
    constexpr int R1 = KS1 < RD ? KS1 : RD, N2 = 2 * CT, R2 = N2 < RD ? N2 : RD;
    u32x4q ring[RD];
#pragma unroll
    for (int j = 0; j < R1; ++j) ring[j] = Lc[j * 64 + lane];
    f32x16 d1;
#pragma unroll
    for (int i = 0; i < 16; ++i) d1[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[ks % R1]);
        if (ks + R1 < KS1) ring[ks % R1] = Lc[(ks + R1) * 64 + lane];
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zb[ks], d1, 0, 0, 0);
    }
    // the second product's fragments in use order: f = q CT + ct  ->  pack slot KS1 + 2 ct + q
#pragma unroll
    for (int f = 0; f < R2; ++f) ring[f] = Lc[(KS1 + 2 * (f % CT) + f / CT) * 64 + lane];
    bf16x8 hb[2];
    gelu_f32x2 gv[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4q bb = *reinterpret_cast<const f32x4q*>(b1t + 8 * g + 4 * h);
        gv[2 * g] = gelu_f32x2{d1[4 * g] + bb.x, d1[4 * g + 1] + bb.y};
        gv[2 * g + 1] = gelu_f32x2{d1[4 * g + 2] + bb.z, d1[4 * g + 3] + bb.w};
    }
    {                                                 // the tile's 16 values, four pairs in lockstep at a time (eight: the kernels at their register limit spill); TWICE
                                                      // the GELU -- W2 is stored halved (ops.pack_channel_mlp)
        gelu_f32x2 ga[4] = {gv[0], gv[1], gv[2], gv[3]}, gb[4] = {gv[4], gv[5], gv[6], gv[7]};
        gelu2x_batch<4>(ga);
        gelu2x_batch<4>(gb);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gv[i] = ga[i]; gv[4 + i] = gb[i]; }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        hb[g >> 1][4 * (g & 1) + 0] = (__bf16)gv[2 * g].x; hb[g >> 1][4 * (g & 1) + 1] = (__bf16)gv[2 * g].y;
        hb[g >> 1][4 * (g & 1) + 2] = (__bf16)gv[2 * g + 1].x; hb[g >> 1][4 * (g & 1) + 3] = (__bf16)gv[2 * g + 1].y;
    }
#pragma unroll
    for (int f = 0; f < N2; ++f) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[f % R2]);
        if (f + R2 < N2) ring[f % R2] = Lc[(KS1 + 2 * ((f + R2) % CT) + (f + R2) / CT) * 64 + lane];
        d2[f % CT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[f / CT], d2[f % CT], 0, 0, 0);
    }
}

// LDS: per hidden tile ht its KS1 W1 fragments and 2 CT W2 fragments (1 KB each: 64 lanes x 16 bytes; ops.pack_channel_mlp), then b1 (32 HT floats), b2 (32 CT floats)

// KS1 = ceil(C / 16) k-steps of the first product, HT = H / 32 hidden tiles, CT = ceil(C / 32) output tiles; C % 8 == 0 (a lane's 8 channels of a k-step are
// all there or all padding).  NW waves per workgroup share the LDS weights; each takes every (grid x NW)-th 32-token tile.
// KX / OX: C == 16 KS1 / C == 32 CT exactly -- no padding lanes, so a lane's offsets are one register plus immediates
template <int KS1, int HT, int CT, int NW, int WPS, bool KX, bool OX, bool STAGED>
__global__ void __launch_bounds__(64 * NW, WPS)
k_channel_mlp(const bf16_t* __restrict__ z, const bf16_t* __restrict__ xres, bf16_t* __restrict__ y, const u32x4q* __restrict__ wfrag, const float* __restrict__ bias,
              int M, int C, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NF = HT * KS1 + CT * 2 * HT;
    const u32x4q* const Lf = reinterpret_cast<const u32x4q*>(lds_raw);
    const float* const Lb1 = reinterpret_cast<const float*>(lds_raw + (size_t)NF * 1024);
    const float* const Lb2 = Lb1 + 32 * HT;
    {
        u32x4q* Lw = reinterpret_cast<u32x4q*>(lds_raw);
        for (int i = threadIdx.x; i < NF * 64; i += 64 * NW) Lw[i] = wfrag[i];
        float* Lb = reinterpret_cast<float*>(lds_raw + (size_t)NF * 1024);
        for (int i = threadIdx.x; i < 32 * (HT + CT); i += 64 * NW) Lb[i] = bias[i];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned nbytes = (unsigned)M * (unsigned)C * 2u;                  // < 2^31 (checked by the launcher): an offset with bit 31 set is out of range whatever the token
    const __amdgpu_buffer_rsrc_t zsrc = __builtin_amdgcn_make_buffer_rsrc((void*)z, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xres, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, nbytes, 0x00020000);
    // the lane's part of its offsets: 8 channels of k-step ks (reads of z), 4 channels of output group (ct, g) (reads of x, stores of y); padding: bit 31
    unsigned zk[KS1], oc[CT][4];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) zk[ks] = KX || 16 * ks + 8 * h < C ? 2u * (16 * ks + 8 * h) : 0x80000000u;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) oc[ct][g] = OX || 32 * ct + 8 * g + 4 * h < C ? 2u * (32 * ct + 8 * g + 4 * h) : 0x80000000u;

    auto hidden_tile = [&](int ht, const bf16x8 (&zb)[KS1], f32x16 (&d2)[CT]) {
        hidden_tile_ring<KS1, CT, 4>(Lf + (size_t)ht * (KS1 + 2 * CT) * 64, Lb1 + 32 * ht, lane, h, zb, d2);
    };
    const int stride = gridDim.x * NW;
    int tile = blockIdx.x * NW + wave;
    if constexpr (STAGED) {
        // ---- two output tiles or fewer (C <= 64): every global access is a whole-wave contiguous kilobyte, the token-on-lane layouts the products need are made in
        // LDS.  (Reading z / x and writing y a token per lane costs the CU's address path 64 cycles a request -- 32 lines touched, 16 or 8 bytes each: 77 us of the
        // 114 us this kernel took that way at 256 x 64 x 56 x 56, profiles/r05_channel_mlp.txt.)  A wave's tile: 32 tokens = 32 RB contiguous bytes.
        constexpr int ZP = 32 * KS1 + 16, OP = 128 * CT + 16;                 // bytes per token row: bf16 z image, float32 output image (16 bytes of padding: banks)
        constexpr int WREG = 32 * (ZP > OP ? ZP : OP);
        unsigned char* const Lt = lds_raw + (size_t)NF * 1024 + sizeof(float) * 32 * (HT + CT) + (size_t)wave * WREG;
        const unsigned RB = 2u * (unsigned)C;
        unsigned go[KS1], za[KS1], oa[KS1];                                   // request i of a tile: bytes [1024 i + 16 lane, + 16) of it -> token t, byte b of its row
        bool live[KS1];
#pragma unroll
        for (int i = 0; i < KS1; ++i) {
            const unsigned o = 1024u * i + 16u * lane, t = o / RB, bb = o - t * RB;
            live[i] = o < 32u * RB;
            go[i] = live[i] ? o : 0x80000000u;
            za[i] = t * ZP + bb;
            oa[i] = t * OP + 2u * bb;
        }
        u32x4q zq[KS1];
        auto load_z = [&](int t) {
#pragma unroll
            for (int i = 0; i < KS1; ++i) zq[i] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(zsrc, (int)((unsigned)t * 32u * RB + go[i]), 0, 0));
        };
        if (tile < ntiles) load_z(tile);
        for (; tile < ntiles; tile += stride) {
            const unsigned base = (unsigned)tile * 32u * RB;
#pragma unroll
            for (int i = 0; i < KS1; ++i)
                if (live[i]) *reinterpret_cast<u32x4q*>(Lt + za[i]) = zq[i];
            wave_sync();
            bf16x8 zb[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                u32x4q v = *reinterpret_cast<const u32x4q*>(Lt + r * ZP + 32 * ks + 16 * h);
                if (!KX && ks == KS1 - 1 && !(16 * ks + 8 * h < C)) v = u32x4q{0u, 0u, 0u, 0u};      // channels past C: whatever the image held last (their weights are zeros, but 0 x NaN is not)
                zb[ks] = __builtin_bit_cast(bf16x8, v);
            }
            wave_sync();
            if (tile + stride < ntiles) load_z(tile + stride);                // the next tile and this tile's residual: in flight during the products
            u32x4q xq[KS1];
#pragma unroll
            for (int i = 0; i < KS1; ++i) xq[i] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(xsrc, (int)(base + go[i]), 0, 0));
            f32x16 d2[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) d2[ct][i] = 0.f;
#pragma unroll 1
            for (int ht = 0; ht < HT; ++ht) hidden_tile(ht, zb, d2);
            // D2 + b2 (token on the lane) -> the float32 image -> rows
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4q bb = *reinterpret_cast<const f32x4q*>(Lb2 + 32 * ct + 8 * g + 4 * h);
                    *reinterpret_cast<f32x4q*>(Lt + r * OP + 4 * (32 * ct + 8 * g + 4 * h)) =
                        f32x4q{d2[ct][4 * g] + bb.x, d2[ct][4 * g + 1] + bb.y, d2[ct][4 * g + 2] + bb.z, d2[ct][4 * g + 3] + bb.w};
                }
            wave_sync();
#pragma unroll
            for (int i = 0; i < KS1; ++i) {
                if (!live[i]) continue;
                const f32x4q lo = *reinterpret_cast<const f32x4q*>(Lt + oa[i]), hi = *reinterpret_cast<const f32x4q*>(Lt + oa[i] + 16);
                const u32x4q xv = xq[i];
                bf16x8 o;
                o[0] = (__bf16)(lo.x + __uint_as_float(xv.x << 16)); o[1] = (__bf16)(lo.y + __uint_as_float(xv.x & 0xffff0000u));
                o[2] = (__bf16)(lo.z + __uint_as_float(xv.y << 16)); o[3] = (__bf16)(lo.w + __uint_as_float(xv.y & 0xffff0000u));
                o[4] = (__bf16)(hi.x + __uint_as_float(xv.z << 16)); o[5] = (__bf16)(hi.y + __uint_as_float(xv.z & 0xffff0000u));
                o[6] = (__bf16)(hi.z + __uint_as_float(xv.w << 16)); o[7] = (__bf16)(hi.w + __uint_as_float(xv.w & 0xffff0000u));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, o), ysrc, (int)(base + go[i]), 0, 0);
            }
            wave_sync();
        }
        return;
    }
    // ---- three output tiles or more (C = 80, 96; C = 128 with RCX_MLP_STREAM=0): the weights take the LDS the images would need; a token per lane straight from / to memory
    u32x4q zf[KS1];
    auto load_z = [&](int t) {
        const unsigned row = (unsigned)(32 * t + r) * (unsigned)C * 2u;       // (a token past M: past the buffer, reads 0)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) zf[ks] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(zsrc, (int)(row + zk[ks]), 0, 0));
    };
    if (tile < ntiles) load_z(tile);
    for (; tile < ntiles; tile += stride) {
        const unsigned row = (unsigned)(32 * tile + r) * (unsigned)C * 2u;
        bf16x8 zb[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) zb[ks] = __builtin_bit_cast(bf16x8, zf[ks]);
        if (tile + stride < ntiles) load_z(tile + stride);                    // the next tile's channels: in flight during this tile's products
        f32x16 d2[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) d2[ct][i] = 0.f;
#pragma unroll 1
        for (int ht = 0; ht < HT; ++ht) hidden_tile(ht, zb, d2);
        u32x2q xr[CT][4];                                                    // the residual (requested after the products: until here its registers are the accumulators')
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) xr[ct][g] = __builtin_bit_cast(u32x2q, __builtin_amdgcn_raw_buffer_load_b64(xsrc, (int)(row + oc[ct][g]), 0, 0));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4q bb = *reinterpret_cast<const f32x4q*>(Lb2 + 32 * ct + 8 * g + 4 * h);
                const unsigned x0 = xr[ct][g].x, x1 = xr[ct][g].y;
                bf16x4 o;
                o[0] = (__bf16)(d2[ct][4 * g + 0] + bb.x + __uint_as_float(x0 << 16));
                o[1] = (__bf16)(d2[ct][4 * g + 1] + bb.y + __uint_as_float(x0 & 0xffff0000u));
                o[2] = (__bf16)(d2[ct][4 * g + 2] + bb.z + __uint_as_float(x1 << 16));
                o[3] = (__bf16)(d2[ct][4 * g + 3] + bb.w + __uint_as_float(x1 & 0xffff0000u));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2q, o), ysrc, (int)(row + oc[ct][g]), 0, 0);
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same block for channel counts whose weights do not fit the LDS (C = 256, H = 512: 512 KB): the hidden-tile chunks of the pack (KS1 + 2 CT fragments, 32 KB at
// C = 256) are STREAMED through a two-deep LDS ring while the products of the previous chunk run -- every workgroup walks the same chunks in the same order, so they
// come from the L2.  NW waves per workgroup, 32 tokens each, hold their tokens' channels (B fragments, 4 KS1 registers) and the output accumulators (16 CT) for the
// whole block; one barrier per hidden tile.  With NW = 4 (one wave per SIMD) a wave has 512 registers: the accumulators sit in the AGPRs.
// CT: output tiles rounded up to even (the pack pads W2 / b2 with zero rows); ZH: the z image is staged in ZH column parts (2: 320 channels, whose whole rows would
// not leave room for the ring); ZPF: request the next block's z during this block (off where its registers are needed)
template <int KS1, int HT, int CT, int NW, int WPS, int ZH, bool ZPF>
__global__ void __launch_bounds__(64 * NW, WPS)
k_channel_mlp_stream(const bf16_t* __restrict__ z, const bf16_t* __restrict__ xres, bf16_t* __restrict__ y, const u32x4q* __restrict__ wfrag, const float* __restrict__ bias,
                     int M, int C, int nblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NCH = KS1 + 2 * CT, NT = 64 * NW, PER = (NCH * 64 + NT - 1) / NT;          // fragments per chunk; 16-byte pieces of a chunk per thread (the last one ragged)
    constexpr bool RAG = NCH * 64 % NT != 0;
    u32x4q* const Lring = reinterpret_cast<u32x4q*>(lds_raw);                        // [2][NCH * 64]
    float* const Lb1 = reinterpret_cast<float*>(lds_raw + (size_t)2 * NCH * 1024);
    float* const Lb2 = Lb1 + 32 * HT;
    for (int i = threadIdx.x; i < 32 * (HT + CT); i += NT) Lb1[i] = bias[i];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned nbytes = (unsigned)M * (unsigned)C * 2u;
    const __amdgpu_buffer_rsrc_t zsrc = __builtin_amdgcn_make_buffer_rsrc((void*)z, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xres, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, nbytes, 0x00020000);
    // chunk 0 -> ring slot 0; chunk 1 -> the registers of set 1.  In step ht (chunk ht in slot ht & 1) the requests for chunk ht + 2 go out first and land in
    // register set ht & 1, the products run, then set (ht + 1) & 1 -- chunk ht + 1, requested a whole step earlier -- is written to the other slot: the L2 round
    // trip (1.5 - 2 us under this load, longer than a step's products) has two steps to complete.  The chunk sequence is periodic, so it runs across blocks.
    static_assert(HT % 2 == 0, "the ring alternates two register sets");
    u32x4q cp[2][PER];
    auto request = [&](u32x4q (&dst)[PER], int chunk) {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (!RAG || i + 1 < PER || threadIdx.x + i * NT < NCH * 64) dst[i] = wfrag[(size_t)chunk * NCH * 64 + threadIdx.x + i * NT];
    };
    auto deposit = [&](const u32x4q (&src)[PER], int slot) {
        u32x4q* const Ln = Lring + slot * NCH * 64;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (!RAG || i + 1 < PER || threadIdx.x + i * NT < NCH * 64) Ln[threadIdx.x + i * NT] = src[i];
    };
    request(cp[0], 0);
    deposit(cp[0], 0);
    request(cp[1], 1 % HT);
    __syncthreads();
    // Global accesses: whole-wave contiguous kilobytes (as the small shapes' STAGED path).  A wave's 32 tokens are 32 RB = 1024 KS1 contiguous bytes: request i =
    // bytes [1024 i + 16 lane, + 16) -> token t, byte bz of its row in the wave's bf16 z image.  The output leaves in halves of 64 channels (two output tiles): a
    // float32 image of 32 x 64, read back as rows -- request j of half hf = the 128-byte pieces [128 hf, + 128) of 8 token rows.
    static_assert(CT % 2 == 0, "output in halves of two tiles");
    static_assert(KS1 % ZH == 0, "column parts of whole k-steps");
    constexpr int RB = 32 * KS1, RBH = RB / ZH, KH = KS1 / ZH, ZP = RBH + 16, OP = 256 + 16, IMG = 32 * (ZP > OP ? ZP : OP), NH = CT / 2;
    unsigned char* const Lt = lds_raw + (size_t)2 * NCH * 1024 + sizeof(float) * 32 * (HT + CT) + (size_t)wave * IMG;
    unsigned zo[KS1], za[KS1];                                                  // request i of column part q = i / KH: piece 1024 (i % KH) + 16 lane of that part
#pragma unroll
    for (int i = 0; i < KS1; ++i) {
        const unsigned o = 1024u * (i % KH) + 16u * lane, t = o / RBH, bz = o % RBH;
        zo[i] = t * RB + (i / KH) * RBH + bz;
        za[i] = t * ZP + bz;
    }
    // half hf, request j (4 per half): token t = 8 j + lane / 8, bytes 128 hf + 16 (lane % 8) of its row
    const unsigned ht_tok = lane >> 3, ht_b = 16u * (lane & 7);
    u32x4q zq[KS1];
    auto load_z = [&](int blk) {
        const unsigned base = (unsigned)(32 * (blk * NW + wave)) * (unsigned)RB;      // (tokens past M: past the buffer -- reads 0, stores dropped)
#pragma unroll
        for (int i = 0; i < KS1; ++i) zq[i] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(zsrc, (int)(base + zo[i]), 0, 0));
    };
    if (ZPF && (int)blockIdx.x < nblocks) load_z(blockIdx.x);
    for (int block = blockIdx.x; block < nblocks; block += gridDim.x) {
        const unsigned base = (unsigned)(32 * (block * NW + wave)) * (unsigned)RB;
        if constexpr (!ZPF) load_z(block);
        bf16x8 zb[KS1];
#pragma unroll
        for (int q = 0; q < ZH; ++q) {
#pragma unroll
            for (int i = 0; i < KH; ++i) *reinterpret_cast<u32x4q*>(Lt + za[q * KH + i]) = zq[q * KH + i];
            wave_sync();
#pragma unroll
            for (int k = 0; k < KH; ++k) zb[q * KH + k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4q*>(Lt + r * ZP + 32 * k + 16 * h));
            wave_sync();
        }
        if (ZPF && block + (int)gridDim.x < nblocks) load_z(block + gridDim.x);    // the next block's channels: in flight during this block
        f32x16 d2[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) d2[ct][i] = 0.f;
        u32x4q xq[NH][4];                                                          // the residual: requested in the last step but one
#pragma unroll 1
        for (int ht = 0; ht < HT; ht += 2) {
            request(cp[0], (ht + 2) % HT);
            if (ht == HT - 2) {
#pragma unroll
                for (int hf = 0; hf < NH; ++hf)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xq[hf][j] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(xsrc, (int)(128u * hf + ht_b < RB ? base + (8u * j + ht_tok) * RB + 128u * hf + ht_b : 0x80000000u), 0, 0));
            }
            hidden_tile_ring<KS1, CT, (NW > 4 || KS1 > 16 ? 4 : 8)>(Lring, Lb1 + 32 * ht, lane, h, zb, d2);                               // chunk ht: slot 0
            deposit(cp[1], 1);                                                                                  // chunk ht + 1
            __syncthreads();                      // every wave is done with slot 0; slot 1 is complete
            request(cp[1], (ht + 3) % HT);
            hidden_tile_ring<KS1, CT, (NW > 4 || KS1 > 16 ? 4 : 8)>(Lring + NCH * 64, Lb1 + 32 * (ht + 1), lane, h, zb, d2);              // chunk ht + 1: slot 1
            deposit(cp[0], 0);                                                                                  // chunk ht + 2 (the next block's chunk 0 after the last)
            __syncthreads();
        }
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ct = 2 * hf + c2;
                    const f32x4q bb = *reinterpret_cast<const f32x4q*>(Lb2 + 32 * ct + 8 * g + 4 * h);
                    *reinterpret_cast<f32x4q*>(Lt + r * OP + 4 * (32 * c2 + 8 * g + 4 * h)) =
                        f32x4q{d2[ct][4 * g] + bb.x, d2[ct][4 * g + 1] + bb.y, d2[ct][4 * g + 2] + bb.z, d2[ct][4 * g + 3] + bb.w};
                }
            wave_sync();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char* src = Lt + (8 * j + ht_tok) * OP + 2 * ht_b;
                const f32x4q lo = *reinterpret_cast<const f32x4q*>(src), hi = *reinterpret_cast<const f32x4q*>(src + 16);
                const u32x4q xv = xq[hf][j];
                bf16x8 o;
                o[0] = (__bf16)(lo.x + __uint_as_float(xv.x << 16)); o[1] = (__bf16)(lo.y + __uint_as_float(xv.x & 0xffff0000u));
                o[2] = (__bf16)(lo.z + __uint_as_float(xv.y << 16)); o[3] = (__bf16)(lo.w + __uint_as_float(xv.y & 0xffff0000u));
                o[4] = (__bf16)(hi.x + __uint_as_float(xv.z << 16)); o[5] = (__bf16)(hi.y + __uint_as_float(xv.z & 0xffff0000u));
                o[6] = (__bf16)(hi.z + __uint_as_float(xv.w << 16)); o[7] = (__bf16)(hi.w + __uint_as_float(xv.w & 0xffff0000u));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, o), ysrc, (int)(128u * hf + ht_b < RB ? base + (8u * j + ht_tok) * RB + 128u * hf + ht_b : 0x80000000u), 0, 0);   // (a half past the row's channels: dropped)
            }
            wave_sync();
        }
    }
}

}  // namespace mlp

// C % 8 == 0, H % 32 == 0 (the host pads the hidden layer with zero units), the weights fit the CU's LDS, M C 2 < 2^31
static bool mlp_shape(int C, int H, int* ks1, int* ht, int* ct)
{
    if (C <= 0 || H <= 0 || C % 8 || H % 32) return false;
    *ks1 = (C + 15) / 16; *ht = H / 32; *ct = (C + 31) / 32;
    if (C > 128 && (*ct & 1)) ++*ct;              // the streamed kernels write their output in halves of two tiles: W2 / b2 padded with zero rows (ops.pack_channel_mlp)
    return true;
}

bool channel_mlp_applicable(int M, int C, int H, int dtype)
{
    int ks1, ht, ct;
    if (dtype != 1 || M <= 0 || !mlp_shape(C, H, &ks1, &ht, &ct)) return false;
    if ((unsigned long long)M * C * 2 >= (1ull << 31)) return false;
    if (C == 256 && ht == 16) return true;
    if (C == 192 && ht == 12) return true;
    if ((C == 160 && ht == 10) || (C == 320 && ht == 20)) return true;
    return (ks1 == 4 && ht == 4 && ct == 2) || (C == 128 && ht == 8) || (ks1 == 3 && ht == 3 && ct == 2) || (C == 96 && ht == 6) || (C == 80 && ht == 5);
}

size_t channel_mlp_pack_bytes(int C, int H)
{
    int ks1, ht, ct;
    if (!mlp_shape(C, H, &ks1, &ht, &ct)) return 0;
    return (size_t)(ht * ks1 + ct * 2 * ht) * 1024;
}

template <int KS1, int HT, int CT, bool KX, bool OX>
static hipError_t launch_mlp(const void* z, const void* x, void* y, const void* wfrag, const float* bias, int M, int C, int ncu, hipStream_t s)
{
    constexpr size_t lds = (size_t)(HT * KS1 + CT * 2 * HT) * 1024 + sizeof(float) * 32 * (HT + CT);
    static_assert(lds <= 160 * 1024, "the weights must fit the LDS");
    // one workgroup per CU shares the LDS weights: 12 waves (three per SIMD, 168 registers) where the accumulators are two output tiles, else 8 (256 registers)
    constexpr int NW = CT <= 2 ? 12 : 8, WPS = NW / 4;
    constexpr bool STAGED = CT <= 2;
    constexpr size_t stage = STAGED ? (size_t)NW * 32 * ((32 * KS1 + 16) > (128 * CT + 16) ? (32 * KS1 + 16) : (128 * CT + 16)) : 0;
    if ((C == 16 * KS1) != KX || (C == 32 * CT) != OX) return hipErrorInvalidConfiguration;
    auto kfn = mlp::k_channel_mlp<KS1, HT, CT, NW, WPS, KX, OX, STAGED>;
    static_assert(lds + stage <= 160 * 1024, "weights + the waves' images must fit the LDS");
    RCX_SET_LDS_ONCE(kfn, lds + stage);
    const int ntiles = (M + 31) / 32;
    int grid = ncu;
    if (grid * NW > ntiles) grid = (ntiles + NW - 1) / NW;
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(64 * NW), lds + stage, s, (const bf16_t*)z, (const bf16_t*)x, (bf16_t*)y, (const mlp::u32x4q*)wfrag, bias, M, C, ntiles);
    return hipGetLastError();
}

// C == 16 KS1 == 32 CT exactly (no padding lanes): the streamed form
template <int KS1, int HT, int CT, int NW, int ZH = 1, bool ZPF = true>
static hipError_t launch_mlp_stream(const void* z, const void* x, void* y, const void* wfrag, const float* bias, int M, int C, int ncu, hipStream_t s)
{
    constexpr int NCH = KS1 + 2 * CT, ZP = 32 * KS1 / ZH + 16, OP = 256 + 16, IMG = 32 * (ZP > OP ? ZP : OP);
    constexpr size_t lds = (size_t)2 * NCH * 1024 + sizeof(float) * 32 * (HT + CT) + (size_t)NW * IMG;
    static_assert(lds <= 160 * 1024, "the ring and the waves' images must fit the LDS");
    static_assert(CT % 2 == 0 && 32 * CT >= 16 * KS1 && 32 * (CT - 2) < 16 * KS1, "output tiles: the channel count rounded up to a multiple of 64");
    if (C != 16 * KS1) return hipErrorInvalidConfiguration;
    auto kfn = mlp::k_channel_mlp_stream<KS1, HT, CT, NW, NW / 4, ZH, ZPF>;
    RCX_SET_LDS_ONCE(kfn, lds);
    const int nblocks = (M + 32 * NW - 1) / (32 * NW);
    const int grid = nblocks < ncu ? nblocks : ncu;
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(64 * NW), lds, s, (const bf16_t*)z, (const bf16_t*)x, (bf16_t*)y, (const mlp::u32x4q*)wfrag, bias, M, C, nblocks);
    return hipGetLastError();
}

hipError_t channel_mlp(const void* z, const void* x, void* y, const void* wfrag, const float* bias, int M, int C, int H, int dtype, hipStream_t s)
{
    int ks1, ht, ct;
    if (!channel_mlp_applicable(M, C, H, dtype) || !mlp_shape(C, H, &ks1, &ht, &ct)) return hipErrorInvalidConfiguration;
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
    if (C == 256 && ht == 16) return launch_mlp_stream<16, 16, 8, 4>(z, x, y, wfrag, bias, M, C, ncu, s);           // M3 / A3 stage 2
    if (C == 192 && ht == 12) return launch_mlp_stream<12, 12, 6, 4>(z, x, y, wfrag, bias, M, C, ncu, s);           // M1 stage 2
    if (C == 160 && ht == 10) return launch_mlp_stream<10, 10, 6, 4>(z, x, y, wfrag, bias, M, C, ncu, s);           // M5 / A5 stage 1
    if (C == 320 && ht == 20) return launch_mlp_stream<20, 20, 10, 4, 2, false>(z, x, y, wfrag, bias, M, C, ncu, s); // M5 / A5 stage 2
    if (ks1 == 4 && ht == 4) return C == 64 ? launch_mlp<4, 4, 2, true, true>(z, x, y, wfrag, bias, M, C, ncu, s)            // M3 / A3 stage 0 ...
                                            : launch_mlp<4, 4, 2, false, false>(z, x, y, wfrag, bias, M, C, ncu, s);         // ... M2 (56 channels)
    if (ks1 == 8 && ht == 8) {
        if (C != 128) return hipErrorInvalidConfiguration;
        const char* v = rcx::opt::value(rcx::opt::MLP_STREAM);              // RCX_MLP_STREAM=0: the whole-weights-in-LDS kernel (A/B)
        return v && *v == '0' ? launch_mlp<8, 8, 4, true, true>(z, x, y, wfrag, bias, M, C, ncu, s) : launch_mlp_stream<8, 8, 4, 8>(z, x, y, wfrag, bias, M, C, ncu, s);
    }
    if (ks1 == 3 && ht == 3) return C == 48 ? launch_mlp<3, 3, 2, true, false>(z, x, y, wfrag, bias, M, C, ncu, s)            // M1 stage 0 ...
                                            : launch_mlp<3, 3, 2, false, false>(z, x, y, wfrag, bias, M, C, ncu, s);         // ... M0 (40 channels)
    if (ks1 == 6 && ht == 6) return C == 96 ? launch_mlp<6, 6, 3, true, true>(z, x, y, wfrag, bias, M, C, ncu, s) : hipErrorInvalidConfiguration;
    if (ks1 == 5 && ht == 5) return C == 80 ? launch_mlp<5, 5, 3, true, false>(z, x, y, wfrag, bias, M, C, ncu, s) : hipErrorInvalidConfiguration;
    return hipErrorInvalidConfiguration;
}

}  // namespace rcx
