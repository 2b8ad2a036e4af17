// How fast can ONE wave issue independent v_fmac_f32, and how many waves per SIMD does the peak need? (development ubench)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DPP>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    float a[16], x = in[lane], w = in[64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = in[128 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (DPP && (i & 3) == 0) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(x) : "v"(a[(i + 7) & 15]));
                asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(w));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *in, *out;
    (void)hipMalloc(&in, 4096);
    (void)hipMalloc(&out, 256 * 64 * 256 * 4);
    (void)hipMemset(in, 0, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int dpp = 0; dpp < 2; ++dpp)
        for (int wgs_per_cu : {1, 2, 3, 4, 6, 8}) {
            const int blocks = 256 * wgs_per_cu;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (dpp) k<1><<<blocks, 256>>>(in, out, iters); else k<0><<<blocks, 256>>>(in, out, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double inst_per_wave = double(iters) * 64 * (dpp ? 1.25 : 1.0);
            // cycles per instruction per SIMD assuming an even spread of wgs_per_cu waves on every SIMD and 2.4 GHz
            printf("dpp=%d waves/SIMD=%d  %.3f ms  %.2f cycles/inst/wave  %.2f cycles/inst/SIMD  (%.1f T lane-FMA/s)\n", dpp, wgs_per_cu, ms,
                   ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / (inst_per_wave * wgs_per_cu),
                   double(iters) * 64 * 64 * 4 * blocks / ms * 1e-9);
        }
    return 0;
}
