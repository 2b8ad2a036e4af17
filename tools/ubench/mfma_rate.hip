// Microbenchmark (development tool): cycles per v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 on gfx950 -- a chain on ONE accumulator against
// NACC independent accumulators, one wave per SIMD and two.   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BIG>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters)
{
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int n = 0; n < NACC; ++n) { for (int i = 0; i < 16; ++i) acc[n][i] = 0.f; for (int i = 0; i < 4; ++i) acc4[n][i] = 0.f; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if constexpr (BIG) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
            else acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[n], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) { for (int i = 0; i < 16; ++i) s += acc[n][i]; for (int i = 0; i < 4; ++i) s += acc4[n][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, bool BIG>
void run(const char* name, int threads)
{
    float* out; long long* cyc; long long h = 0;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(threads), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC, flops = (BIG ? 32768.0 : 16384.0) * n * (threads / 64) * 256;
    printf("%-28s waves/SIMD %d  s_memtime ticks per MFMA %.1f  us %.1f  => %.0f TFLOP/s, %.1f ns per MFMA per wave\n", name, threads / 256, h / n, ms * 1e3, flops / ms / 1e9, ms * 1e6 / n);
}

int main()
{
    run<1, true>("32x32x16 one accumulator", 256);
    run<4, true>("32x32x16 four accumulators", 256);
    run<8, true>("32x32x16 eight accumulators", 256);
    run<4, true>("32x32x16 four acc", 512);
    run<1, false>("16x16x32 one accumulator", 256);
    run<4, false>("16x16x32 four accumulators", 256);
    run<8, false>("16x16x32 eight acc", 512);
    return 0;
}
