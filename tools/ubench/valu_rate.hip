// Micro-benchmark (development tool): issue cost of v_fma_f32 vs v_pk_fma_f32 vs ds_read_b64 on gfx950,
// at 1/2/4 waves per SIMD. Prints cycles per instruction per wave (s_memtime around an unrolled loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, int iters)
{
    extern __shared__ float2 lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float2(i * 0.001f, 1.f);
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    v2f p0 = {a0, 1}, p1 = {1, 2}, p2 = {2, 3}, p3 = {3, 4}, p4 = {a0, 2}, p5 = {2, 2}, p6 = {1, 1}, p7 = {0, 3};
    const float w = 1.0001f, b = 0.5f;
    const v2f pw = {1.0001f, 0.9999f}, pb = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a0 = fmaf(a0, w, b); a1 = fmaf(a1, w, b); a2 = fmaf(a2, w, b); a3 = fmaf(a3, w, b);
                a4 = fmaf(a4, w, b); a5 = fmaf(a5, w, b); a6 = fmaf(a6, w, b); a7 = fmaf(a7, w, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p0 = __builtin_elementwise_fma(p0, pw, pb); p1 = __builtin_elementwise_fma(p1, pw, pb);
                p2 = __builtin_elementwise_fma(p2, pw, pb); p3 = __builtin_elementwise_fma(p3, pw, pb);
                p4 = __builtin_elementwise_fma(p4, pw, pb); p5 = __builtin_elementwise_fma(p5, pw, pb);
                p6 = __builtin_elementwise_fma(p6, pw, pb); p7 = __builtin_elementwise_fma(p7, pw, pb);
            }
        } else {
            // 8 independent ds_read_b64 + 8 fma on them
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2* q = lds + ((threadIdx.x + r * 64 + it) & 2047);
                float2 v0 = q[0], v1 = q[64], v2 = q[128], v3 = q[192], v4 = q[256], v5 = q[320], v6 = q[384], v7 = q[448];
                a0 = fmaf(v0.x, w, a0); a1 = fmaf(v1.x, w, a1); a2 = fmaf(v2.y, w, a2); a3 = fmaf(v3.x, w, a3);
                a4 = fmaf(v4.y, w, a4); a5 = fmaf(v5.x, w, a5); a6 = fmaf(v6.y, w, a6); a7 = fmaf(v7.x, w, a7);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads)
{
    const int blocks = 256, iters = 200;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 32768, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 32768, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= h.size();
    const double insts = (double)iters * 16 * 8 * (MODE == 2 ? 2 : 1);
    printf("%-14s waves/SIMD=%d  cycles/inst/wave=%6.2f  wall=%.3f ms  clock(cycles/wall)=%.2f GHz\n", name, threads / 256,
           s / insts, ms, s / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int t : {256, 512, 1024}) { run<0>("v_fma_f32", t); run<1>("v_pk_fma_f32", t); run<2>("ds_read_b64+fma", t); }
    return 0;
}
