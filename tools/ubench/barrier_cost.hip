// Micro-benchmark (development tool): cost of a __syncthreads() phase on gfx950 as a function of workgroup size
// and of the LDS footprint / dynamic LDS request (large requests pin one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WORK>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, int iters)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (WORK >= 1) acc += lds[(threadIdx.x + it * 7) % blockDim.x];
        if (WORK >= 2) {
#pragma unroll
            for (int r = 0; r < 25; ++r) acc = fmaf(acc, 1.0001f, lds[(threadIdx.x + r) % blockDim.x]);
        }
        __syncthreads();
        if (WORK >= 1) lds[threadIdx.x] = acc;
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WORK>
void run(const char* name, int threads, size_t lds_bytes)
{
    const int blocks = 1024, iters = 100;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WORK>, dim3(blocks), dim3(threads), lds_bytes, 0, out, cyc, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<WORK>, dim3(blocks), dim3(threads), lds_bytes, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= blocks;
    printf("%-22s threads=%4d lds=%6zu KB  ticks per (2 barriers + work) = %8.1f   wall %.3f ms\n", name, threads, lds_bytes / 1024, s / iters, ms);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int t : {128, 256, 512, 1024})
        for (size_t l : {(size_t)8 * 1024, (size_t)150 * 1024}) {
            run<0>("barrier only", t, l);
            run<1>("1 LDS rd + 1 wr", t, l);
            run<2>("25 LDS rd + fma", t, l);
        }
    return 0;
}
