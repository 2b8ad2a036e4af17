#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FMAC_DPP(acc, src, w, ctrl) asm volatile("v_fmac_f32_dpp %0, %1, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(acc) : "v"(src), "v"(w))
#define FMAC(acc, src, w) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(src), "v"(w))

template <int MODE>
__global__ void rate(const float* __restrict__ in, float* __restrict__ out, int iters) {
    int lane = threadIdx.x & 63;
    float r0 = in[lane], r1 = in[64 + lane], w0 = in[128 + lane], w1 = in[192 + lane];
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { FMAC(a[i], r0, w0); FMAC(a[i], r1, w1); FMAC(a[i], r0, w1); FMAC(a[i], r1, w0); }
            if (MODE == 1) { FMAC_DPP(a[i], r0, w0, "row_shr:1"); FMAC_DPP(a[i], r1, w1, "row_shl:2"); FMAC_DPP(a[i], r0, w1, "row_shl:1"); FMAC_DPP(a[i], r1, w0, "row_shr:2"); }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void sem(const float* __restrict__ in, float* __restrict__ out) {
    int lane = threadIdx.x;
    float v = in[lane];
    float one = 1.f;
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    asm volatile("s_nop 4");
    FMAC_DPP(a, v, one, "row_shr:2");
    FMAC_DPP(b, v, one, "row_shl:2");
    if ((lane & 15) < 14) {        // EXEC-masked: do disabled source lanes read as 0?
        asm volatile("s_nop 4");
        FMAC_DPP(c, v, one, "row_shl:1");
        FMAC_DPP(d, v, one, "row_shr:1");
    }
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c; out[192 + lane] = d;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 1024 * 64 * 4);
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (i % 64) + 1;
    hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(in, out);
    std::vector<float> o(256);
    hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    const char* nm[4] = {"shr2", "shl2", "shl1(exec<14)", "shr1(exec<14)"};
    for (int k = 0; k < 4; ++k) { printf("%s:", nm[k]); for (int i = 0; i < 32; ++i) printf(" %g", o[k * 64 + i]); printf("\n"); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        int iters = 4000; int blocks = 256 * 8, threads = 256;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<blocks, threads>>>(in, out, iters); else rate<1><<<blocks, threads>>>(in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fma = double(blocks) * threads * iters * 32;
        printf("mode %d: %.3f ms  %.1f T lane-FMA/s\n", mode, ms, fma / ms * 1e-9);
    }
    return 0;
}
