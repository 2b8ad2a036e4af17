// Stand-alone timing / phase-stamp harness for rcx_cpt.hip (development tool; no torch, no library): compiles the kernel file
// itself, launches it on random data, prints the HIP-event time per launch and, with -DRCX_STAMPS, the phase timeline.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize [-DRCX_STAMPS] [-D...] tools/cpt_bench.hip -o tools/cpt_bench
//   tools/cpt_bench [H=56] [C=64] [N=256] [dtype: 1=bf16 0=f32] [iters=20] [mx: 1 = the matrix-core variant (56x56, bf16)]
#include "../recnext_amd/csrc/rcx_cpt.hip"
#include "../recnext_amd/csrc/rcx_cpt2.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 56, C = argc > 2 ? atoi(argv[2]) : 64, N = argc > 3 ? atoi(argv[3]) : 256;
    const int dt = argc > 4 ? atoi(argv[4]) : 1, iters = argc > 5 ? atoi(argv[5]) : 20;
    const int mx = argc > 6 ? atoi(argv[6]) : 0;
    const int level = H == 56 ? 4 : 3, esz = dt ? 2 : 4;
    const size_t elems = (size_t)N * C * H * H;
    std::vector<unsigned short> hx16(dt ? elems : 0);
    std::vector<float> hx32(dt ? 0 : elems);
    srand(1);
    for (size_t i = 0; i < elems; ++i) {
        const float v = (float)(rand() % 2001 - 1000) / 500.f;
        if (dt) { unsigned u; memcpy(&u, &v, 4); hx16[i] = (unsigned short)(u >> 16); } else hx32[i] = v;
    }
    std::vector<float> hw((size_t)(level + 2) * 25 * C);
    for (auto& w : hw) w = (float)(rand() % 2001 - 1000) / 5000.f;
    void *x, *y; float* w;
    CK(hipMalloc(&x, elems * esz)); CK(hipMalloc(&y, elems * esz)); CK(hipMalloc(&w, hw.size() * 4));
    CK(hipMemcpy(x, dt ? (void*)hx16.data() : (void*)hx32.data(), elems * esz, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
#ifdef RCX_STAMPS
    unsigned long long* st; const size_t nst = 512 * 8 * 16;
    CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(rcx::cpt::g_cpt_stamps), &st, sizeof(st)));
#endif
    // the matrix pack built on the host (k_pack_mx's layout: [conv][u][slot][i][C] x four bf16)
    void* mxp = nullptr;
    if (mx) {
        std::vector<unsigned short> hm((size_t)(level + 2) * 5 * 3 * 4 * C * 4, 0);
        for (int j = 0; j < level + 2; ++j) for (int u = 0; u < 5; ++u) for (int kb = 0; kb < 3; ++kb) for (int i = 0; i < 4; ++i) for (int c = 0; c < C; ++c)
            for (int k = 0; k < 4; ++k) {
                const int st = j == 0 ? 2 : 1, v = 4 * kb + k - st * i;
                float wv = (v >= 0 && v <= 4 && (st == 2 || kb < 2)) ? hw[(size_t)(j * 25 + u * 5 + v) * C + c] : 0.f;
                unsigned bits; memcpy(&bits, &wv, 4);
                bits += 0x7FFF + ((bits >> 16) & 1);
                hm[((((size_t)(j * 5 + u) * 3 + kb) * 4 + i) * C + c) * 4 + k] = (unsigned short)(bits >> 16);
            }
        CK(hipMalloc(&mxp, hm.size() * 2)); CK(hipMemcpy(mxp, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
    }
    auto run = [&](hipStream_t st_) { return mx ? rcx::cpt_mx_recconv(x, y, w, nullptr, mxp, N, C, 0, dt, st_) : rcx::cpt_recconv(x, y, w, nullptr, N, C, H, level, 0, dt, st_); };
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int i = 0; i < 3; ++i) CK(run(s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) CK(run(s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / iters * 1000.f);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = 2.0 * elems * esz + (double)(level + 2) * 25 * C * esz;
    printf("%sH=%d C=%d N=%d %s: %.2f us per launch (min %.2f)  %.0f GB/s algorithmic = %.3f of 8 TB/s\n", mx ? "matrix cores " : "", H, C, N, dt ? "bf16" : "f32", ts[2], ts[0],
           bytes / ts[2] / 1e3, bytes / ts[2] / 1e3 / 8000.0);
#ifdef RCX_STAMPS
    CK(hipMemset(st, 0, nst * 8)); CK(hipDeviceSynchronize());
    CK(run(s)); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"start", "LDS zeroed, taps", "pass 1 done", "barrier", "down ladder", "up pieces", "T1 formed", "C1 done", "pass 2 done"};
    const int nw = H == 56 ? 8 : 4;
    unsigned long long t00 = ~0ull;
    for (int b = 0; b < 512; ++b) if (h[(size_t)(b * 8) * 16]) t00 = std::min(t00, h[(size_t)(b * 8) * 16]);
    for (int id = 0; id < 9; ++id) {
        std::vector<double> rel, absd;
        for (int b = 0; b < 512; ++b)
            for (int wv = 0; wv < nw; ++wv) {
                const unsigned long long* p = &h[(size_t)(b * 8 + wv) * 16];
                if (!p[0] || !p[id]) continue;
                rel.push_back((double)(p[id] - p[id ? id - 1 : 0]));
                absd.push_back((double)(p[id] - t00));
            }
        if (rel.empty()) continue;
        std::sort(rel.begin(), rel.end()); std::sort(absd.begin(), absd.end());
        printf("  %-18s phase: median %8.0f  p10 %8.0f  p90 %8.0f   | since first start: median %8.0f  max %8.0f   (ticks, %zu waves)\n", names[id],
               rel[rel.size() / 2], rel[rel.size() / 10], rel[rel.size() * 9 / 10], absd[absd.size() / 2], absd.back(), rel.size());
    }
    {   // wall-clock timeline (100 MHz s_memrealtime): workgroup start and end relative to the first start
        unsigned long long r0 = ~0ull;
        for (int b = 0; b < 512; ++b) if (h[(size_t)(b * 8) * 16 + 9]) r0 = std::min(r0, h[(size_t)(b * 8) * 16 + 9]);
        std::vector<double> st0, en, du;
        for (int b = 0; b < 512; ++b) {
            const unsigned long long* p = &h[(size_t)(b * 8) * 16];
            if (!p[9] || !p[10]) continue;
            st0.push_back((p[9] - r0) / 100.0); en.push_back((p[10] - r0) / 100.0); du.push_back((p[10] - p[9]) / 100.0);
        }
        std::sort(st0.begin(), st0.end()); std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
        const size_t n = st0.size();
        printf("  workgroups %zu: start us p10 %.1f p50 %.1f p90 %.1f max %.1f | end us p10 %.1f p50 %.1f p90 %.1f max %.1f | duration us p10 %.1f p50 %.1f p90 %.1f\n", n,
               st0[n / 10], st0[n / 2], st0[n * 9 / 10], st0[n - 1], en[n / 10], en[n / 2], en[n * 9 / 10], en[n - 1], du[n / 10], du[n / 2], du[n * 9 / 10]);
    }
#endif
    return 0;
}
