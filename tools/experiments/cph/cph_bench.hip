// Stand-alone timing / phase-stamp harness for rcx_cph_kernel.h (development tool; no torch, no library): launches the half-tile kernel on
// random data, prints the HIP-event time per launch and, with -DRCX_STAMPS, the phase timeline.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Irecnext_amd/csrc [-DRCX_STAMPS] [-DRCX_CPH_ABL=n] tools/experiments/cph/cph_bench.hip -o /tmp/cph_bench
//   /tmp/cph_bench [H=56] [N=256] [iters=20] [grid cap]        (bf16, C = 64 at 56x56, 128 at 28x28)
#include "rcx_cph_kernel.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 56, N = argc > 2 ? atoi(argv[2]) : 256, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int C = H == 56 ? 64 : 128, level = H == 56 ? 4 : 3, esz = 2;
    const size_t elems = (size_t)N * C * H * H;
    std::vector<unsigned short> hx16(elems);
    srand(1);
    for (size_t i = 0; i < elems; ++i) {
        const float v = (float)(rand() % 2001 - 1000) / 500.f;
        unsigned u; memcpy(&u, &v, 4); hx16[i] = (unsigned short)(u >> 16);
    }
    std::vector<float> hw((size_t)(level + 2) * 25 * C);
    for (auto& w : hw) w = (float)(rand() % 2001 - 1000) / 5000.f;
    void *x, *y; float* w;
    CK(hipMalloc(&x, elems * esz)); CK(hipMalloc(&y, elems * esz)); CK(hipMalloc(&w, hw.size() * 4));
    CK(hipMemcpy(x, hx16.data(), elems * esz, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
#ifdef RCX_STAMPS
    unsigned long long* st; const size_t nst = 512 * 16 * 16;
    CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(rcx::cph::g_cph_stamps), &st, sizeof(st)));
#endif
    const rcx::cpt::SavedPyr sv{};
    auto run = [&](hipStream_t st_) {
        return H == 56 ? rcx::cph::launch<4, 0, 128, rcx::bf16_t>(x, y, w, nullptr, N, C, st_, sv) : rcx::cph::launch<2, 0, 256, rcx::bf16_t>(x, y, w, nullptr, N, C, st_, sv);
    };
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
    printf("H=%d C=%d N=%d bf16: %.2f us per launch (min %.2f)  %.0f GB/s algorithmic = %.3f of 8 TB/s\n", H, C, N, ts[2], ts[0], bytes / ts[2] / 1e3, bytes / ts[2] / 1e3 / 8000.0);
#ifdef RCX_STAMPS
    CK(hipMemset(st, 0, nst * 8)); CK(hipDeviceSynchronize());
    CK(run(s)); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"start", "first barrier", "pass 1 done", "barrier", "down ladder", "up pieces", "T1 formed", "C1 done", "pass 2 done"};
    const int nw = H == 56 ? 16 : 8;
    for (int id = 0; id < 9; ++id) {
        std::vector<double> rel;
        for (int b = 0; b < 512; ++b)
            for (int wv = 0; wv < nw; ++wv) {
                const unsigned long long* p = &h[(size_t)(b * 16 + wv) * 16];
                if (!p[0] || !p[id]) continue;
                rel.push_back((double)(p[id] - p[id ? id - 1 : 0]));
            }
        if (rel.empty()) continue;
        std::sort(rel.begin(), rel.end());
        printf("  %-14s phase: median %8.0f  p10 %8.0f  p90 %8.0f  (cycles, %zu waves)\n", names[id], rel[rel.size() / 2], rel[rel.size() / 10], rel[rel.size() * 9 / 10], rel.size());
    }
    {   // per half: pass 1 and C1 (waves of half 1 have 3/4 of the work)
        for (int hh = 0; hh < 2; ++hh) {
            std::vector<double> p1, p2;
            for (int b = 0; b < 512; ++b)
                for (int wv = 0; wv < nw; ++wv) {
                    const int half = H == 56 ? ((wv >> 2) & 1) : (wv >> 2);
                    const unsigned long long* p = &h[(size_t)(b * 16 + wv) * 16];
                    if (half != hh || !p[0] || !p[8]) continue;
                    p1.push_back((double)(p[2] - p[1])); p2.push_back((double)(p[8] - p[7]));
                }
            if (p1.empty()) continue;
            std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
            printf("  half %d: pass 1 median %.0f  pass 2 median %.0f\n", hh, p1[p1.size() / 2], p2[p2.size() / 2]);
        }
        unsigned long long r0 = ~0ull;
        for (int b = 0; b < 512; ++b) if (h[(size_t)(b * 16) * 16 + 9]) r0 = std::min(r0, h[(size_t)(b * 16) * 16 + 9]);
        std::vector<double> du;
        for (int b = 0; b < 512; ++b) {
            const unsigned long long* p = &h[(size_t)(b * 16) * 16];
            if (!p[9] || !p[10]) continue;
            du.push_back((p[10] - p[9]) / 100.0);
        }
        std::sort(du.begin(), du.end());
        if (!du.empty()) printf("  last unit of a workgroup, wall: p10 %.1f p50 %.1f p90 %.1f us\n", du[du.size() / 10], du[du.size() / 2], du[du.size() * 9 / 10]);
    }
#endif
    return 0;
}
