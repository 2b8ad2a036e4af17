// Timing / phase-stamp harness for the matrix-core variant alone (development tool): one instantiation, so it builds in under a minute;
// -DRCX_ABL=<bits> switches parts of the passes off (rcx_cpt_kernel.h) to see what a step's time is made of -- results are then wrong.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DRCX_STAMPS [-DRCX_ABL=n] [-DRCX_MX_AHEAD=n] tools/mx_bench.hip -o tools/mx_bench
#include "rcx_cpt_kernel_mx.h"   /* the frozen header with the matrix-core passes (this directory) */
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int H = 56, C = 64, N = argc > 1 ? atoi(argv[1]) : 256, iters = 20, level = 4;
    const size_t elems = (size_t)N * C * H * H;
    std::vector<unsigned short> hx(elems);
    srand(1);
    for (size_t i = 0; i < elems; ++i) { const float v = (float)(rand() % 2001 - 1000) / 500.f; unsigned u; memcpy(&u, &v, 4); hx[i] = (unsigned short)(u >> 16); }
    std::vector<float> hw((size_t)(level + 2) * 25 * C);
    for (auto& w : hw) w = (float)(rand() % 2001 - 1000) / 5000.f;
    std::vector<unsigned short> hm((size_t)(level + 2) * 5 * 3 * 4 * C * 4, 0);
    for (int j = 0; j < level + 2; ++j) for (int u = 0; u < 5; ++u) for (int kb = 0; kb < 3; ++kb) for (int i = 0; i < 4; ++i) for (int c = 0; c < C; ++c)
        for (int k = 0; k < 4; ++k) {
            const int st = j == 0 ? 2 : 1, v = 4 * kb + k - st * i;
            float wv = (v >= 0 && v <= 4 && (st == 2 || kb < 2)) ? hw[(size_t)(j * 25 + u * 5 + v) * C + c] : 0.f;
            unsigned bits; memcpy(&bits, &wv, 4); bits += 0x7FFF + ((bits >> 16) & 1);
            hm[((((size_t)(j * 5 + u) * 3 + kb) * 4 + i) * C + c) * 4 + k] = (unsigned short)(bits >> 16);
        }
    void *x, *y, *mxp; float* w;
    CK(hipMalloc(&x, elems * 2)); CK(hipMalloc(&y, elems * 2)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&mxp, hm.size() * 2));
    CK(hipMemcpy(x, hx.data(), elems * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(mxp, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
#ifdef RCX_STAMPS
    unsigned long long* st; const size_t nst = 512 * 8 * 16;
    CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(rcx::cpt::g_cpt_stamps), &st, sizeof(st)));
#endif
    const rcx::cpt::SavedPyr sv{};
    auto run = [&](hipStream_t s) { return rcx::cpt::launch<4, 2, 0, 128, rcx::bf16_t, false, true>(x, y, w, nullptr, N, C, s, sv, mxp); };
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
    printf("ABL=%d AHEAD=%d/%d N=%d: %.2f us per launch (min %.2f)", RCX_ABL, RCX_MX_AHEAD1, RCX_MX_AHEAD2, N, ts[2], ts[0]);
#ifdef RCX_STAMPS
    CK(hipMemset(st, 0, nst * 8)); CK(hipDeviceSynchronize());
    CK(run(s)); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"start", "taps", "pass1", "barrier", "ladder", "pieces", "T1", "C1", "pass2"};
    printf("  | phase medians (cycles):");
    for (int id = 1; id < 9; ++id) {
        std::vector<double> rel;
        for (int b = 0; b < 512; ++b) for (int wv = 0; wv < 8; ++wv) {
            const unsigned long long* p = &h[(size_t)(b * 8 + wv) * 16];
            if (p[0] && p[id]) rel.push_back((double)(p[id] - p[id - 1]));
        }
        if (rel.empty()) continue;
        std::sort(rel.begin(), rel.end());
        printf(" %s %.0f", names[id], rel[rel.size() / 2]);
    }
#endif
    printf("\n");
    return 0;
}
