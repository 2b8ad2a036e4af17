// One instantiation of k_recconv_cpt, timed the way a model runs it (development tool; no torch, no library): x is rewritten by a copy
// kernel and 256 MB of other data are touched before EVERY launch, and each launch is bracketed by its own pair of events.  A loop over a
// read-only x (tools/cpt_bench.hip) ranks some variants the other way round (profiles/archive/r03_cpt_cb16.txt, r03_cpt_fresh_sweep.txt).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DVARIANT=0 [-DRCX_CPT_AHEAD1=4 ...] tools/cpt_one.hip -o tools/cpt_one_v0
//   tools/cpt_one_v0 [N=256] [iters=40] [fresh=1]
// VARIANT 0: <4, 2, 0, 128, bf16> (56x56x64), 1: <4, 4, 0, 128, bf16>, 2: <2, 1, 0, 256, bf16> (28x28x128), 3: <2, 2, 0, 0, bf16> (28x28x96),
//         4: 16-pixel tiles, 64x64x128 / level 3 (config 5, default N = 32)
//   -DSTGN=3 (56x56) / 2 (28x28): x rows by LDS-DMA (with -DRCX_CPT_STG_P2=0/1: pass 1 only / both passes); -DRCX_STAMPS: the phase timeline of the last unit
#include "../recnext_amd/csrc/rcx_cpt_kernel.h"
namespace rcx { LaunchEvents take_launch_events() { return LaunchEvents{}; } }
#ifndef STGN
#define STGN 0
#endif


#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef VARIANT
#define VARIANT 0
#endif
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

__global__ void k_scrub(float* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f; }
__global__ void k_copy16(const uint4* a, uint4* b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }

int main(int argc, char** argv)
{
    using namespace rcx;
    const int N = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 40, fresh = argc > 3 ? atoi(argv[3]) : 1;
    constexpr int H = VARIANT == 4 ? 64 : (VARIANT < 2 ? 56 : 28), C = VARIANT == 4 ? 128 : (VARIANT < 2 ? 64 : (VARIANT == 2 ? 128 : 96)), level = VARIANT < 2 ? 4 : 3;
    const size_t elems = (size_t)N * C * H * H;
    std::vector<unsigned short> hx(elems);
    srand(1);
    for (size_t i = 0; i < elems; ++i) { const float v = (float)(rand() % 2001 - 1000) / 500.f; unsigned u; memcpy(&u, &v, 4); hx[i] = (unsigned short)(u >> 16); }
    std::vector<float> hw((size_t)(level + 2) * 25 * C);
    for (auto& w : hw) w = (float)(rand() % 2001 - 1000) / 5000.f;
    void *x, *x0, *y; float *w, *scrub;
    const size_t nscrub = 64u << 20;
    CK(hipMalloc(&x, elems * 2)); CK(hipMalloc(&x0, elems * 2)); CK(hipMalloc(&y, elems * 2)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&scrub, nscrub * 4));
    CK(hipMemcpy(x0, hx.data(), elems * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x, x0, elems * 2, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(scrub, 0, nscrub * 4));
    const cpt::SavedPyr sv{};
    auto run = [&](hipStream_t s) {
#if VARIANT == 0
        return cpt::launch<4, 2, 0, 128, bf16_t, false, 4, STGN>(x, y, w, nullptr, N, C, s, sv);
#elif VARIANT == 1
        return cpt::launch<4, 4, 0, 128, bf16_t, false, 4, STGN>(x, y, w, nullptr, N, C, s, sv);
#elif VARIANT == 2
        return cpt::launch<2, 1, 0, 256, bf16_t, false, 3, STGN>(x, y, w, nullptr, N, C, s, sv);
#elif VARIANT == 4
        (void)sv;
        return cpt::launch16<0, bf16_t>(x, y, w, nullptr, N, C, s);
#else
        return cpt::launch<2, 2, 0, 0, bf16_t, false, 3, STGN>(x, y, w, nullptr, N, C, s, sv);
#endif
    };
    hipStream_t s; CK(hipStreamCreate(&s));
#ifdef RCX_STAMPS
    unsigned long long* st; const size_t nst = 512 * 8 * 4 * 16;
    CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(rcx::cpt::g_cpt_stamps), &st, sizeof(st)));
#endif
    for (int i = 0; i < 3; ++i) CK(run(s));
    CK(hipStreamSynchronize(s));
    std::vector<hipEvent_t> e0(iters), e1(iters);
    for (int i = 0; i < iters; ++i) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    for (int i = 0; i < iters; ++i) {
        if (fresh) {
            hipLaunchKernelGGL(k_scrub, dim3(2048), dim3(256), 0, s, scrub, nscrub);
            hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, s, (const uint4*)x0, (uint4*)x, elems * 2 / 16);
        }
        CK(hipEventRecord(e0[i], s));
        CK(run(s));
        CK(hipEventRecord(e1[i], s));
    }
    CK(hipStreamSynchronize(s));
    std::vector<float> ts(iters);
    for (int i = 0; i < iters; ++i) { CK(hipEventElapsedTime(&ts[i], e0[i], e1[i])); ts[i] *= 1000.f; }
    std::sort(ts.begin(), ts.end());
    const double bytes = 2.0 * elems * 2 + (double)(level + 2) * 25 * C * 2;
    printf("variant %d STG=%d/P2=%d SKIPW=%d H=%d C=%d N=%d bf16 %s AHEAD1=%d AHEAD2=%d: median %.2f us (min %.2f, p90 %.2f)  %.3f of 8 TB/s\n", VARIANT, STGN, RCX_CPT_STG_P2, RCX_CPT_SKIPW, H, C, N, fresh ? "fresh" : "loop",
           RCX_CPT_AHEAD1, RCX_CPT_AHEAD2, ts[iters / 2], ts[0], ts[iters * 9 / 10], bytes / ts[iters / 2] / 1e3 / 8000.0);
    // checksum of y so that variants can be compared
    std::vector<unsigned short> hy(elems);
    CK(hipMemcpy(hy.data(), y, elems * 2, hipMemcpyDeviceToHost));
    unsigned long long cs = 0; for (size_t i = 0; i < elems; ++i) cs = cs * 1315423911ull + hy[i];
    printf("  y checksum %016llx\n", cs);
#ifdef RCX_STAMPS
    CK(hipMemset(st, 0, nst * 8)); CK(hipDeviceSynchronize());
    if (fresh) {
        hipLaunchKernelGGL(k_scrub, dim3(2048), dim3(256), 0, s, scrub, nscrub);
        hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, s, (const uint4*)x0, (uint4*)x, elems * 2 / 16);
    }
    CK(run(s)); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(nst);
    CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"start", "taps, barrier", "pass 1", "barrier", "down ladder", "up pieces", "T1", "C1", "pass 2"};
    const int nw = VARIANT == 0 ? 8 : (VARIANT == 3 ? 2 : 4);        // waves per workgroup
    unsigned long long r0 = ~0ull;                       // the first wave's start on the 100 MHz clock
    for (size_t k = 0; k < nst / 16; ++k) if (h[k * 16 + 9]) r0 = std::min(r0, h[k * 16 + 9]);
    for (int it = 0; it < 4; ++it) {
        double tot = 0;
        std::vector<double> t0s, t1s;
        printf("  unit %d of a workgroup:", it);
        for (int id = 1; id < 9; ++id) {
            std::vector<double> rel;
            for (int b = 0; b < 512; ++b)
                for (int wv = 0; wv < nw; ++wv) {
                    const unsigned long long* p = &h[(size_t)((b * 8 + wv) * 4 + it) * 16];
                    if (!p[0] || !p[id]) continue;
                    rel.push_back((double)(p[id] - p[id - 1]));
                    if (id == 1) { t0s.push_back((p[9] - r0) / 100.0); t1s.push_back((p[10] - r0) / 100.0); }
                }
            if (rel.empty()) break;
            std::sort(rel.begin(), rel.end());
            tot += rel[rel.size() / 2];
            printf(" %s %.0f (%.0f-%.0f)", names[id], rel[rel.size() / 2], rel[rel.size() / 10], rel[rel.size() * 9 / 10]);
        }
        if (t0s.empty()) { printf(" -\n"); continue; }
        std::sort(t0s.begin(), t0s.end()); std::sort(t1s.begin(), t1s.end());
        printf("\n     sum of medians %.0f cycles; wall clock since the first wave: starts p10 %.1f p50 %.1f p90 %.1f us, ends p10 %.1f p50 %.1f p90 %.1f max %.1f us\n", tot,
               t0s[t0s.size() / 10], t0s[t0s.size() / 2], t0s[t0s.size() * 9 / 10], t1s[t1s.size() / 10], t1s[t1s.size() / 2], t1s[t1s.size() * 9 / 10], t1s.back());
    }
#endif
    return 0;
}
