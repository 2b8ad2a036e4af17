// Micro-benchmark (development tool): what a vector-memory LOAD wave-instruction costs the CU on gfx950, by access width and by how many
// 128-byte lines its 64 lanes touch -- the quantity the tiled RecConv2d kernels are bound by.  Every wave streams over its own
// region (L1-missing, L2-resident), NW waves per CU, loads issued in groups of 8 behind one counted wait; prints CU cycles per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/vmem_rate.hip -o tools/ubench/vmem_rate && tools/ubench/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// MODE: 0 short_d16_hi, 1 ushort, 2 dword, 3 dwordx2, 4 dwordx4;  LINES: lines per instruction for the 2-byte modes (1, 2, 4, 8: the wave is
// split into LINES groups of 64 / LINES lanes, each group contiguous inside a different line)
template <int MODE, int LINES>
__global__ void __launch_bounds__(1024) k(const char* __restrict__ base, unsigned* out, unsigned long long* cyc, int iters, int region)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wid = blockIdx.x * (blockDim.x >> 6) + wave;
    i32x4 rs;
    const unsigned long long a = (unsigned long long)(base + (size_t)wid * region);
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    rs.z = region;
    rs.w = 0x00020000;
    constexpr int BPL = MODE <= 1 ? 2 : (MODE == 2 ? 4 : (MODE == 3 ? 8 : 16));     // bytes per lane
    constexpr int STEP = MODE <= 1 ? 128 * LINES : 64 * BPL;                          // bytes one instruction covers (as address range)
    unsigned voff;
    if (MODE <= 1) { const int g = lane / (64 / LINES), i = lane % (64 / LINES); voff = g * 128 + i * 2; }
    else voff = lane * BPL;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int so = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned r[8];
        u32x2 r2[8];
        u32x4 r4[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) asm volatile("buffer_load_short_d16_hi %0, %1, %2, %3 offen" : "=v"(r[j]) : "v"(voff), "s"(rs), "s"(so));
            else if (MODE == 1) asm volatile("buffer_load_ushort %0, %1, %2, %3 offen" : "=v"(r[j]) : "v"(voff), "s"(rs), "s"(so));
            else if (MODE == 2) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r[j]) : "v"(voff), "s"(rs), "s"(so));
            else if (MODE == 3) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(r2[j]) : "v"(voff), "s"(rs), "s"(so));
            else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r4[j]) : "v"(voff), "s"(rs), "s"(so));
            so += STEP;
            if (so + STEP > region) so = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE <= 2) { asm volatile("" : "+v"(r[j])); acc ^= r[j]; }
            else if (MODE == 3) { asm volatile("" : "+v"(r2[j])); acc ^= r2[j].x ^ r2[j].y; }
            else { asm volatile("" : "+v"(r4[j])); acc ^= r4[j].x ^ r4[j].w; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[wid] = t1 - t0;
}

template <int MODE, int LINES>
void run(const char* name, int nw, const char* buf, unsigned* out, unsigned long long* cyc, int region)
{
    const int blocks = 256, iters = 400;
    hipLaunchKernelGGL((k<MODE, LINES>), dim3(blocks), dim3(nw * 64), 0, 0, buf, out, cyc, iters, region);
    hipLaunchKernelGGL((k<MODE, LINES>), dim3(blocks), dim3(nw * 64), 0, 0, buf, out, cyc, iters, region);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * nw);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // a CU holds one block: nw waves issue iters * 8 instructions each in `med` cycles
    printf("%-34s waves/CU %2d: %6.2f cycles of the CU per wave-instruction (%.1f B/clk/CU)\n", name, nw, med / (iters * 8.0 * nw),
           (MODE <= 1 ? 128.0 : MODE == 2 ? 256.0 : MODE == 3 ? 512.0 : 1024.0) * iters * 8.0 * nw / med);
}

int main()
{
    const int region = 8192;                             // bytes per wave: 16 waves x 8 KB = 128 KB per CU (L1 is 32 KB), 32 MB for the chip
    char* buf; unsigned* out; unsigned long long* cyc;
    hipMalloc(&buf, (size_t)256 * 16 * region); hipMemset(buf, 1, (size_t)256 * 16 * region);
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    for (int nw : {4, 8, 16}) {
        run<0, 1>("short_d16_hi, 1 line", nw, buf, out, cyc, region);
        run<0, 2>("short_d16_hi, 2 lines", nw, buf, out, cyc, region);
        run<0, 4>("short_d16_hi, 4 lines", nw, buf, out, cyc, region);
        run<0, 8>("short_d16_hi, 8 lines", nw, buf, out, cyc, region);
        run<1, 2>("ushort, 2 lines", nw, buf, out, cyc, region);
        run<2, 1>("dword (256 B contiguous)", nw, buf, out, cyc, region);
        run<3, 1>("dwordx2 (512 B contiguous)", nw, buf, out, cyc, region);
        run<4, 1>("dwordx4 (1 KB contiguous)", nw, buf, out, cyc, region);
    }
    return 0;
}
