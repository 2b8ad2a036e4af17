// Probe (development tool): what `buffer_load_dwordx4 ... offen lds` (LDS-DMA) and `ds_read_b64_tr_b16` do on gfx950, checked with exact data:
//   (1) lane i's 16 bytes land at M0 + inst_offset + 16 i -- also for M0 beyond 64 KB (the staging area of the tiled kernels sits above 100 KB);
//   (2) an out-of-range lane writes zeros; (3) an EXEC-masked lane writes nothing;
//   (4) the transposing read: lane 16 g + 4 q + p supplies the address of row q, 8-byte chunk p; lane 16 g + i receives element i of rows 0..3.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/ldsdma_probe.hip -o tools/ubench/ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64) k(const unsigned short* src, unsigned* out, int m0base)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 4; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    i32x4 rs;
    const unsigned long long a = (unsigned long long)src;
    rs.x = (int)(unsigned)a; rs.y = (int)(unsigned)(a >> 32) & 0xffff; rs.z = 64 * 64 * 2; rs.w = 0x00020000;    // 64 pixels x 64 channels of 16 bits
    // piece 0: lanes 0..63 = 16 pixels x 4 chunks of 16 bytes (channels 0..31 of a 128-byte pixel): LDS image [pixel][64 B]
    unsigned voff = (lane >> 2) * 128 + (lane & 3) * 16;
    if (lane == 5) voff = 0x80000000u;                        // out of range: zeros?
    const int soff = 0;
    // piece 1 (EXEC = lanes 0..31 only), 1 KB behind piece 0
    asm volatile("s_mov_b32 m0, %[m]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[rs], %[so] offen lds\n\t"
                 "s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffffffff\n\t"
                 "buffer_load_dwordx4 %[v], %[rs], %[so] offen offset:1024 lds\n\ts_mov_b64 exec, s[20:21]\n\ts_waitcnt vmcnt(0)"
                 :: [m] "s"(m0base), [v] "v"(voff), [rs] "s"(rs), [so] "s"(soff) : "memory", "s20", "s21");
    __syncthreads();
    // dump 2 KB from m0base
    for (int i = lane; i < 512; i += 64) out[i] = reinterpret_cast<unsigned*>(lds + m0base)[i];
    // transposing read of pixels 4..7: group g = lane >> 4 -> channels 16 (g & 1) .. + 15 (groups 2, 3 repeat 0, 1); lane 4 q + p: pixel 4 + q, chunk p
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds + m0base + (4 + q) * 64 + (g & 1) * 32 + p * 8);
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[512 + 2 * lane] = r.x;
    out[512 + 2 * lane + 1] = r.y;
}

int main()
{
    std::vector<unsigned short> h(64 * 64);
    for (int px = 0; px < 64; ++px) for (int c = 0; c < 64; ++c) h[px * 64 + c] = (unsigned short)(px * 256 + c);
    unsigned short* src; unsigned* out;
    hipMalloc(&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 1024 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int m0 : {4096, 70000 & ~15, 150000 & ~15}) {
        hipMemset(out, 0, 1024 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, src, out, m0);
        std::vector<unsigned> o(1024);
        if (hipMemcpy(o.data(), out, 1024 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
        int bad = 0, oob_zero = 1, masked_kept = 1;
        for (int lane = 0; lane < 64; ++lane)
            for (int d = 0; d < 4; ++d) {
                const unsigned got = o[lane * 4 + d];
                const int px = lane >> 2, c0 = (lane & 3) * 8 + d * 2;
                const unsigned want = (unsigned)(px * 256 + c0) | ((unsigned)(px * 256 + c0 + 1) << 16);
                if (lane == 5) { if (got != 0) oob_zero = 0; }
                else if (got != want) ++bad;
                const unsigned got2 = o[256 + lane * 4 + d];                                   // piece 1
                if (lane >= 32) { if (got2 != 0xdeadbeefu) masked_kept = 0; }
                else if (lane != 5 && got2 != want) ++bad;
            }
        int trbad = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, i = lane & 15, c = (g & 1) * 16 + i;
            for (int e = 0; e < 4; ++e) {
                const unsigned v = (o[512 + 2 * lane + (e >> 1)] >> (16 * (e & 1))) & 0xffff;
                unsigned want = (unsigned)((4 + e) * 256 + c);
                if (4 + e == 1 && c >= 8 && c < 16) want = 0;                                   // (pixel 1 chunk 1 was the out-of-range lane: not in 4..7)
                if (v != want) ++trbad;
            }
        }
        printf("M0 = %6d: placement errors %d, out-of-range lane wrote zeros: %s, EXEC-masked lanes wrote nothing: %s, transposing read errors %d (lane 17 got %08x %08x)\n",
               m0, bad, oob_zero ? "yes" : "NO", masked_kept ? "yes" : "NO", trbad, o[512 + 34], o[512 + 35]);
    }
    return 0;
}
