// What does a D16 load leave in the OTHER half of its destination register on gfx950 (SRAM-ECC on)?  Development probe.
// The compiler will not select *_d16_hi loads here (it assumes the other half is not preserved); if the hardware zero-fills it,
// a bf16 element can be loaded straight into float32 position (no shift) from global memory and from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_probe(const uint16_t* src, uint32_t* out)
{
    __shared__ uint16_t sm[64];
    sm[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    uint32_t a = 0xAAAAAAAAu, b = 0xBBBBBBBBu, c = 0xCCCCCCCCu, d = 0xDDDDDDDDu;
    const uint16_t* p = src + threadIdx.x;
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)(sm + threadIdx.x);
    asm volatile("global_load_short_d16_hi %0, %4, off\n"
                 "global_load_short_d16 %1, %4, off\n"
                 "ds_read_u16_d16_hi %2, %5\n"
                 "ds_read_u16_d16 %3, %5\n"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(p), "v"(la) : "memory");
    out[threadIdx.x * 4 + 0] = a; out[threadIdx.x * 4 + 1] = b; out[threadIdx.x * 4 + 2] = c; out[threadIdx.x * 4 + 3] = d;
}

int main()
{
    uint16_t h[64]; for (int i = 0; i < 64; ++i) h[i] = (uint16_t)(0x1200 + i);
    uint16_t* ds; uint32_t* dout; uint32_t o[256];
    (void)hipMalloc(&ds, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    (void)hipMemcpy(ds, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, ds, dout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("element 0x1203 loaded into registers preset to 0xAAAAAAAA / 0xBBBBBBBB / 0xCCCCCCCC / 0xDDDDDDDD:\n");
    printf("  global_load_short_d16_hi -> %08x\n  global_load_short_d16    -> %08x\n  ds_read_u16_d16_hi       -> %08x\n  ds_read_u16_d16          -> %08x\n",
           o[12], o[13], o[14], o[15]);
    printf("other half is %s\n", (o[12] & 0xffff) == 0 ? "ZERO-FILLED" : ((o[12] & 0xffff) == 0xAAAA ? "PRESERVED" : "something else"));
    return 0;
}
