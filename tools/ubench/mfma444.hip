// v_mfma_f32_4x4x4_16b_bf16 on gfx950: operand maps, issue rate, and what shares a SIMD with what.  Development probe
// (VERDICT r2 "next round" items 2 and 3).  Build: hipcc -O3 --offload-arch=gfx950 mfma444.hip -o mfma444
//
//   part 1  lane maps of A, B and C/D, checked with exact integer data that is different in every block, row and column
//   part 2  cycles per instruction of ONE wave's stream on its SIMD: the MFMA back to back (one accumulator / four), the MFMA with
//           1..3 independent vector instructions behind it, the plain vector / LDS / global-load streams
//   part 3  two waves on one SIMD (waves w and w+4 of a 512-thread workgroup): stream A alone, stream B alone, A beside B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

static inline uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

// ---------------------------------------------------------------- part 1
__global__ void k_map(const s4* a, const s4* b, f4* d)
{
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    d[threadIdx.x] = c;
}

static int part1()
{
    // hypothesis: lane = 4*block + r.  A: lane holds A[i = r][k = 0..3];  B: lane holds B[k = 0..3][j = r];  D: lane holds D[i = 0..3][j = r]
    float A[16][4][4], B[16][4][4];
    for (int bl = 0; bl < 16; ++bl) for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) {
        A[bl][i][k] = (float)(1 + ((bl * 7 + i * 3 + k * 5) % 11));        // small integers: exact in bf16, products exact in f32
        B[bl][i][k] = (float)(1 + ((bl * 5 + i * 2 + k * 7) % 13));        // B[bl][k = i][j = k], asymmetric
    }
    std::vector<uint16_t> ha(64 * 4), hb(64 * 4);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        ha[l * 4 + e] = bf16_of(A[l / 4][l % 4][e]);
        hb[l * 4 + e] = bf16_of(B[l / 4][e][l % 4]);
    }
    s4 *da, *db; f4* dd; float hd[64][4];
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 1024);
    (void)hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad_h1 = 0, bad_h2 = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int bl = l / 4, r = l % 4;
        float h1 = 0.f, h2 = 0.f;
        for (int k = 0; k < 4; ++k) { h1 += A[bl][v][k] * B[bl][k][r]; h2 += A[bl][r][k] * B[bl][k][v]; }
        bad_h1 += hd[l][v] != h1; bad_h2 += hd[l][v] != h2;
    }
    printf("part 1  operand maps (lane = 4*block + r; A: row r, B: column r):\n");
    printf("  D register v of lane (block, r) = D[i = v][j = r]: %s (%d mismatches);  = D[i = r][j = v]: %s (%d)\n",
           bad_h1 ? "no" : "YES", bad_h1, bad_h2 ? "no" : "YES", bad_h2);
    if (bad_h1 && bad_h2) { printf("  lane 5 registers: %g %g %g %g\n", hd[5][0], hd[5][1], hd[5][2], hd[5][3]); }
    return 0;
}

// ---------------------------------------------------------------- streams
enum Role { NONE = 0, MFMA4, MFMA1, PKFMA, FMA, LDSB32, LDSB64, GLD16, MFMA_PK1, MFMA_PK2, MFMA_PK3, MFMA_FMA2, MFMA_PERM2, MFMA_CVT2,
            MFMA_LDS1, MFMA_GLD1, PKFMA_GLD1, PKFMA_LDS1, PERM, CVT, NROLES };
static const char* role_name[NROLES] = { "-", "mfma x4 accumulators", "mfma one accumulator", "v_pk_fma_f32", "v_fma_f32", "ds_read_b32", "ds_read_b64",
    "global_load_short_d16_hi", "mfma + 1 v_pk_fma_f32", "mfma + 2 v_pk_fma_f32", "mfma + 3 v_pk_fma_f32", "mfma + 2 v_fma_f32", "mfma + 2 v_perm_b32",
    "mfma + 2 v_cvt_pk_bf16_f32", "mfma + 1 ds_read_b64", "mfma + 1 global_load_short", "v_pk_fma_f32 + 1 global_load_short", "v_pk_fma_f32 + 1 ds_read_b64",
    "v_perm_b32", "v_cvt_pk_bf16_f32" };
// instructions per unrolled group, (matrix, other)
static const int role_n[NROLES][2] = { {0,0}, {8,0}, {8,0}, {0,8}, {0,8}, {0,8}, {0,8}, {0,8}, {8,8}, {8,16}, {8,24}, {8,16}, {8,16}, {8,16}, {8,8}, {8,8}, {0,16}, {0,16}, {0,8}, {0,8} };

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, acc, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)
#define PK(x) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(p2), "v"(q2))
#define FM(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(p1), "v"(q1))
#define PM(x) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x) : "v"(u0), "v"(u1), "v"(u2))
#define CV(x) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(p1), "v"(q1))
#define L64(x) asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(lds_addr))
#define L32(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lds_addr))
#define G16(x) asm volatile("global_load_short_d16_hi %0, %1, off" : "=v"(x) : "v"(gp))

template <int R>
__device__ __forceinline__ void stream(int iters, s4 a, s4 b, const uint16_t* gp, unsigned lds_addr, float* sink)
{
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f2 x0 = {1, 1}, x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0;
    f2 p2 = {1.0001f, 0.9999f}, q2 = {1e-9f, 1e-9f};
    float y0 = 1, y1 = 1, y2 = 1, y3 = 1, y4 = 1, y5 = 1, y6 = 1, y7 = 1, p1 = 1.0001f, q1 = 1e-9f;
    unsigned u0 = 0x12345678u, u1 = 0x9abcdef0u, u2 = 0x07060302u, w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;
    f2 l0, l1, l2, l3, l4, l5, l6, l7;
    float g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0;
    l0 = l1 = l2 = l3 = l4 = l5 = l6 = l7 = x0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (R == MFMA4)   { MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3); }
        if constexpr (R == MFMA1)   { MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); MF(c0); }
        if constexpr (R == PKFMA)   { PK(x0); PK(x1); PK(x2); PK(x3); PK(x4); PK(x5); PK(x6); PK(x7); }
        if constexpr (R == FMA)     { FM(y0); FM(y1); FM(y2); FM(y3); FM(y4); FM(y5); FM(y6); FM(y7); }
        if constexpr (R == PERM)    { PM(w0); PM(w1); PM(w2); PM(w3); PM(w4); PM(w5); PM(w6); PM(w7); }
        if constexpr (R == CVT)     { CV(w0); CV(w1); CV(w2); CV(w3); CV(w4); CV(w5); CV(w6); CV(w7); }
        if constexpr (R == LDSB32)  { L32(g0); L32(g1); L32(g2); L32(g3); L32(g4); L32(g5); L32(g6); L32(g7); asm volatile("s_waitcnt lgkmcnt(4)"); }
        if constexpr (R == LDSB64)  { L64(l0); L64(l1); L64(l2); L64(l3); L64(l4); L64(l5); L64(l6); L64(l7); asm volatile("s_waitcnt lgkmcnt(4)"); }
        if constexpr (R == GLD16)   { G16(g0); G16(g1); G16(g2); G16(g3); G16(g4); G16(g5); G16(g6); G16(g7); asm volatile("s_waitcnt vmcnt(24)"); }
        if constexpr (R == MFMA_PK1) { MF(c0); SB; PK(x0); SB; MF(c1); SB; PK(x1); SB; MF(c2); SB; PK(x2); SB; MF(c3); SB; PK(x3); SB;
                                       MF(c0); SB; PK(x4); SB; MF(c1); SB; PK(x5); SB; MF(c2); SB; PK(x6); SB; MF(c3); SB; PK(x7); SB; }
        if constexpr (R == MFMA_PK2) { MF(c0); SB; PK(x0); PK(x1); SB; MF(c1); SB; PK(x2); PK(x3); SB; MF(c2); SB; PK(x4); PK(x5); SB; MF(c3); SB; PK(x6); PK(x7); SB;
                                       MF(c0); SB; PK(x0); PK(x1); SB; MF(c1); SB; PK(x2); PK(x3); SB; MF(c2); SB; PK(x4); PK(x5); SB; MF(c3); SB; PK(x6); PK(x7); SB; }
        if constexpr (R == MFMA_PK3) { MF(c0); SB; PK(x0); PK(x1); PK(x2); SB; MF(c1); SB; PK(x3); PK(x4); PK(x5); SB; MF(c2); SB; PK(x6); PK(x7); PK(x0); SB; MF(c3); SB; PK(x1); PK(x2); PK(x3); SB;
                                       MF(c0); SB; PK(x4); PK(x5); PK(x6); SB; MF(c1); SB; PK(x7); PK(x0); PK(x1); SB; MF(c2); SB; PK(x2); PK(x3); PK(x4); SB; MF(c3); SB; PK(x5); PK(x6); PK(x7); SB; }
        if constexpr (R == MFMA_FMA2) { MF(c0); SB; FM(y0); FM(y1); SB; MF(c1); SB; FM(y2); FM(y3); SB; MF(c2); SB; FM(y4); FM(y5); SB; MF(c3); SB; FM(y6); FM(y7); SB;
                                        MF(c0); SB; FM(y0); FM(y1); SB; MF(c1); SB; FM(y2); FM(y3); SB; MF(c2); SB; FM(y4); FM(y5); SB; MF(c3); SB; FM(y6); FM(y7); SB; }
        if constexpr (R == MFMA_PERM2) { MF(c0); SB; PM(w0); PM(w1); SB; MF(c1); SB; PM(w2); PM(w3); SB; MF(c2); SB; PM(w4); PM(w5); SB; MF(c3); SB; PM(w6); PM(w7); SB;
                                         MF(c0); SB; PM(w0); PM(w1); SB; MF(c1); SB; PM(w2); PM(w3); SB; MF(c2); SB; PM(w4); PM(w5); SB; MF(c3); SB; PM(w6); PM(w7); SB; }
        if constexpr (R == MFMA_CVT2) { MF(c0); SB; CV(w0); CV(w1); SB; MF(c1); SB; CV(w2); CV(w3); SB; MF(c2); SB; CV(w4); CV(w5); SB; MF(c3); SB; CV(w6); CV(w7); SB;
                                        MF(c0); SB; CV(w0); CV(w1); SB; MF(c1); SB; CV(w2); CV(w3); SB; MF(c2); SB; CV(w4); CV(w5); SB; MF(c3); SB; CV(w6); CV(w7); SB; }
        if constexpr (R == MFMA_LDS1) { MF(c0); SB; L64(l0); SB; MF(c1); SB; L64(l1); SB; MF(c2); SB; L64(l2); SB; MF(c3); SB; L64(l3); SB;
                                        MF(c0); SB; L64(l4); SB; MF(c1); SB; L64(l5); SB; MF(c2); SB; L64(l6); SB; MF(c3); SB; L64(l7); SB; asm volatile("s_waitcnt lgkmcnt(4)"); }
        if constexpr (R == MFMA_GLD1) { MF(c0); SB; G16(g0); SB; MF(c1); SB; G16(g1); SB; MF(c2); SB; G16(g2); SB; MF(c3); SB; G16(g3); SB;
                                        MF(c0); SB; G16(g4); SB; MF(c1); SB; G16(g5); SB; MF(c2); SB; G16(g6); SB; MF(c3); SB; G16(g7); SB; asm volatile("s_waitcnt vmcnt(24)"); }
        if constexpr (R == PKFMA_GLD1) { PK(x0); G16(g0); PK(x1); G16(g1); PK(x2); G16(g2); PK(x3); G16(g3); PK(x4); G16(g4); PK(x5); G16(g5); PK(x6); G16(g6); PK(x7); G16(g7);
                                         asm volatile("s_waitcnt vmcnt(24)"); }
        if constexpr (R == PKFMA_LDS1) { PK(x0); L64(l0); PK(x1); L64(l1); PK(x2); L64(l2); PK(x3); L64(l3); PK(x4); L64(l4); PK(x5); L64(l5); PK(x6); L64(l6); PK(x7); L64(l7);
                                         asm volatile("s_waitcnt lgkmcnt(4)"); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    f2 xs = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + l0 + l1 + l2 + l3 + l4 + l5 + l6 + l7;
    float s = c0.x + c1.y + c2.z + c3.w + xs.x + xs.y + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7
            + (float)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7);
    if (s == 123.456f) *sink = s;
}

// waves 0..3 of a workgroup run stream RA, waves 4..7 stream RB (waves w and w+4 share a SIMD); 256-thread launch: RA only
template <int RA, int RB>
__global__ __launch_bounds__(1024) void k_pair(int iters, const s4* ab, const uint16_t* g, float* sink, unsigned long long* cyc)
{
    __shared__ float lds[4096];
    lds[threadIdx.x] = 1.0f; lds[threadIdx.x + 1024] = 1.f; lds[threadIdx.x + 2048] = 1.f; lds[threadIdx.x + 3072] = 1.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const s4 a = ab[lane], b = ab[64 + lane];
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds) + (unsigned)lane * 8u;   // conflict-free 8-byte reads
    const uint16_t* gp = g + lane;                                                                                        // 128 contiguous bytes per wave, L1-resident
    const bool roleA = ((wave >> 2) & 1) == 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (roleA) stream<RA>(iters, a, b, gp, lds_addr, sink); else stream<RB>(iters, a, b, gp, lds_addr, sink);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

struct Ctx { s4* ab; uint16_t* g; float* sink; unsigned long long* cyc; };

template <int RA, int RB>
static void run_pair(const Ctx& c, int threads, int blocks, int iters, double* cycA, double* cycB)
{
    std::vector<unsigned long long> h(16 * blocks);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_pair<RA, RB>), dim3(blocks), dim3(threads), 0, 0, iters, c.ab, c.g, c.sink, c.cyc);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), c.cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0; int na = 0, nb = 0;
    const int waves = threads / 64;
    for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < waves; ++w) {
        if (((w >> 2) & 1) == 0) { a += (double)h[bl * 16 + w]; ++na; } else { b += (double)h[bl * 16 + w]; ++nb; }
    }
    *cycA = na ? a / na / iters : 0; *cycB = nb ? b / nb / iters : 0;
}

template <int R>
static void line_single(const Ctx& c, int blocks)
{
    const int iters = 2000;
    double a1, b1, a2, b2, a4, b4;
    run_pair<R, R>(c, 256, blocks, iters, &a1, &b1);
    run_pair<R, R>(c, 512, blocks, iters, &a2, &b2);
    run_pair<R, R>(c, 1024, blocks, iters, &a4, &b4);
    const int nm = role_n[R][0], no = role_n[R][1];
    const double per = nm ? nm : no;      // per matrix instruction when the stream has any, else per instruction
    printf("  %-36s group = %2d mfma + %2d other | cycles per %s: 1 wave/SIMD %6.2f   2 waves %6.2f (per SIMD %6.2f)   4 waves %6.2f (per SIMD %6.2f)\n",
           role_name[R], nm, no, nm ? "mfma" : "instr", a1 / per, (a2 + b2) / 2 / per, (a2 + b2) / 4 / per, (a4 + b4) / 2 / per, (a4 + b4) / 8 / per);
}

template <int RA, int RB>
static void line_pair(const Ctx& c, int blocks)
{
    const int iters = 2000;
    double aa, xx, bb, yy, pa, pb;
    run_pair<RA, NONE>(c, 512, blocks, iters, &aa, &xx);      // A with an empty partner
    run_pair<NONE, RB>(c, 512, blocks, iters, &yy, &bb);      // B with an empty partner
    run_pair<RA, RB>(c, 512, blocks, iters, &pa, &pb);
    const double na = role_n[RA][0] + role_n[RA][1], nb = role_n[RB][0] + role_n[RB][1];
    printf("  A = %-28s B = %-28s | cycles per group of %2.0f / %2.0f instr: A alone %7.1f  B alone %7.1f  A beside B %7.1f  B beside A %7.1f   (sum %7.1f, max %7.1f)\n",
           role_name[RA], role_name[RB], na, nb, aa, bb, pa, pb, aa + bb, aa > bb ? aa : bb);
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;
    part1();
    Ctx c;
    std::vector<uint16_t> hab(128 * 4);
    for (size_t i = 0; i < hab.size(); ++i) hab[i] = bf16_of((float)((i * 7) % 5) * 0.25f);
    std::vector<uint16_t> hg(4096, bf16_of(0.5f));
    (void)hipMalloc(&c.ab, 1024); (void)hipMalloc(&c.g, 8192); (void)hipMalloc(&c.sink, 64); (void)hipMalloc(&c.cyc, 16 * 8 * (size_t)blocks);
    (void)hipMemcpy(c.ab, hab.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(c.g, hg.data(), 8192, hipMemcpyHostToDevice);

    printf("part 2  one stream per wave, %d workgroup(s); s_memtime cycles\n", blocks);
    line_single<MFMA4>(c, blocks);   line_single<MFMA1>(c, blocks);  line_single<PKFMA>(c, blocks);    line_single<FMA>(c, blocks);
    line_single<PERM>(c, blocks);    line_single<CVT>(c, blocks);
    line_single<LDSB32>(c, blocks);  line_single<LDSB64>(c, blocks); line_single<GLD16>(c, blocks);
    line_single<MFMA_PK1>(c, blocks); line_single<MFMA_PK2>(c, blocks); line_single<MFMA_PK3>(c, blocks); line_single<MFMA_FMA2>(c, blocks);
    line_single<MFMA_PERM2>(c, blocks); line_single<MFMA_CVT2>(c, blocks); line_single<MFMA_LDS1>(c, blocks); line_single<MFMA_GLD1>(c, blocks);
    line_single<PKFMA_GLD1>(c, blocks); line_single<PKFMA_LDS1>(c, blocks);

    printf("part 3  two waves on one SIMD (512-thread workgroup: waves 0-3 run A, waves 4-7 run B)\n");
    line_pair<PKFMA, GLD16>(c, blocks);
    line_pair<PKFMA, LDSB32>(c, blocks);
    line_pair<PKFMA, LDSB64>(c, blocks);
    line_pair<PKFMA, PKFMA>(c, blocks);
    line_pair<FMA, FMA>(c, blocks);
    line_pair<MFMA4, PKFMA>(c, blocks);
    line_pair<MFMA4, FMA>(c, blocks);
    line_pair<MFMA4, MFMA4>(c, blocks);
    line_pair<MFMA4, GLD16>(c, blocks);
    line_pair<MFMA4, LDSB64>(c, blocks);
    line_pair<MFMA_PK2, MFMA_PK2>(c, blocks);
    line_pair<MFMA_PK1, GLD16>(c, blocks);
    return 0;
}
