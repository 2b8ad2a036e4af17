// Public side of the channel-per-lane tiled RecConv2d kernels (rcx_cpt_kernel.h) and the 56x56 / level 4 instantiations; the
// 28x28 / level 3 ones are compiled in rcx_cpt2.hip (two translation units so that they build in parallel).
#include "rcx_cpt_kernel.h"
#include "rcx_opts.h"

namespace rcx {
namespace cpt {
hipError_t launch_t2(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s, const SavedPyr& sv);
hipError_t launch_lv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int H, int mode, int dtype, hipStream_t s);   // rcx_cpt3.hip
hipError_t launch_ts16(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s);        // rcx_cpt4.hip
}  // namespace cpt

// RCX_CPT=0 gives both blocks back to the banded lanes kernels.  The 28x28 block with a channel count that is not a multiple of 64
// (RecNeXt-M0/M1/M2/M5) stays on the banded kernel by default; RCX_CPT=32 puts it on 32-channel workgroups (k_recconv_cpt<2, 2, ...>, four
// per CU).  Measured (profiles/archive/r03_cpt_28_ragged.txt, bf16, us, 32-channel workgroups / banded): 256 x 112 (1 024 units) 51.9 / 57.4,
// 256 x 96 (768) 45.3 / 46.9 (inside RecNeXt-M1: 46.7 / 47.0), 256 x 80 46.8 / 46.9, 128 x 160 (640) 41.7 / 42.6 -- but with fewer units a
// unit's own ~37 us of phases is the floor (128 x 96: 37.8 / 28.9, 64 x 160: 37.3 / 27.8) and with more the second round runs on a fraction
// of the chip (256 x 160: 63.0 / 67.3 alone, 75.6 / 68.7 inside RecNeXt-M5; 512 x 96: 84.0 / 79.3).  A rule on the unit count would pick a
// different summation order for a batch and for its shards (the two kernels agree to float32 round-off, not bit for bit), and "a batch
// shard gives the same rows"  is kept exact: the choice between kernels that are NOT bit-identical never depends on N.  (cb16() below
// and the reload form of rcx_cpl14.hip do look at N -- they choose between variants that are bit-identical by construction and by test:
// test_56_block_with_16_and_32_channel_workgroups_is_the_same_function, test_14_block_reload_form_is_the_same_function.)
static bool cpt28_ragged(int N, int C)
{
    (void)N; (void)C;
    const char* v = rcx::opt::value(rcx::opt::CPT);
    return v && *v == '3';
}

bool cpt_applicable(int N, int C, int H, int W, int level, int k, int dtype)
{
    if (!cpt::enabled() || k != 5 || C < 1 || !(dtype == 0 || dtype == 1 || dtype == 2)) return false;
    if (H == 56 && W == 56 && level == 4) return true;
    if (H == 28 && W == 28 && level == 3) return C % 64 == 0 || cpt28_ragged(N, C);
    // 64 x 64 / level 3 on 16-pixel tiles (round 5; RCX_CPT16=0: the banded lanes kernel).  Chosen by the plane alone, never by N: a unit's chain of
    // phases (~40 us) is shorter than the lanes kernel's one wave per four channel planes (~58 us) at every batch size
    if (H == 64 && W == 64 && level == 3) return !rcx::opt::is_zero(rcx::opt::CPT16);
    // one level less (round 3): stages 1 and 2 of a 448 x 448 input, inner blocks of the nested schedule; RCX_CPT=full: not these
    const char* v = rcx::opt::value(rcx::opt::CPT);
    if (v && *v == 'f') return false;
    if (H == 56 && W == 56 && level == 3) return true;
    if (H == 28 && W == 28 && level == 2) return true;
    return false;
}

// the training forward has the bilinear, whole-block instantiations only
bool cpt_train_applicable(int N, int C, int H, int W, int level, int k, int mode, int dtype)
{
    return mode == 0 && H != 64 && cpt_applicable(N, C, H, W, level, k, dtype) && !(H == 28 && C % 64 != 0) && level == (H == 56 ? 4 : 3);
}

int cpt_describe(int N, int C, int H, int level, int mode, int dtype, char* buf, int len)
{
    if (H == 64)
        return snprintf(buf, len, "cpt(k_recconv_cpt<4, 4, %d, 0, ts=16>,cb=16,nt=256,units=%d,lds=%d)", mode, N * ((C + 15) / 16), cpt::Geo<4, 4, 0, float, 3, 0, 16>::LDS_BYTES);
    const int T = H / 14, halves = T == 4 ? (cpt::cb16(N, C) ? 4 : 2) : (C % 64 != 0 ? 2 : 1), pixf = 64 / halves;
    const bool full = level == (T == 4 ? 4 : 3);
    const int pixb = full && C == (T == 4 ? 64 : 128) ? C * (dtype == 0 ? 4 : 2) : 0;
    const int total = N * ((C + pixf - 1) / pixf);
    return snprintf(buf, len, full ? "cpt(k_recconv_cpt<%d, %d, %d, %d>,cb=%d,nt=%d,units=%d,lds=%d)" : "cpt(k_recconv_cpt<%d, %d, %d, %d>,levels-1,cb=%d,nt=%d,units=%d,lds=%d)",
                    T, halves, mode, pixb, pixf, T * T / halves * 64, total,
                    T == 4 ? (halves == 4 ? cpt::Geo<4, 4, 0, float>::LDS_BYTES : cpt::Geo<4, 2, 0, float>::LDS_BYTES)
                           : (halves == 2 ? cpt::Geo<2, 2, 0, float>::LDS_BYTES : cpt::Geo<2, 1, 0, float>::LDS_BYTES));
}

hipError_t cpt_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int H, int level, int mode, int dtype, hipStream_t s,
                       float* saved, const size_t* f_off, const size_t* c_off)
{
    if (H == 64) return saved ? hipErrorInvalidConfiguration : cpt::launch_ts16(x, y, wpack, bpack, N, C, mode, dtype, s);
    if (level != (H == 56 ? 4 : 3)) return saved ? hipErrorInvalidConfiguration : cpt::launch_lv(x, y, wpack, bpack, N, C, H, mode, dtype, s);
    cpt::SavedPyr sv{};
    sv.base = saved;
    if (saved)
        for (int l = 1; l <= (H == 56 ? 4 : 3); ++l) { sv.f_off[l] = f_off[l]; sv.c_off[l] = c_off[l]; }
    if (H == 56) return (!saved && cpt::cb16(N, C)) ? cpt::launch_md<4, 4>(x, y, wpack, bpack, N, C, mode, dtype, s, sv)
                                               : cpt::launch_md<4, 2>(x, y, wpack, bpack, N, C, mode, dtype, s, sv);
    return cpt::launch_t2(x, y, wpack, bpack, N, C, mode, dtype, s, sv);
}

}  // namespace rcx
