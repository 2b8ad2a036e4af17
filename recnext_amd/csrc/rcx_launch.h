// Host-side launcher prototypes shared between the kernel translation units and rcx_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>

// Raise a kernel's dynamic-LDS limit once per kernel instantiation AND DEVICE (the attribute belongs to the device's copy of the
// function; the statics live in the expanding template function), not per launch; a later launch that needs more raises it again.
// Atomics: two host threads may launch the same kernel; the worst case is one redundant hipFuncSetAttribute call.
#include <atomic>
#define RCX_MAX_DEVICES 64
#define RCX_SET_LDS_ONCE(kfn, bytes)                                                                                              \
    do {                                                                                                                          \
        static std::atomic<size_t> rcx_lds_set_[RCX_MAX_DEVICES];                                                                 \
        int rcx_dev_ = 0;                                                                                                         \
        if (hipGetDevice(&rcx_dev_) != hipSuccess || rcx_dev_ < 0 || rcx_dev_ >= RCX_MAX_DEVICES) rcx_dev_ = RCX_MAX_DEVICES - 1; \
        if ((size_t)(bytes) > 64 * 1024 && (size_t)(bytes) > rcx_lds_set_[rcx_dev_].load(std::memory_order_acquire)) {            \
            hipError_t rcx_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (rcx_e_ != hipSuccess) return rcx_e_;                                                                              \
            if (rcx_dev_ != RCX_MAX_DEVICES - 1) rcx_lds_set_[rcx_dev_].store((size_t)(bytes), std::memory_order_release);        \
        }                                                                                                                         \
    } while (0)

namespace rcx {

// rcx_time_next_launch(): a pair of HIP events the caller wants recorded by the command processor AT the start and the end of the next fused-block
// kernel (hipExtLaunchKernelGGL), not around its launch: an event pair recorded on the stream brackets the dispatch gaps too (+2.5 .. 4 us on a
// ~100 us kernel) and delays the next kernel.  Thread-local, consumed by the first launch that supports it, cleared when rcx_recconv2d_fwd returns.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
LaunchEvents take_launch_events();
#define RCX_LAUNCH_TIMED(kfn, grid, block, lds, stream, ...)                                                                       \
    do {                                                                                                                           \
        const rcx::LaunchEvents rcx_ev_ = rcx::take_launch_events();                                                               \
        if (rcx_ev_.start || rcx_ev_.stop) hipExtLaunchKernelGGL(kfn, grid, block, lds, stream, rcx_ev_.start, rcx_ev_.stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kfn, grid, block, lds, stream, __VA_ARGS__);                                                       \
    } while (0)

// rcx_generic.hip
hipError_t generic_dwconv(const void* x, void* y, const float* w, const float* b,
                          int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt, hipStream_t s);
hipError_t generic_upadd_dwconv(const void* x, const void* coarse, void* y, const float* w, const float* b,
                                int N, int C, int H, int W, int Hc, int Wc, int k, int mode,
                                int x_dt, int c_dt, int out_dt, hipStream_t s);
hipError_t generic_dwconv_mult2(const void* x, void* y, const float* w, const float* b,
                                int N, int Cin, int H, int W, int k, int stride, int dt, hipStream_t s);
hipError_t pack_dw_weight(const void* w, float* dst, int C, int k, int dt, hipStream_t s);
hipError_t selftest_d16(const void* src, void* flag, hipStream_t s);      // the D16-hi zero-fill the 16-bit load paths rely on
hipError_t pack_bias(const void* b, float* dst, int C, int dt, hipStream_t s);
struct PackPtrs { const void* w[10]; const void* b[10]; };         // RCX_MAX_LEVEL + 2 convs
hipError_t pack_params(const PackPtrs& P, float* wpack, float* wflip, float* bpack, int count, int C, int k, int dt, hipStream_t s);
hipError_t unpack_grads(const float* gwpack, const PackPtrs& P, int count, int C, int k, hipStream_t s);

// rcx_plane.hip -- fused single-launch schedule (k=5, C%8==0, pyramid fits in LDS)
bool plane_applicable(int N, int C, int H, int W, int level, int k, int dtype);
int plane_describe(int N, int C, int H, int W, int level, int k, int dtype, char* buf, int len);
hipError_t plane_recconv(const void* x, void* y, const float* wpack, const float* bpack,
                         int N, int C, int H, int W, int level, int k, int mode, int dtype, hipStream_t s);

// rcx_lanes.hip -- register-resident schedule for the 7*2^k planes (k=5, natural level)
bool lanes_applicable(int N, int C, int H, int W, int level, int k, int dtype);
int lanes_describe(int N, int C, int H, int W, int level, int k, int mode, int dtype, char* buf, int len);
hipError_t lanes_recconv(const void* x, void* y, const float* wpack, const float* bpack,
                         int N, int C, int H, int W, int level, int k, int mode, int dtype, hipStream_t s);

// rcx_lanes16.hip -- the same schedule on the 16*2^k planes (called by lanes_recconv)
hipError_t lanes16_recconv(const void* x, void* y, const float* wpack, const float* bpack,
                           int N, int C, int H, int W, int level, int k, int mode, int dtype, hipStream_t s);


// channel-per-lane kernel of the 14x14 / level 2 block (rcx_cpl14.hip): any channel count
bool cpl14_applicable(int N, int C, int H, int W, int level, int k, int dtype);
bool cpl14_short_applicable(int N, int C, int H, int W, int level, int k, int dtype);       // 14 x 14 / level 1, inference
int cpl14_short_describe(int N, int C, int mode, char* buf, int len);
hipError_t cpl14_short_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s);
int cpl14_describe(int N, int C, int mode, int dtype, char* buf, int len);
// saved != nullptr (training forward): the launch also writes the float32 pyramid F_l at saved + f_off[l], C_l at saved + c_off[l] (bytes, l >= 1)
hipError_t cpl14_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s,
                         float* saved = nullptr, const size_t* f_off = nullptr, const size_t* c_off = nullptr);

// one step, channel per lane: y = conv5(x + resize(coarse)) + bias on the 14x14 plane (RecAttn2d's fused kernel; rcx_cpl14.hip)
// rcx_upcpt.hip -- conv5(x + resize2x(coarse)) on any even plane of at least 28 x 28 (per-row descriptors; ragged last tiles): channel per
// lane, tiled, no LDS (round 3)
bool upadd_cpt_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt);
int upadd_cpt_describe(int N, int C, int H, int W, int mode, int x_dt, char* buf, int len);
hipError_t upadd_cpt(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, int H, int W, int mode,
                     int x_dt, int c_dt, hipStream_t s);
bool down7m2_cpt_applicable(int N, int Cin, int H, int W, int k, int stride, int dtype);
hipError_t down7m2_cpt(const void* x, void* y, const float* w, const float* b, int N, int Cin, int H, int W, int dtype, hipStream_t s);
bool down5_cpt_applicable(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt);
hipError_t down5_cpt(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int in_dt, int out_dt, hipStream_t s);
bool upadd_cpl14_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt);
hipError_t upadd_cpl14(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, int H, int mode, int x_dt, int c_dt,
                       hipStream_t s);            // H = 14 (coarse 7 x 7) or 7 (coarse 4 x 4)
// the stride-2 conv5 of the 7 x 7 plane (and of the 14 x 14 plane of 16-bit activations), float32 out (round 4)
bool down5_cpl7_applicable(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt);
hipError_t down5_cpl7(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int in_dt, hipStream_t s);

bool cpl7b_applicable(int N, int C, int H, int W, int level, int k, int dtype);
int cpl7b_describe(int N, int C, int mode, char* buf, int len);
hipError_t cpl7b_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int mode, int dtype, hipStream_t s,
                         float* saved = nullptr, const size_t* f_off = nullptr, const size_t* c_off = nullptr);

// channel-per-lane backward of the 14x14 / level 2 and 7x7 / level 1 blocks in one launch (rcx_cplbwd.hip).  part[j]: one row of
// (25 + 1) * C partial sums per image for job j (0 = the shared down conv, 1 + j = convs[j]), reduced by bwd_wgrad_reduce_jobs
bool cplbwd_applicable(int N, int C, int H, int W, int level, int k, int dtype);
hipError_t cplbwd_recconv(const void* x, const void* gy, const float* wpack, const float* wflip, const void* saved,
                          const size_t* f_off, const size_t* c_off, void* gx, float* const* part,
                          int N, int C, int H, int level, int mode, int dtype, hipStream_t s, int gy_dt = 0);      // gy_dt: 0 float32 | 1 bfloat16 (with dtype 1)

// tiled channel-per-lane weight gradient of a stride-1 5x5 conv over T = a + R(coarse) on the 56x56 / 28x28 planes (rcx_cplwgrad.hip):
// one partial row of (25 + 1) * C sums per (image, 14-row band)
bool wgrad_cpl_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int stride, bool has_coarse);
hipError_t wgrad_cpl(const void* a, int a_dt, const float* coarse, const void* g, int g_dt, float* partial, int N, int C, int H, int mode,
                     hipStream_t s, int* rows_out);            // g_dt: 0 = float32, or a_dt (a 16-bit gradient of a's own type)

// ... and of the shared stride-2 5x5 conv (a: H x W, g: H/2 x W/2): one partial row per (image, 14-row band of g)
bool wgrad2_cpl_applicable(int N, int C, int H, int W, int Ho, int Wo, int k, int stride, bool has_coarse);
hipError_t wgrad2_cpl(const void* a, int a_dt, const float* g, float* partial, int N, int C, int H, hipStream_t s, int* rows_out);
// ... and of the Downsample conv (7x7, stride 2, channel multiplier 2; Cout = 2 Cin channels of g and of the weight)
bool wgrad2m_cpl_applicable(int N, int Cout, int H, int W, int k);
hipError_t wgrad2m_cpl(const void* a, int a_dt, const float* g, float* partial, int N, int Cout, int H, hipStream_t s, int* rows_out);

// tiled channel-per-lane adjoints of the fine levels on the 56x56 / 28x28 planes (rcx_cptbwd.hip): gC = R^T (K^ g) at half resolution, and
// out = K^ g + D^T G (G = nullptr: the conv adjoint alone); g_dt / out_dt: dtype ids, G and gC float32
bool bwd_cpt_applicable(int N, int C, int H, int W, int k);
hipError_t bwd_gc_cpt(const void* g, int g_dt, float* gC, const float* wf, int N, int C, int H, int mode, hipStream_t s);
bool bwd_wgrad_dm_cpt_applicable(int N, int Cout, int H, int W, int k);
hipError_t bwd_wgrad_dm_cpt(const void* a, int a_dt, const float* G, float* partial, int N, int Cout, int H, hipStream_t s, int* rows_out);
bool bwd_down7m2_cpt_applicable(int N, int Cin, int H, int W, int k);
hipError_t bwd_down7m2_cpt(const float* g, void* gx, int x_dt, const float* w, int N, int Cin, int H, hipStream_t s);
hipError_t bwd_wgrad_k_cpt(const void* a, int a_dt, const float* coarse, const void* g, int g_dt, float* partial, int N, int C, int H, int mode, hipStream_t s,
                           int* rows_out);
hipError_t bwd_wgrad_d_cpt(const void* a, int a_dt, const float* G, float* partial, int N, int C, int H, hipStream_t s, int* rows_out);
hipError_t bwd_dT_cpt(const float* G, void* out, int out_dt, const float* wd, int N, int C, int H, hipStream_t s);       // out = D^T G alone
hipError_t bwd_gx_cpt(const void* g, int g_dt, const float* G, void* out, int out_dt, const float* wf, const float* wd, int N, int C, int H, hipStream_t s);

// channel-per-lane, tiled kernel of the 56x56 / level 4 and 28x28 / level 3 blocks (rcx_cpt.hip): any channel count
bool cpt_applicable(int N, int C, int H, int W, int level, int k, int dtype);
int cpt_describe(int N, int C, int H, int level, int mode, int dtype, char* buf, int len);
bool cpt_train_applicable(int N, int C, int H, int W, int level, int k, int mode, int dtype);
hipError_t cpt_recconv(const void* x, void* y, const float* wpack, const float* bpack, int N, int C, int H, int level, int mode, int dtype, hipStream_t s,
                       float* saved = nullptr, const size_t* f_off = nullptr, const size_t* c_off = nullptr);
// rcx_down.hip -- register-resident depthwise 7x7 stride-2 conv with channel multiplier 2 (Downsample) on the 7*2^k planes
bool down_lanes_applicable(int N, int Cin, int H, int W, int k, int stride, int dtype);
hipError_t down_lanes(const void* x, void* y, const float* w, const float* b, int N, int Cin, int H, int W, int k, int stride, int dtype, hipStream_t s);

// rcx_upadd.hip -- register-resident single steps on the 7*2^k planes: conv5(x + resize(coarse)) and the stride-2 conv5
bool upadd_lanes_applicable(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt);
hipError_t upadd_lanes(const void* x, const void* coarse, void* y, const float* w, const float* b,
                       int N, int C, int H, int W, int mode, int x_dt, int c_dt, hipStream_t s);
bool conv5_lanes_applicable(int N, int C, int H, int W, int k, int x_dt, int out_dt);
hipError_t conv5_lanes(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int x_dt, hipStream_t s);
bool down5_lanes_applicable(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt);
hipError_t down5_lanes(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int in_dt, int out_dt, hipStream_t s);

// rcx_attn.hip -- linear-attention core of RecAttn2d (after the qk projection)
hipError_t linattn_core(const void* qpre, const void* kpre, const void* v, const void* pe, void* out,
                        int B, int n, int C, int heads, int dtype, hipStream_t s, const float* pew = nullptr, const float* peb = nullptr, int Wp = 0);
bool linattn_core_fuses_pe(int n, int C, int heads, int dtype);
// rcx_qkcore.hip -- the qk projection + the core + pe on the matrix cores, 16-bit-activation callers: one launch (a workgroup per image, a wave
// per head) up to 64 tokens, two launches (k^T v partial sums in the workspace, then the outputs) above
bool recattn_qkcore_applicable(int B, int Hp, int Wp, int C, int heads);
int recattn_qkcore_launches(int B, int Hp, int Wp, int C, int heads);          // 0: no kernel for this shape, 1 / 2: launches
size_t recattn_qkcore_workspace_bytes(int B, int Hp, int Wp, int C, int heads);
// ... and with RecAttn2d's stride-2 conv inside (x = the 14 x 14 / 7 x 7 plane of 16-bit activations): one launch from x to the attention output
bool recattn_down_qkcore_applicable(int B, int H, int W, int C, int heads, int x_dt);
hipError_t recattn_down_qkcore(const void* x, const float* wdn, const float* bdn, const void* wqk_bf16, const float* bqk, const float* wpe, const float* bpe,
                               float* out, int B, int H, int C, int heads, int x_dt, hipStream_t s);
// ... and RecAttn2d.forward whole (nearest resize): + the final conv(x + resize(a)), one launch from x to y
// rcx_stem.hip: RecNextStem (conv3x3 s2 + GELU + conv3x3 s2) in one launch (bf16)
bool stem_applicable(int N, int H, int W, int CM, int CO, int dtype);
size_t stem_pack_bytes(int CM, int CO);
hipError_t stem_fwd(const void* x, void* y, const void* w1frag, const float* b1, const void* w2frag, const float* b2, int N, int H, int W, int CM, int CO, int dtype, hipStream_t s);
// rcx_mlp.hip: the channel mixer + residual of a block in one launch (bf16)
bool channel_mlp_applicable(int M, int C, int H, int dtype);
size_t channel_mlp_pack_bytes(int C, int H);
hipError_t channel_mlp(const void* z, const void* x, void* y, const void* wfrag, const float* bias, int M, int C, int H, int dtype, hipStream_t s);
bool recattn2d_unit_applicable(int B, int H, int W, int C, int heads, int x_dt, int mode);
hipError_t recattn2d_unit(const void* x, const float* wdn, const float* bdn, const void* wqk_bf16, const float* bqk, const float* wpe, const float* bpe,
                          const float* wcv, const float* bcv, void* y, int B, int H, int C, int heads, int x_dt, hipStream_t s);
hipError_t recattn_qkcore(const float* d, const void* wqk_bf16, const float* bqk, const float* wpe, const float* bpe, float* out, void* workspace,
                          int B, int Hp, int Wp, int C, int heads, hipStream_t s);

hipError_t linattn_core_bwd(const void* qpre, const void* kpre, const void* v, const void* gout, void* gq, void* gk, void* gv,
                            int B, int n, int C, int heads, int dtype, hipStream_t s);

// rcx_bwd.hip -- backward pieces (deterministic gathers + two-stage weight-gradient reduction)
size_t wgrad_partial_bytes(int C, int k);
hipError_t bwd_wgrad(const void* a, int a_dt, const float* coarse, const float* g, float* partial, float* gw, float* gb,
                     int N, int C, int H, int W, int Hc, int Wc, int Ho, int Wo, int k, int stride, int mode, int accumulate, hipStream_t s,
                     int* rows_out = nullptr,    // rows_out: leave the partial sums in `partial` (that many rows) for bwd_wgrad_reduce_jobs
                     int g_dt = 0);              // g of a's own 16-bit type: only where the tiled kernel runs (wgrad_cpl_applicable), else an error
// every weight gradient of one block's backward reduced in one launch: job j sums nslots[j] partial buffers (rows[j][.] rows each) into gw[j] / gb[j]
struct WgradJobs {
    int njobs, kk, C;
    int nslots[10];
    int rows[10][8];
    const float* part[10][8];
    float* gw[10];
    float* gb[10];
    // param_layout 0: gw[j] is a row of the packed (k*k, C) float32 gradient, gb[j] (C) float32.  1: gw[j] / gb[j] are the PARAMETERS' gradients
    // themselves, (C, 1, k, k) contiguous / (C), elements of dtype id param_dt (rcx_recconv2d_bwd's gw_out / gb_out)
    int param_layout, param_dt;
};
hipError_t bwd_wgrad_reduce_jobs(const WgradJobs& J, hipStream_t s);
hipError_t bwd_mult2(const void* x, int x_dt, const float* g, const float* w, void* gx, float* partial, float* gw, float* gb,
                     int N, int Cin, int H, int W, int k, hipStream_t s);
hipError_t bwd_down_input(const float* base, const float* g, void* out, int out_dt, const float* w,
                          int N, int C, int H, int W, int Hc, int Wc, int k, hipStream_t s);
hipError_t bwd_resize(const float* gfine, float* gcoarse, int N, int C, int H, int W, int Hc, int Wc, int mode, hipStream_t s);

}  // namespace rcx
