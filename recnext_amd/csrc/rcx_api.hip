// C ABI of librecnext_amd.so (include/recnext_amd.h): argument checking, schedule selection,
// error reporting.  No allocation, no synchronisation, no retained pointers.
#include "../../include/recnext_amd.h"
#include "rcx_opts.h"
#include "rcx_launch.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

inline bool known_dtype(int d) { return d == RCX_DTYPE_F32 || d == RCX_DTYPE_BF16 || d == RCX_DTYPE_F16; }
inline int down_size(int h, int k) { const int p = k / 2; return (h + 2 * p - k) / 2 + 1; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ladder {
    int h[RCX_MAX_LEVEL + 1], w[RCX_MAX_LEVEL + 1];
    size_t f_off[RCX_MAX_LEVEL + 1];   // workspace offsets of F_1..F_level (float)
    size_t c_off[2];                   // two ping-pong conv-output buffers (float), each sized for level 1
    size_t total;
};

Ladder make_ladder(int N, int C, int H, int W, int level, int k)
{
    Ladder L{};
    L.h[0] = H; L.w[0] = W;
    size_t off = 0;
    for (int l = 1; l <= level; ++l) {
        L.h[l] = down_size(L.h[l - 1], k);
        L.w[l] = down_size(L.w[l - 1], k);
        L.f_off[l] = off;
        off += align256(sizeof(float) * (size_t)N * C * L.h[l] * L.w[l]);
    }
    const size_t c1 = level >= 1 ? align256(sizeof(float) * (size_t)N * C * L.h[1] * L.w[1]) : 0;
    const size_t c2 = level >= 2 ? align256(sizeof(float) * (size_t)N * C * L.h[2] * L.w[2]) : 0;
    L.c_off[0] = off; off += c1;       // holds C_1, C_3, ...
    L.c_off[1] = off; off += c2;       // holds C_2, C_4, ...
    L.total = off;
    return L;
}

int check_common(const void* x, const void* y, int N, int C, int H, int W, int k, int dtype)
{
    if (!x || !y) return fail(RCX_ERR_BAD_ARG, "null activation pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent N=%d C=%d H=%d W=%d", N, C, H, W);
    if (k <= 0 || (k & 1) == 0) return fail(RCX_ERR_BAD_ARG, "kernel_size must be odd and positive, got %d", k);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    return 0;
}

// Schedule choice: the fused plane kernel whenever it applies; RCX_FORCE_GENERIC=1 pins the
// one-launch-per-ladder-step schedule (used by tests to cover both, and for A/B timing).
bool use_plane(int N, int C, int H, int W, int level, int k, int dtype)
{
    const char* f = rcx::opt::value(rcx::opt::FORCE_GENERIC);
    if (f && *f && *f != '0') return false;
    return rcx::plane_applicable(N, C, H, W, level, k, dtype);
}

// Split schedule for a plane whose level-1 plane does not fit the register file (128x128 / level 4): three launches,
//   F_1 = down(x) (step kernel, float32) ; C_1 = RecConv2d_{level-1}(F_1) with the first level+1 packs (register-resident) ;
//   y = conv_L(x + resize(C_1)) (step kernel).  F_1 and C_1 live in the caller's workspace.
bool use_split(int N, int C, int H, int W, int level, int k, int dtype)
{
    const char* f = rcx::opt::value(rcx::opt::FORCE_GENERIC);
    if (f && *f && *f != '0') return false;
    if (level < 1 || k != 5 || H != W || (H & 1)) return false;
    if (H < 64 && !rcx::opt::value(rcx::opt::FORCE_SPLIT)) return false;   // small planes: the fused LDS-pyramid kernel (or the nested schedule) decides
    const char* force = rcx::opt::value(rcx::opt::FORCE_SPLIT);                  // A/B knob: three launches even where one fused kernel exists
    if (rcx::lanes_applicable(N, C, H, W, level, k, dtype) && !(force && *force == '1')) return false;
    return rcx::down5_lanes_applicable(N, C, H, W, k, 2, dtype, RCX_DTYPE_F32) &&
           rcx::lanes_applicable(N, C, H / 2, W / 2, level - 1, k, RCX_DTYPE_F32) &&
           rcx::upadd_lanes_applicable(N, C, H, W, H / 2, W / 2, k, dtype, RCX_DTYPE_F32, dtype);
}

size_t split_bytes(int N, int C, int H, int W) { return 2 * align256(sizeof(float) * (size_t)N * C * (H / 2) * (W / 2)); }

// the register-resident schedule takes precedence where it applies (RCX_LANES=0 switches it off)
bool use_lanes(int N, int C, int H, int W, int level, int k, int dtype)
{
    const char* f = rcx::opt::value(rcx::opt::FORCE_GENERIC);
    if (f && *f && *f != '0') return false;
    return rcx::lanes_applicable(N, C, H, W, level, k, dtype);
}

bool lanes_off()
{
    const char* f = rcx::opt::value(rcx::opt::FORCE_GENERIC);
    return f && *f && *f != '0';
}

// one ladder rung / one up-recursion step: the register-resident kernel where it applies, else the generic one
// RCX_UPADD_CPT=all: the tiled single-step kernels (rcx_upcpt.hip) also where the lanes kernels keep a ragged channel count (tests)
bool upcpt_everywhere()
{
    const char* v = rcx::opt::value(rcx::opt::UPADD_CPT);
    return v && *v == 'a';
}

// Which single-step kernel a plane gets.  Never a function of N: a batch and its shards must give the same rows bit for bit.
enum StepKernel { STEP_GENERIC, STEP_LANES, STEP_CPT, STEP_CPL14, STEP_CONV5_LANES };

// stride-2 conv5 (stride 1: the plain conv5): the register-resident lanes kernel where it has a plan (the 7 * 2^k / 16 * 2^k squares: it
// already runs at the copy ceiling on 56 x 56 / 28 x 28, 256 x 64 x 56 x 56 24.7 us against 26.1), the tiled channel-per-lane kernel
// (rcx_upcpt.hip) on every other even plane whose width is a multiple of 14 (112 x 112: 15.1 us against 50.8; 200 x 336: 84.8 against 218)
// and on the 16 * 2^k squares in bfloat16 (16-wide tiles: 32 x 64 x 128 x 128 22.4 us against the lanes kernel's 39.9, 32 x 256 x 32 x 32
// 12.1 against 16.2, 64 x 64 16.2 / 16.6).
StepKernel pick_dwconv(int N, int C, int H, int W, int k, int stride, int in_dt, int out_dt)
{
    if (lanes_off()) return STEP_GENERIC;
    if (rcx::down5_cpl7_applicable(N, C, H, W, k, stride, in_dt, out_dt)) return STEP_CPL14;          // the 7 x 7 plane, whole (round 4)
    const bool lanes_ok = rcx::down5_lanes_applicable(N, C, H, W, k, stride, in_dt, out_dt);
    if (lanes_ok && !upcpt_everywhere() && !(W % 14 != 0 && C % 64 == 0)) return STEP_LANES;
    if (rcx::down5_cpt_applicable(N, C, H, W, k, stride, in_dt, out_dt)) return STEP_CPT;
    if (lanes_ok) return STEP_LANES;
    if (stride == 1 && rcx::conv5_lanes_applicable(N, C, H, W, k, in_dt, out_dt)) return STEP_CONV5_LANES;
    return STEP_GENERIC;
}

// conv5(x + resize(coarse)): the whole-plane kernel on 14 x 14; the tiled channel-per-lane kernel (rcx_upcpt.hip) for whole 64-channel
// waves (256 x 64 x 56 x 56: 61 - 63 us against the lanes kernel's 74 - 76) and wherever the lanes kernel has no plan (float16, planes that
// are not 7 * 2^k / 16 * 2^k squares: 32 x 64 x 200 x 336 168 - 176 us against the generic kernel's 860 - 1 413); ragged channel counts on a
// lanes plane stay with the lanes kernel (256 x 96 x 28 x 28: 24 - 26 us against 28 - 52), and so do 16-wide tiles on a plane lower than
// 64 rows, whose third tile row is mostly empty (32 x 256 x 32 x 32: 20 us against 14; 64 x 64: 31 against 35; 128 x 128: 47 - 52 against 85 - 91).
StepKernel pick_upadd(int N, int C, int H, int W, int Hc, int Wc, int k, int x_dt, int c_dt, int out_dt, bool has_coarse)
{
    if (lanes_off()) return STEP_GENERIC;
    if (!has_coarse) return rcx::conv5_lanes_applicable(N, C, H, W, k, x_dt, out_dt) ? STEP_CONV5_LANES : STEP_GENERIC;
    if (rcx::upadd_cpl14_applicable(N, C, H, W, Hc, Wc, k, x_dt, c_dt, out_dt)) return STEP_CPL14;
    const bool lanes_ok = rcx::upadd_lanes_applicable(N, C, H, W, Hc, Wc, k, x_dt, c_dt, out_dt);
    const bool whole = C % 64 == 0 && !(W % 14 != 0 && H < 64);
    if ((whole || !lanes_ok || upcpt_everywhere()) && rcx::upadd_cpt_applicable(N, C, H, W, Hc, Wc, k, x_dt, c_dt, out_dt)) return STEP_CPT;
    return lanes_ok ? STEP_LANES : STEP_GENERIC;
}

// Nested schedule for a plane no fused kernel takes (COCO stages, 112 x 112, ...): the split schedule made recursive,
//   F_1 = conv5 stride 2 (x) (float32, workspace) ; C_1 = RecConv2d_{level-1}(F_1) by WHATEVER schedule that block has -- a fused kernel
//   (lanes / LDS pyramid), or this schedule again -- ; y = conv5(x + resize(C_1)),
// both outer steps on single-step kernels (rcx_upcpt.hip / the lanes step kernels).  200 x 336 / level 4: five launches (two down, the
// LDS-pyramid kernel on 50 x 84 / level 2, two up) instead of the generic ladder's nine.  It only ever replaces the generic ladder.
bool use_nested(int N, int C, int H, int W, int level, int k, int dtype)
{
    if (lanes_off() || level < 1 || k != 5 || (H & 1) || (W & 1)) return false;
    const char* v = rcx::opt::value(rcx::opt::NESTED);
    if (v && *v == '0') return false;
    if (use_lanes(N, C, H, W, level, k, dtype) || use_split(N, C, H, W, level, k, dtype) || use_plane(N, C, H, W, level, k, dtype)) return false;
    const StepKernel d = pick_dwconv(N, C, H, W, k, 2, dtype, RCX_DTYPE_F32);
    const StepKernel u = pick_upadd(N, C, H, W, H / 2, W / 2, k, dtype, RCX_DTYPE_F32, dtype, true);
    if (d == STEP_GENERIC || u == STEP_GENERIC) return false;
    const int h2 = H / 2, w2 = W / 2;
    return use_lanes(N, C, h2, w2, level - 1, k, RCX_DTYPE_F32) || use_split(N, C, h2, w2, level - 1, k, RCX_DTYPE_F32) ||
           use_plane(N, C, h2, w2, level - 1, k, RCX_DTYPE_F32) || use_nested(N, C, h2, w2, level - 1, k, RCX_DTYPE_F32);
}

size_t nested_own_bytes(int N, int C, int H, int W) { return 2 * align256(sizeof(float) * (size_t)N * C * (H / 2) * (W / 2)); }

hipError_t step_dwconv(const void* x, void* y, const float* w, const float* b, int N, int C, int H, int W, int k, int stride,
                       int in_dt, int out_dt, hipStream_t s)
{
    switch (pick_dwconv(N, C, H, W, k, stride, in_dt, out_dt)) {
    case STEP_CPL14: return rcx::down5_cpl7(x, y, w, b, N, C, H, in_dt, s);
    case STEP_LANES: return rcx::down5_lanes(x, y, w, b, N, C, H, W, in_dt, out_dt, s);
    case STEP_CPT: return rcx::down5_cpt(x, y, w, b, N, C, H, W, in_dt, out_dt, s);
    case STEP_CONV5_LANES: return rcx::conv5_lanes(x, y, w, b, N, C, H, W, in_dt, s);
    default: return rcx::generic_dwconv(x, y, w, b, N, C, H, W, k, stride, in_dt, out_dt, s);
    }
}

hipError_t step_upadd(const void* x, const void* coarse, void* y, const float* w, const float* b, int N, int C, int H, int W,
                      int Hc, int Wc, int k, int mode, int x_dt, int c_dt, int out_dt, hipStream_t s)
{
    switch (pick_upadd(N, C, H, W, Hc, Wc, k, x_dt, c_dt, out_dt, coarse != nullptr)) {
    case STEP_CPL14: return rcx::upadd_cpl14(x, coarse, y, w, b, N, C, H, mode, x_dt, c_dt, s);
    case STEP_CPT: return rcx::upadd_cpt(x, coarse, y, w, b, N, C, H, W, mode, x_dt, c_dt, s);
    case STEP_LANES: return rcx::upadd_lanes(x, coarse, y, w, b, N, C, H, W, mode, x_dt, c_dt, s);
    case STEP_CONV5_LANES: return rcx::conv5_lanes(x, y, w, b, N, C, H, W, x_dt, s);
    default: return rcx::generic_upadd_dwconv(x, coarse, y, w, b, N, C, H, W, Hc, Wc, k, mode, x_dt, c_dt, out_dt, s);
    }
}

}  // namespace

#ifdef RCX_STAMPS
namespace rcx { hipError_t set_stamp_buffer(void* p); namespace lanes { hipError_t set_stamp_buffer(void* p); } }
#endif

extern "C" {

int rcx_abi_version(void) { return RCX_ABI_VERSION; }

void rcx_reload_options(void) { rcx::opt::reload(); }

int rcx_selftest_d16(const void* src, void* flag, void* stream)
{
    if (!src || !flag) return fail(RCX_ERR_BAD_ARG, "rcx_selftest_d16: null pointer");
    hipError_t e = rcx::selftest_d16(src, flag, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_selftest_d16");
}

const char* rcx_last_error(void) { return g_err; }

const char* rcx_recconv2d_fwd_plan(int N, int C, int H, int W, int level, int k, int mode, int dtype)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || level < 0 || level > RCX_MAX_LEVEL || k <= 0 || (k & 1) == 0) return "invalid";
    static thread_local char desc[256];
    if (!(rcx::opt::value(rcx::opt::FORCE_SPLIT) && use_split(N, C, H, W, level, k, dtype)) && use_lanes(N, C, H, W, level, k, dtype) && rcx::lanes_describe(N, C, H, W, level, k, mode == RCX_MODE_NEAREST ? 1 : 0, dtype, desc, (int)sizeof(desc)) > 0) return desc;
    if (use_split(N, C, H, W, level, k, dtype)) {
        char inner[128];
        rcx::lanes_describe(N, C, H / 2, W / 2, level - 1, k, mode == RCX_MODE_NEAREST ? 1 : 0, RCX_DTYPE_F32, inner, (int)sizeof(inner));
        const bool dcpt = pick_dwconv(N, C, H, W, k, 2, dtype, RCX_DTYPE_F32) == STEP_CPT;
        const bool ucpt = pick_upadd(N, C, H, W, H / 2, W / 2, k, dtype, RCX_DTYPE_F32, dtype, true) == STEP_CPT;
        snprintf(desc, sizeof(desc), "split(%s + %s + %s)", dcpt ? "k_down5_cpt" : "k_down5_lanes", inner, ucpt ? "k_upadd_cpt" : "k_upadd_lanes");
        return desc;
    }
    if (use_nested(N, C, H, W, level, k, dtype)) {
        char inner[192];
        snprintf(inner, sizeof(inner), "%s", rcx_recconv2d_fwd_plan(N, C, H / 2, W / 2, level - 1, k, mode, RCX_DTYPE_F32));   // (overwrites desc)
        const bool dcpt = pick_dwconv(N, C, H, W, k, 2, dtype, RCX_DTYPE_F32) == STEP_CPT;
        const bool ucpt = pick_upadd(N, C, H, W, H / 2, W / 2, k, dtype, RCX_DTYPE_F32, dtype, true) == STEP_CPT;
        snprintf(desc, sizeof(desc), "nested(%s + %s + %s)", dcpt ? "k_down5_cpt" : "k_down5_lanes", inner, ucpt ? "k_upadd_cpt" : "k_upadd_lanes");
        return desc;
    }
    if (!use_plane(N, C, H, W, level, k, dtype)) return "generic";
    if (rcx::plane_describe(N, C, H, W, level, k, dtype, desc, (int)sizeof(desc)) <= 0) return "generic";
    return desc;
}

int rcx_pack_dw_weight(const void* w_ckk, float* dst_kkc, int C, int k, int dtype, void* stream)
{
    if (!w_ckk || !dst_kkc || C <= 0 || k <= 0) return fail(RCX_ERR_BAD_ARG, "rcx_pack_dw_weight: bad argument");
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    hipError_t e = rcx::pack_dw_weight(w_ckk, dst_kkc, C, k, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_pack_dw_weight");
}

int rcx_pack_recconv_params(const void* const* w, const void* const* b, float* wpack, float* wpack_flipped, float* bpack,
                            int count, int C, int k, int dtype, void* stream)
{
    if (!w || !wpack || count <= 0 || count > RCX_MAX_LEVEL + 2 || C <= 0 || k <= 0) return fail(RCX_ERR_BAD_ARG, "rcx_pack_recconv_params: bad argument");
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (bpack && !b) return fail(RCX_ERR_BAD_ARG, "rcx_pack_recconv_params: bias destination without bias sources");
    rcx::PackPtrs P{};
    for (int j = 0; j < count; ++j) {
        if (!w[j]) return fail(RCX_ERR_BAD_ARG, "rcx_pack_recconv_params: null weight %d", j);
        P.w[j] = w[j];
        P.b[j] = b ? b[j] : nullptr;
    }
    hipError_t e = rcx::pack_params(P, wpack, wpack_flipped, bpack, count, C, k, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_pack_recconv_params");
}

int rcx_unpack_recconv_grads(const float* gwpack, void* const* gw, int count, int C, int k, void* stream)
{
    if (!gwpack || !gw || count <= 0 || count > RCX_MAX_LEVEL + 2 || C <= 0 || k <= 0) return fail(RCX_ERR_BAD_ARG, "rcx_unpack_recconv_grads: bad argument");
    rcx::PackPtrs P{};
    for (int j = 0; j < count; ++j) {
        if (!gw[j]) return fail(RCX_ERR_BAD_ARG, "rcx_unpack_recconv_grads: null destination %d", j);
        P.w[j] = gw[j];
    }
    hipError_t e = rcx::unpack_grads(gwpack, P, count, C, k, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_unpack_recconv_grads");
}

int rcx_pack_bias(const void* b, float* dst, int C, int dtype, void* stream)
{
    if (!b || !dst || C <= 0) return fail(RCX_ERR_BAD_ARG, "rcx_pack_bias: bad argument");
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    hipError_t e = rcx::pack_bias(b, dst, C, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_pack_bias");
}

size_t rcx_recconv2d_fwd_workspace_bytes(int N, int C, int H, int W, int level, int k, int dtype)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || level < 0 || level > RCX_MAX_LEVEL || k <= 0 || (k & 1) == 0) return 0;
    if (rcx::opt::value(rcx::opt::FORCE_SPLIT) && use_split(N, C, H, W, level, k, dtype)) return split_bytes(N, C, H, W);
    if (use_lanes(N, C, H, W, level, k, dtype)) return 0;          // registers only
    if (use_split(N, C, H, W, level, k, dtype)) return split_bytes(N, C, H, W);
    if (use_plane(N, C, H, W, level, k, dtype)) return 0;          // the fused schedule keeps every intermediate in LDS
    if (use_nested(N, C, H, W, level, k, dtype))                    // F_1, C_1, then whatever the inner block needs
        return nested_own_bytes(N, C, H, W) + rcx_recconv2d_fwd_workspace_bytes(N, C, H / 2, W / 2, level - 1, k, RCX_DTYPE_F32);
    return make_ladder(N, C, H, W, level, k).total;
}

// ---- rcx_time_next_launch: see rcx_launch.h
extern "C++" {
namespace rcx {
static thread_local LaunchEvents g_launch_events;
LaunchEvents take_launch_events()
{
    const LaunchEvents e = g_launch_events;
    g_launch_events = LaunchEvents{};
    return e;
}
}  // namespace rcx
}

int rcx_time_next_launch(void* start_event, void* stop_event)
{
    rcx::g_launch_events.start = (hipEvent_t)start_event;
    rcx::g_launch_events.stop = (hipEvent_t)stop_event;
    return 0;
}

int rcx_launch_events_pending(void) { return (rcx::g_launch_events.start || rcx::g_launch_events.stop) ? 1 : 0; }

static int recconv2d_fwd_impl(const void* x, void* y, const float* wpack, const float* bpack, void* workspace, size_t workspace_bytes,
                              int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream);

int rcx_recconv2d_fwd(const void* x, void* y, const float* wpack, const float* bpack,
                      void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream)
{
    const int rc = recconv2d_fwd_impl(x, y, wpack, bpack, workspace, workspace_bytes, N, C, H, W, level, k, mode, dtype, stream);
    rcx::g_launch_events = rcx::LaunchEvents{};        // a pair the schedule did not consume (several launches, an unsupported kernel) does not wait for a later call
    return rc;
}

static int recconv2d_fwd_impl(const void* x, void* y, const float* wpack, const float* bpack,
                      void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream)
{
    if (int rc = check_common(x, y, N, C, H, W, k, dtype)) return rc;
    if (!wpack) return fail(RCX_ERR_BAD_ARG, "null weight pack");
    if (x == y) return fail(RCX_ERR_BAD_ARG, "y must not alias x");
    if (level < 0 || level > RCX_MAX_LEVEL) return fail(RCX_ERR_BAD_ARG, "level %d outside [0,%d]", level, RCX_MAX_LEVEL);
    if (mode != RCX_MODE_BILINEAR && mode != RCX_MODE_NEAREST) return fail(RCX_ERR_BAD_ARG, "unknown mode %d", mode);
    if (!(rcx::opt::value(rcx::opt::FORCE_SPLIT) && use_split(N, C, H, W, level, k, dtype)) && use_lanes(N, C, H, W, level, k, dtype)) {
        hipError_t le = rcx::lanes_recconv(x, y, wpack, bpack, N, C, H, W, level, k, mode, dtype, (hipStream_t)stream);
        return le == hipSuccess ? 0 : hip_fail(le, "lanes schedule");
    }
    if (use_split(N, C, H, W, level, k, dtype)) {
        const size_t need = split_bytes(N, C, H, W);
        if (!workspace || workspace_bytes < need)
            return fail(RCX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        hipStream_t s = (hipStream_t)stream;
        float* f1 = (float*)workspace;
        float* c1 = (float*)((char*)workspace + need / 2);
        const size_t wsz = (size_t)k * k * C;
        hipError_t e = step_dwconv(x, f1, wpack, bpack, N, C, H, W, k, 2, dtype, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "split schedule: down");
        e = rcx::lanes_recconv(f1, c1, wpack, bpack, N, C, H / 2, W / 2, level - 1, k, mode, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "split schedule: inner block");
        e = step_upadd(x, c1, y, wpack + (size_t)(1 + level) * wsz, bpack ? bpack + (size_t)(1 + level) * C : nullptr,
                       N, C, H, W, H / 2, W / 2, k, mode, dtype, RCX_DTYPE_F32, dtype, s);
        return e == hipSuccess ? 0 : hip_fail(e, "split schedule: final conv");
    }
    if (use_plane(N, C, H, W, level, k, dtype)) {
        hipError_t pe = rcx::plane_recconv(x, y, wpack, bpack, N, C, H, W, level, k, mode, dtype, (hipStream_t)stream);
        return pe == hipSuccess ? 0 : hip_fail(pe, "plane schedule");
    }
    if (use_nested(N, C, H, W, level, k, dtype)) {
        const size_t own = nested_own_bytes(N, C, H, W);
        const size_t need = own + rcx_recconv2d_fwd_workspace_bytes(N, C, H / 2, W / 2, level - 1, k, RCX_DTYPE_F32);
        if (!workspace || workspace_bytes < need)
            return fail(RCX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
        hipStream_t s = (hipStream_t)stream;
        float* f1 = (float*)workspace;
        float* c1 = (float*)((char*)workspace + own / 2);
        const size_t wsz = (size_t)k * k * C;
        hipError_t e = step_dwconv(x, f1, wpack, bpack, N, C, H, W, k, 2, dtype, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "nested schedule: down");
        // the inner block owns the first level + 1 packs (down, convs[0 .. level - 1]) exactly as the whole block owns level + 2
        if (int rc = rcx_recconv2d_fwd(f1, c1, wpack, bpack, (char*)workspace + own, need - own, N, C, H / 2, W / 2, level - 1, k, mode, RCX_DTYPE_F32, stream))
            return rc;
        e = step_upadd(x, c1, y, wpack + (size_t)(1 + level) * wsz, bpack ? bpack + (size_t)(1 + level) * C : nullptr,
                       N, C, H, W, H / 2, W / 2, k, mode, dtype, RCX_DTYPE_F32, dtype, s);
        return e == hipSuccess ? 0 : hip_fail(e, "nested schedule: final conv");
    }
    const Ladder L = make_ladder(N, C, H, W, level, k);
    if (L.total > 0 && (!workspace || workspace_bytes < L.total))
        return fail(RCX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const size_t wsz = (size_t)k * k * C;
    auto W_ = [&](int i) { return wpack + (size_t)i * wsz; };                       // 0 = down, 1+j = convs[j]
    auto B_ = [&](int i) { return bpack ? bpack + (size_t)i * C : nullptr; };
    char* ws = (char*)workspace;
    auto F_ = [&](int l) { return (float*)(ws + L.f_off[l]); };
    auto Cb = [&](int l) { return (float*)(ws + L.c_off[(l + 1) & 1]); };           // C_1 -> buf 0, C_2 -> buf 1, ...
    hipError_t e;
    // down ladder, shared weight (model/recnext.py:27-29); F_l kept in float32
    for (int l = 1; l <= level; ++l) {
        const void* src = l == 1 ? x : (const void*)F_(l - 1);
        e = step_dwconv(src, F_(l), W_(0), B_(0), N, C, L.h[l - 1], L.w[l - 1], k, 2,          // the best single-step kernel each plane has
                        l == 1 ? dtype : RCX_DTYPE_F32, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "down ladder");
    }
    // up recursion, coarsest first (model/recnext.py:31-33): C_l = conv_j(F_l + resize(C_{l+1}))
    for (int l = level, j = 0; l >= 1; --l, ++j) {
        const float* coarse = l == level ? nullptr : Cb(l + 1);
        e = step_upadd(F_(l), coarse, Cb(l), W_(1 + j), B_(1 + j), N, C, L.h[l], L.w[l],
                       l == level ? 0 : L.h[l + 1], l == level ? 0 : L.w[l + 1], k, mode,
                       RCX_DTYPE_F32, RCX_DTYPE_F32, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "up recursion");
    }
    // final conv (model/recnext.py:34)
    e = step_upadd(x, level >= 1 ? Cb(1) : nullptr, y, W_(1 + level), B_(1 + level), N, C, H, W,
                   level >= 1 ? L.h[1] : 0, level >= 1 ? L.w[1] : 0, k, mode,
                   dtype, RCX_DTYPE_F32, dtype, s);
    if (e != hipSuccess) return hip_fail(e, "final conv");
    return 0;
}

// ---- matrix-core schedules: 16-bit activations whose taps may be rounded to the same type ----
namespace {
}  // namespace

// ---- training: forward that keeps the fp32 pyramid, and the backward pass (rcx_bwd.hip) ----
namespace {

struct TrainLadder {
    int h[RCX_MAX_LEVEL + 1], w[RCX_MAX_LEVEL + 1];
    size_t f_off[RCX_MAX_LEVEL + 1], c_off[RCX_MAX_LEVEL + 1];   // F_l, C_l (l >= 1), all distinct
    size_t saved_total;
    size_t g_off[RCX_MAX_LEVEL + 1];                               // backward scratch: gT_0..gT_L
    size_t gc_off, part_off, part_bytes, bwd_total;
};

TrainLadder make_train_ladder(int N, int C, int H, int W, int level, int k)
{
    TrainLadder L{};
    L.h[0] = H; L.w[0] = W;
    size_t off = 0;
    for (int l = 1; l <= level; ++l) {
        L.h[l] = down_size(L.h[l - 1], k); L.w[l] = down_size(L.w[l - 1], k);
        const size_t b = align256(sizeof(float) * (size_t)N * C * L.h[l] * L.w[l]);
        L.f_off[l] = off; off += b;
        L.c_off[l] = off; off += b;
    }
    L.saved_total = off;
    off = 0;
    for (int l = 0; l <= level; ++l) { L.g_off[l] = off; off += align256(sizeof(float) * (size_t)N * C * L.h[l] * L.w[l]); }
    L.gc_off = off; off += level >= 1 ? align256(sizeof(float) * (size_t)N * C * L.h[1] * L.w[1]) : 0;
    L.part_bytes = align256(rcx::wgrad_partial_bytes(C, k));
    // the tiled weight-gradient kernels (rcx_cptbwd_kernels.h) leave one row of (k*k + 1) * C sums per (image, 14-row band): N * H / 14 rows
    if (rcx::bwd_cpt_applicable(N, C, H, W, k)) {
        const size_t tiled = align256(sizeof(float) * (size_t)N * ((H + 13) / 14) * (size_t)(k * k + 1) * C);
        if (tiled > L.part_bytes) L.part_bytes = tiled;
    }
    L.part_off = off; off += L.part_bytes * (size_t)(2 * level + 1);     // one partial buffer per weight-gradient call: reduced together
    L.bwd_total = off;
    return L;
}

}  // namespace

size_t rcx_recconv2d_train_saved_bytes(int N, int C, int H, int W, int level, int k)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || level < 0 || level > RCX_MAX_LEVEL || k <= 0 || (k & 1) == 0) return 0;
    return make_train_ladder(N, C, H, W, level, k).saved_total;
}

size_t rcx_recconv2d_bwd_workspace_bytes(int N, int C, int H, int W, int level, int k)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || level < 0 || level > RCX_MAX_LEVEL || k <= 0 || (k & 1) == 0) return 0;
    return make_train_ladder(N, C, H, W, level, k).bwd_total;
}

int rcx_recconv2d_fwd_train(const void* x, void* y, const float* wpack, const float* bpack, void* saved, size_t saved_bytes,
                            int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream)
{
    if (int rc = check_common(x, y, N, C, H, W, k, dtype)) return rc;
    if (!wpack) return fail(RCX_ERR_BAD_ARG, "null weight pack");
    if (level < 0 || level > RCX_MAX_LEVEL) return fail(RCX_ERR_BAD_ARG, "level %d outside [0,%d]", level, RCX_MAX_LEVEL);
    if (mode != RCX_MODE_BILINEAR && mode != RCX_MODE_NEAREST) return fail(RCX_ERR_BAD_ARG, "unknown mode %d", mode);
    const TrainLadder L = make_train_ladder(N, C, H, W, level, k);
    if (L.saved_total > 0 && (!saved || saved_bytes < L.saved_total))
        return fail(RCX_ERR_WORKSPACE, "saved-activation buffer too small: need %zu bytes, got %zu", L.saved_total, saved_bytes);
    hipStream_t s = (hipStream_t)stream;
    // the blocks of RecNeXt at 224x224 (channel-per-lane kernels): the inference kernel itself leaves the pyramid behind --
    // one launch instead of 2 * level + 1 (RCX_TRAIN_FUSED=0: the per-step schedule, for A/B runs)
    {
        const char* tf = rcx::opt::value(rcx::opt::TRAIN_FUSED);
        const bool fused_ok = !(tf && *tf == '0') && !lanes_off() && !(rcx::opt::value(rcx::opt::FORCE_SPLIT));
        const int md = mode == RCX_MODE_NEAREST ? 1 : 0;
        if (fused_ok && rcx::cpt_train_applicable(N, C, H, W, level, k, md, dtype)) {
            hipError_t fe = rcx::cpt_recconv(x, y, wpack, bpack, N, C, H, level, md, dtype, s, (float*)saved, L.f_off, L.c_off);
            return fe == hipSuccess ? 0 : hip_fail(fe, "train fwd: fused tiled block");
        }
        if (fused_ok && rcx::cpl14_applicable(N, C, H, W, level, k, dtype)) {
            hipError_t fe = rcx::cpl14_recconv(x, y, wpack, bpack, N, C, md, dtype, s, (float*)saved, L.f_off, L.c_off);
            return fe == hipSuccess ? 0 : hip_fail(fe, "train fwd: fused 14x14 block");
        }
        if (fused_ok && rcx::cpl7b_applicable(N, C, H, W, level, k, dtype)) {
            hipError_t fe = rcx::cpl7b_recconv(x, y, wpack, bpack, N, C, md, dtype, s, (float*)saved, L.f_off, L.c_off);
            return fe == hipSuccess ? 0 : hip_fail(fe, "train fwd: fused 7x7 block");
        }
    }
    const size_t wsz = (size_t)k * k * C;
    auto W_ = [&](int i) { return wpack + (size_t)i * wsz; };
    auto B_ = [&](int i) { return bpack ? bpack + (size_t)i * C : nullptr; };
    char* ws = (char*)saved;
    auto F_ = [&](int l) { return (float*)(ws + L.f_off[l]); };
    auto C_ = [&](int l) { return (float*)(ws + L.c_off[l]); };
    hipError_t e;
    for (int l = 1; l <= level; ++l) {
        e = step_dwconv(l == 1 ? x : (const void*)F_(l - 1), F_(l), W_(0), B_(0), N, C, L.h[l - 1], L.w[l - 1], k, 2,
                        l == 1 ? dtype : RCX_DTYPE_F32, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "train fwd: down ladder");
    }
    for (int l = level, j = 0; l >= 1; --l, ++j) {
        e = step_upadd(F_(l), l == level ? nullptr : C_(l + 1), C_(l), W_(1 + j), B_(1 + j), N, C, L.h[l], L.w[l],
                       l == level ? 0 : L.h[l + 1], l == level ? 0 : L.w[l + 1], k, mode,
                       RCX_DTYPE_F32, RCX_DTYPE_F32, RCX_DTYPE_F32, s);
        if (e != hipSuccess) return hip_fail(e, "train fwd: up recursion");
    }
    e = step_upadd(x, level >= 1 ? C_(1) : nullptr, y, W_(1 + level), B_(1 + level), N, C, H, W,
                   level >= 1 ? L.h[1] : 0, level >= 1 ? L.w[1] : 0, k, mode, dtype, RCX_DTYPE_F32, dtype, s);
    return e == hipSuccess ? 0 : hip_fail(e, "train fwd: final conv");
}

// the fine levels above a 14 x 14 / level 2 tail run on the tiled adjoint kernels (rcx_cptbwd.hip) when every one of their planes is 56 x 56 or 28 x 28:
// the number of such levels (2: the 56 x 56 / level 4 block, 1: 28 x 28 / level 3), 0 = the per-step schedule
static int bwd_cpt_levels(const TrainLadder& L, int N, int C, int level, int k, int dtype)
{
    if (level < 3 || lanes_off() || rcx::opt::is_zero(rcx::opt::BWD_FUSED) || rcx::opt::is_zero(rcx::opt::BWD_NESTED)) return 0;
    const int m = level - 2;
    if (L.h[m] != 14 || L.w[m] != 14 || !rcx::cplbwd_applicable(N, C, 14, 14, 2, k, RCX_DTYPE_F32)) return 0;
    for (int l = 0; l < m; ++l)
        if (!rcx::bwd_cpt_applicable(N, C, L.h[l], L.w[l], k) || L.h[l + 1] * 2 != L.h[l] || L.w[l + 1] * 2 != L.w[l]) return 0;
    (void)dtype;
    return m;
}

int rcx_recconv2d_bwd_gy_dtype(int N, int C, int H, int W, int level, int k, int dtype)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || level < 0 || level > RCX_MAX_LEVEL || k <= 0 || (k & 1) == 0 || !known_dtype(dtype)) return RCX_DTYPE_F32;
    // bfloat16 only: with float16 rows on both sides the tiled weight-gradient kernel does not fit its register budget (rcx_cptbwd_kernels.h)
    if (dtype != RCX_DTYPE_BF16 || C % 4) return RCX_DTYPE_F32;
    if (level >= 1 && !lanes_off() && rcx::cplbwd_applicable(N, C, H, W, level, k, dtype)) return dtype;      // the one-launch 14 x 14 / 7 x 7 backward
    return bwd_cpt_levels(make_train_ladder(N, C, H, W, level, k), N, C, level, k, dtype) > 0 ? dtype : RCX_DTYPE_F32;
}

int rcx_recconv2d_bwd(const void* x, const void* gy, int gy_dtype, const float* wpack, const float* wpack_flipped, const void* saved,
                      void* gx, float* gwpack, float* gbpack, void* const* gw_out, void* const* gb_out, int grad_dtype,
                      void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream)
{
    if (int rc = check_common(x, gx, N, C, H, W, k, dtype)) return rc;
    if (!gy || !wpack || !wpack_flipped || (!gwpack && !gw_out)) return fail(RCX_ERR_BAD_ARG, "null gradient / weight pointer");
    if (!known_dtype(gy_dtype) || (gw_out && !known_dtype(grad_dtype))) return fail(RCX_ERR_BAD_ARG, "unknown gy / gradient dtype");
    if (level < 0 || level > RCX_MAX_LEVEL) return fail(RCX_ERR_BAD_ARG, "level %d outside [0,%d]", level, RCX_MAX_LEVEL);
    if (mode != RCX_MODE_BILINEAR && mode != RCX_MODE_NEAREST) return fail(RCX_ERR_BAD_ARG, "unknown mode %d", mode);
    if (C % 4) return fail(RCX_ERR_UNSUPPORTED, "the backward kernels need C %% 4 == 0, got C=%d", C);
    const TrainLadder L = make_train_ladder(N, C, H, W, level, k);
    if (level >= 1 && !saved) return fail(RCX_ERR_BAD_ARG, "null saved-activation buffer");
    if (!workspace || workspace_bytes < L.bwd_total)
        return fail(RCX_ERR_WORKSPACE, "backward workspace too small: need %zu bytes, got %zu", L.bwd_total, workspace_bytes);
    const int mcpt = bwd_cpt_levels(L, N, C, level, k, dtype);
    const bool whole = level >= 1 && !lanes_off() && rcx::cplbwd_applicable(N, C, H, W, level, k, dtype);
    if (gy_dtype != RCX_DTYPE_F32 && !((mcpt > 0 || whole) && gy_dtype == dtype && dtype == RCX_DTYPE_BF16))
        return fail(RCX_ERR_UNSUPPORTED, "gy of dtype %d: this problem takes float32 (rcx_recconv2d_bwd_gy_dtype)", gy_dtype);
    if (gw_out)
        for (int i = 0; i < level + 2; ++i)
            if (!gw_out[i]) return fail(RCX_ERR_BAD_ARG, "null gw_out[%d]", i);
    hipStream_t s = (hipStream_t)stream;
    const size_t wsz = (size_t)k * k * C;
    auto W_ = [&](int i) { return wpack + (size_t)i * wsz; };
    auto Wf = [&](int i) { return wpack_flipped + (size_t)i * wsz; };
    // where the final reduction leaves conv i's gradients: the packed float32 rows, or the parameters' own tensors
    auto GW = [&](int i) { return gw_out ? (float*)gw_out[i] : gwpack + (size_t)i * wsz; };
    auto GB = [&](int i) { return gw_out ? (gb_out ? (float*)gb_out[i] : nullptr) : (gbpack ? gbpack + (size_t)i * C : nullptr); };
    const char* sv = (const char*)saved;
    auto F_ = [&](int l) { return (const float*)(sv + L.f_off[l]); };
    auto C_ = [&](int l) { return (const float*)(sv + L.c_off[l]); };
    char* ws = (char*)workspace;
    auto G_ = [&](int l) { return (float*)(ws + L.g_off[l]); };
    float* gC = (float*)(ws + L.gc_off);
    int slot = 0;
    auto PART = [&](int i) { return (float*)(ws + L.part_off + (size_t)i * L.part_bytes); };
    rcx::WgradJobs J{};
    J.kk = k * k; J.C = C;
    J.param_layout = gw_out ? 1 : 0;
    J.param_dt = grad_dtype;
    // job 0 = the shared down conv (its partial buffers are appended as the ladder is walked), job 1 + j = convs[j]
    J.njobs = level + 2;
    J.gw[0] = GW(0); J.gb[0] = GB(0); J.nslots[0] = 0;
    for (int i = 0; i <= level; ++i) { J.gw[1 + i] = GW(1 + i); J.gb[1 + i] = GB(1 + i); J.nslots[1 + i] = 0; }
    auto add_slot = [&](int job, const float* p, int rows) { J.part[job][J.nslots[job]] = p; J.rows[job][J.nslots[job]] = rows; ++J.nslots[job]; };
    int rows = 0;
    hipError_t e;
#define RCX_TRY(call, what) do { e = (call); if (e != hipSuccess) return hip_fail(e, what); } while (0)
    // the blocks whose planes fit one lane: the whole backward in one launch + the batch reduction (RCX_BWD_FUSED=0: per-step schedule)
    if (whole) {
        float* parts[RCX_MAX_LEVEL + 2];
        for (int j = 0; j < level + 2; ++j) { parts[j] = PART(j); add_slot(j, PART(j), N); }
        RCX_TRY(rcx::cplbwd_recconv(x, gy, wpack, wpack_flipped, saved, L.f_off, L.c_off, gx, parts, N, C, H, level,
                                    mode == RCX_MODE_NEAREST ? 1 : 0, dtype, s, gy_dtype), "bwd: fused block");
        RCX_TRY(rcx::bwd_wgrad_reduce_jobs(J, s), "bwd: weight-gradient reduction");
        return 0;
    }
    // The 56x56 / level 4 and 28x28 / level 3 blocks (RecNeXt at 224x224): the fine levels on the tiled adjoint kernels, the 14x14 / level 2 tail as
    // its one launch.  Top-down: gW_j from (a_l, C_{l+1}, g_l) and gC_{l+1} = R^T K^ g_l; the tail returns G_m; bottom-up: gW_d from (a_l, G_{l+1})
    // and G_l = K^ g_l + D^T G_{l+1} (G_0 = gx).  g_0 = gy in its own type, g_l = gC_l float32 (parked in the full-resolution slot G_(0) the per-step
    // schedule keeps gT_0 in: no gT plane exists here).
    if (mcpt > 0) {
        const int m = mcpt, md = mode == RCX_MODE_NEAREST ? 1 : 0;
        float* gcl[RCX_MAX_LEVEL + 1] = {};
        {
            size_t off = 0;
            for (int l = 1; l <= m; ++l) { gcl[l] = (float*)(ws + L.g_off[0] + off); off += align256(sizeof(float) * (size_t)N * C * L.h[l] * L.w[l]); }
        }
        auto g_of = [&](int l) { return l == 0 ? gy : (const void*)gcl[l]; };
        auto gdt_of = [&](int l) { return l == 0 ? gy_dtype : RCX_DTYPE_F32; };
        auto a_of = [&](int l) { return l == 0 ? x : (const void*)F_(l); };
        auto adt_of = [&](int l) { return l == 0 ? dtype : RCX_DTYPE_F32; };
        // (Round 6 measured the four weight-gradient kernels on a second stream, forked and joined by events inside the call -- nothing but the final
        // reduction waits for them, and the chain's middle runs on a fraction of the chip: the event hand-offs cost more than the overlap returns,
        // 342 vs 301 us at 128 x 64 x 56 x 56, 301 vs 220 us at 256 x 128 x 28 x 28, equal at 256 x 64 x 56 x 56; profiles/r06_backward_side_stream.txt.)
        for (int l = 0; l < m; ++l) {
            const int j = level - l;                              // convs[j] is level l's conv
            RCX_TRY(rcx::bwd_wgrad_k_cpt(a_of(l), adt_of(l), C_(l + 1), g_of(l), gdt_of(l), PART(slot), N, C, L.h[l], md, s, &rows), "bwd: conv weight grad");
            add_slot(1 + j, PART(slot++), rows);
            RCX_TRY(rcx::bwd_gc_cpt(g_of(l), gdt_of(l), gcl[l + 1], Wf(1 + j), N, C, L.h[l], md, s), "bwd: gradient handed down");
        }
        {
            float* parts[4];
            for (int q = 0; q < 4; ++q) parts[q] = PART(slot++);
            RCX_TRY(rcx::cplbwd_recconv(F_(m), gcl[m], wpack, wpack_flipped, saved, L.f_off + m, L.c_off + m, G_(m), parts, N, C, 14, 2, md,
                                        RCX_DTYPE_F32, s), "bwd: fused nested block");
            for (int q = 0; q < 4; ++q) add_slot(q, parts[q], N);
        }
        for (int l = m - 1; l >= 0; --l) {
            RCX_TRY(rcx::bwd_wgrad_d_cpt(a_of(l), adt_of(l), G_(l + 1), PART(slot), N, C, L.h[l], s, &rows), "bwd: down weight grad");
            add_slot(0, PART(slot++), rows);
            RCX_TRY(rcx::bwd_gx_cpt(g_of(l), gdt_of(l), G_(l + 1), l == 0 ? gx : (void*)G_(l), l == 0 ? dtype : RCX_DTYPE_F32, Wf(1 + level - l), W_(0),
                                    N, C, L.h[l], s), "bwd: gradient handed up");
        }
        RCX_TRY(rcx::bwd_wgrad_reduce_jobs(J, s), "bwd: weight-gradient reduction");
        return 0;
    }
    const float* gyf = (const float*)gy;                          // the per-step schedule reads float32
    // final conv (model/recnext.py:34): gT_0 = K_L^T gy ; gW_L = <x + R(C_1), gy>
    if (level == 0) RCX_TRY(step_dwconv(gyf, gx, Wf(1), nullptr, N, C, H, W, k, 1, RCX_DTYPE_F32, dtype, s), "bwd: final conv input grad");
    else RCX_TRY(step_dwconv(gyf, G_(0), Wf(1 + level), nullptr, N, C, H, W, k, 1, RCX_DTYPE_F32, RCX_DTYPE_F32, s), "bwd: final conv input grad");
    RCX_TRY(rcx::bwd_wgrad(x, dtype, level >= 1 ? C_(1) : nullptr, gyf, PART(slot), GW(1 + level), GB(1 + level), N, C, H, W,
                           level >= 1 ? L.h[1] : 0, level >= 1 ? L.w[1] : 0, H, W, k, 1, mode, 0, s, &rows), "bwd: final conv weight grad");
    add_slot(1 + level, PART(slot++), rows);
    // A deeper block whose level (level - 2) plane is 14x14 -- the 28x28 / level 3 and 56x56 / level 4 blocks of RecNeXt at 224x224 --
    // ends in exactly the 14x14 / level 2 block (input F_m, output C_m, convs[0..2], the shared down conv): its whole backward is
    // the one fused launch, the levels above it keep the per-step kernels.
    int m = 0;
    if (level >= 3 && !lanes_off() && L.h[level - 2] == 14 && L.w[level - 2] == 14 &&
        rcx::cplbwd_applicable(N, C, 14, 14, 2, k, RCX_DTYPE_F32)) {
        const char* nv = rcx::opt::value(rcx::opt::BWD_NESTED);
        if (!(nv && *nv == '0')) m = level - 2;
    }
    // up recursion (:31-33), finest level first in the backward direction
    for (int l = 1; l <= level; ++l) {
        const int j = level - l;
        RCX_TRY(rcx::bwd_resize(G_(l - 1), gC, N, C, L.h[l - 1], L.w[l - 1], L.h[l], L.w[l], mode, s), "bwd: resize adjoint");
        if (m && l == m) {
            float* parts[4];
            for (int q = 0; q < 4; ++q) parts[q] = PART(slot++);
            RCX_TRY(rcx::cplbwd_recconv(F_(m), gC, wpack, wpack_flipped, saved, L.f_off + m, L.c_off + m, G_(m), parts, N, C, 14, 2,
                                        mode == RCX_MODE_NEAREST ? 1 : 0, RCX_DTYPE_F32, s), "bwd: fused nested block");
            for (int q = 0; q < 4; ++q) add_slot(q, parts[q], N);
            break;
        }
        RCX_TRY(step_dwconv(gC, G_(l), Wf(1 + j), nullptr, N, C, L.h[l], L.w[l], k, 1, RCX_DTYPE_F32, RCX_DTYPE_F32, s), "bwd: conv input grad");
        RCX_TRY(rcx::bwd_wgrad(F_(l), RCX_DTYPE_F32, l < level ? C_(l + 1) : nullptr, gC, PART(slot), GW(1 + j), GB(1 + j), N, C, L.h[l], L.w[l],
                               l < level ? L.h[l + 1] : 0, l < level ? L.w[l + 1] : 0, L.h[l], L.w[l], k, 1, mode, 0, s, &rows), "bwd: conv weight grad");
        add_slot(1 + j, PART(slot++), rows);
    }
    // down ladder (:27-29), coarsest first: the shared weight accumulates over all levels
    for (int l = m ? m : level; l >= 1; --l) {
        RCX_TRY(rcx::bwd_wgrad(l == 1 ? x : (const void*)F_(l - 1), l == 1 ? dtype : RCX_DTYPE_F32, nullptr, G_(l), PART(slot), GW(0), GB(0),
                               N, C, L.h[l - 1], L.w[l - 1], 0, 0, L.h[l], L.w[l], k, 2, mode, 0, s, &rows), "bwd: down weight grad");
        add_slot(0, PART(slot++), rows);
        RCX_TRY(rcx::bwd_down_input(G_(l - 1), G_(l), l == 1 ? gx : (void*)G_(l - 1), l == 1 ? dtype : RCX_DTYPE_F32, W_(0),
                                    N, C, L.h[l - 1], L.w[l - 1], L.h[l], L.w[l], k, s), "bwd: down input grad");
    }
    // every weight gradient of the block in one reduction launch (job 0 sums the down conv's levels, coarsest first); with no ladder
    // job 0 has no buffer and writes zeros -- replaced by the memset below
    if (level == 0) { J.njobs = 1; J.gw[0] = GW(1); J.gb[0] = GB(1); J.nslots[0] = J.nslots[1]; J.part[0][0] = J.part[1][0]; J.rows[0][0] = J.rows[1][0]; }
    RCX_TRY(rcx::bwd_wgrad_reduce_jobs(J, s), "bwd: weight-gradient reduction");
#undef RCX_TRY
    if (level == 0) {   // no ladder: the shared down weight is unused, its gradient is zero
        const size_t esz = gw_out ? (grad_dtype == RCX_DTYPE_F32 ? 4 : 2) : 4;
        float* gw0 = gw_out ? (float*)gw_out[0] : gwpack;
        float* gb0 = gw_out ? (gb_out ? (float*)gb_out[0] : nullptr) : gbpack;
        e = hipMemsetAsync(gw0, 0, esz * wsz, s);
        if (e == hipSuccess && gb0) e = hipMemsetAsync(gb0, 0, esz * C, s);
        if (e != hipSuccess) return hip_fail(e, "bwd: zero down grad");
    }
    return 0;
}

int rcx_dwconv2d_fwd(const void* x, void* y, const float* w_kkc, const float* bias,
                     int N, int C, int H, int W, int k, int stride, int in_dtype, int out_dtype, void* stream)
{
    if (int rc = check_common(x, y, N, C, H, W, k, in_dtype)) return rc;
    if (!known_dtype(out_dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", out_dtype);
    if (!w_kkc) return fail(RCX_ERR_BAD_ARG, "null weight");
    if (stride != 1 && stride != 2) return fail(RCX_ERR_UNSUPPORTED, "stride %d not supported (1 or 2)", stride);
    hipError_t e = step_dwconv(x, y, w_kkc, bias, N, C, H, W, k, stride, in_dtype, out_dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_dwconv2d_fwd");
}

int rcx_dwconv2d_mult2_fwd(const void* x, void* y, const float* w_kkc, const float* bias,
                           int N, int Cin, int H, int W, int k, int stride, int dtype, void* stream)
{
    if (int rc = check_common(x, y, N, Cin, H, W, k, dtype)) return rc;
    if (!w_kkc) return fail(RCX_ERR_BAD_ARG, "null weight");
    if (stride != 1 && stride != 2) return fail(RCX_ERR_UNSUPPORTED, "stride %d not supported (1 or 2)", stride);
    // Downsample's conv: the register-resident lanes kernel on the planes it was built for (28 x 28 .. 64 x 64: 256 x 64 x 56 x 56 56.7 us
    // = 0.34 of 8 TB/s against the tiled kernel's 64.3), the tiled channel-per-lane kernel (rcx_upcpt.hip, round 3) on the large lanes planes
    // (32 x 64 x 128 x 128: 55 us against 164; 64 x 64 x 112 x 112: 61 against 246), on 14 x 14 (20.6 against 25.9) and on everything the
    // lanes kernel has no plan for (float16; COCO stages 200 x 336: 17 us against the generic kernel's 62).  Never a function of N.
    hipError_t e;
    const bool lanes_ok = !lanes_off() && rcx::down_lanes_applicable(N, Cin, H, W, k, stride, dtype);
    const bool tiled_ok = !lanes_off() && rcx::down7m2_cpt_applicable(N, Cin, H, W, k, stride, dtype);
    if (tiled_ok && (!lanes_ok || H > 64 || W > 64 || (H <= 14 && W <= 14) || upcpt_everywhere()))
        e = rcx::down7m2_cpt(x, y, w_kkc, bias, N, Cin, H, W, dtype, (hipStream_t)stream);
    else if (lanes_ok)
        e = rcx::down_lanes(x, y, w_kkc, bias, N, Cin, H, W, k, stride, dtype, (hipStream_t)stream);
    else
        e = rcx::generic_dwconv_mult2(x, y, w_kkc, bias, N, Cin, H, W, k, stride, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_dwconv2d_mult2_fwd");
}

const char* rcx_upadd_dwconv_fwd_plan(int N, int C, int H, int W, int Hc, int Wc, int k, int mode, int x_dtype, int coarse_dtype, int out_dtype, int has_coarse)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || (k & 1) == 0) return "invalid";
    static thread_local char desc[160];
    switch (pick_upadd(N, C, H, W, Hc, Wc, k, x_dtype, coarse_dtype, out_dtype, has_coarse != 0)) {
    case STEP_CPL14: return H == 7 ? "upadd_cpl14(k_upadd_cpl7)" : "upadd_cpl14(k_upadd_cpl14)";
    case STEP_CPT: return rcx::upadd_cpt_describe(N, C, H, W, mode == RCX_MODE_NEAREST ? 1 : 0, x_dtype, desc, (int)sizeof(desc)) > 0 ? desc : "upadd_cpt(k_upadd_cpt)";
    case STEP_LANES: return "upadd_lanes(k_upadd_lanes)";
    case STEP_CONV5_LANES: return "conv5_lanes(k_upadd_lanes)";
    default: return "generic";
    }
}

int rcx_upadd_dwconv_fwd(const void* x, const void* coarse, void* y, const float* w_kkc, const float* bias,
                         int N, int C, int H, int W, int Hc, int Wc, int k, int mode,
                         int x_dtype, int coarse_dtype, int out_dtype, void* stream)
{
    if (int rc = check_common(x, y, N, C, H, W, k, x_dtype)) return rc;
    if (!known_dtype(out_dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", out_dtype);
    if (!w_kkc) return fail(RCX_ERR_BAD_ARG, "null weight");
    if (mode != RCX_MODE_BILINEAR && mode != RCX_MODE_NEAREST) return fail(RCX_ERR_BAD_ARG, "unknown mode %d", mode);
    if (coarse) {
        if (Hc <= 0 || Wc <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive coarse extent %dx%d", Hc, Wc);
        if (!known_dtype(coarse_dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", coarse_dtype);
    }
    hipError_t e = step_upadd(x, coarse, y, w_kkc, bias, N, C, H, W, Hc, Wc, k, mode, x_dtype, coarse_dtype, out_dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_upadd_dwconv_fwd");
}

size_t rcx_dwconv2d_bwd_workspace_bytes(int C, int k)
{
    if (C <= 0 || k <= 0 || (k & 1) == 0) return 0;
    return align256(rcx::wgrad_partial_bytes(C, k));
}

int rcx_dwconv2d_bwd(const void* x, const float* gy, const float* w_kkc, const float* w_flipped_kkc,
                     void* gx, float* gw, float* gb, void* workspace, size_t workspace_bytes,
                     int N, int C, int H, int W, int k, int stride, int x_dtype, void* stream)
{
    if (!x || !gy || !gw) return fail(RCX_ERR_BAD_ARG, "rcx_dwconv2d_bwd: null pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent N=%d C=%d H=%d W=%d", N, C, H, W);
    if (k <= 0 || (k & 1) == 0) return fail(RCX_ERR_BAD_ARG, "kernel_size must be odd and positive, got %d", k);
    if (!known_dtype(x_dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", x_dtype);
    if (stride != 1 && stride != 2) return fail(RCX_ERR_UNSUPPORTED, "stride %d not supported (1 or 2)", stride);
    if (C % 4) return fail(RCX_ERR_UNSUPPORTED, "the backward kernels need C %% 4 == 0, got C=%d", C);
    if (gx && (!w_kkc || !w_flipped_kkc)) return fail(RCX_ERR_BAD_ARG, "rcx_dwconv2d_bwd: weights are needed for the input gradient");
    const size_t need = rcx_dwconv2d_bwd_workspace_bytes(C, k);
    if (!workspace || workspace_bytes < need) return fail(RCX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int p = k / 2, Ho = (H + 2 * p - k) / stride + 1, Wo = (W + 2 * p - k) / stride + 1;
    hipError_t e;
    if (gx) {
        // the stride-2 conv5 on the 56 x 56 / 28 x 28 planes (RecAttn2d's `down` conv): the tiled adjoint kernels of rcx_recconv2d_bwd (round 6)
        const bool tiled = stride == 2 && !lanes_off() && rcx::bwd_cpt_applicable(N, C, H, W, k) && Ho * 2 == H && Wo * 2 == W;
        if (stride == 1) e = step_dwconv(gy, gx, w_flipped_kkc, nullptr, N, C, H, W, k, 1, RCX_DTYPE_F32, x_dtype, s);
        else if (tiled) e = rcx::bwd_dT_cpt(gy, gx, x_dtype, w_kkc, N, C, H, s);
        else e = rcx::bwd_down_input(nullptr, gy, gx, x_dtype, w_kkc, N, C, H, W, Ho, Wo, k, s);
        if (e != hipSuccess) return hip_fail(e, "rcx_dwconv2d_bwd: input gradient");
    }
    if (stride == 2 && !lanes_off() && rcx::bwd_cpt_applicable(N, C, H, W, k) && Ho * 2 == H && Wo * 2 == W && N * (H / 14) <= 512) {      // 512 partial rows: the workspace above
        int rows = 0;
        e = rcx::bwd_wgrad_d_cpt(x, x_dtype, gy, (float*)workspace, N, C, H, s, &rows);
        if (e == hipSuccess) {
            rcx::WgradJobs J{};
            J.njobs = 1; J.kk = k * k; J.C = C;
            J.nslots[0] = 1; J.part[0][0] = (const float*)workspace; J.rows[0][0] = rows; J.gw[0] = gw; J.gb[0] = gb;
            e = rcx::bwd_wgrad_reduce_jobs(J, s);
        }
        return e == hipSuccess ? 0 : hip_fail(e, "rcx_dwconv2d_bwd: weight gradient");
    }
    e = rcx::bwd_wgrad(x, x_dtype, nullptr, gy, (float*)workspace, gw, gb, N, C, H, W, 0, 0, Ho, Wo, k, stride, 0, 0, s);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_dwconv2d_bwd: weight gradient");
}

// ---- backward of conv(x + resize(coarse)) (RecAttn2d's last line in a training step) ----
static bool upadd_bwd_tiled(int N, int C, int H, int W, int Hc, int Wc, int k)
{
    return !lanes_off() && rcx::bwd_cpt_applicable(N, C, H, W, k) && Hc * 2 == H && Wc * 2 == W;
}

size_t rcx_upadd_dwconv_bwd_workspace_bytes(int N, int C, int H, int W, int Hc, int Wc, int k)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || (k & 1) == 0) return 0;
    const size_t part = align256(rcx::wgrad_partial_bytes(C, k));
    if (upadd_bwd_tiled(N, C, H, W, Hc, Wc, k)) return align256(sizeof(float) * (size_t)N * ((H + 13) / 14) * (size_t)(k * k + 1) * C);      // one partial row per (image, band)
    return part + align256(sizeof(float) * (size_t)N * C * H * W);      // + the float32 gT the resize adjoint reads
}

int rcx_upadd_dwconv_bwd_gy_dtype(int N, int C, int H, int W, int Hc, int Wc, int k, int dtype)
{
    if (dtype != RCX_DTYPE_BF16 || N <= 0 || C <= 0 || C % 4 || k <= 0 || (k & 1) == 0) return RCX_DTYPE_F32;
    return upadd_bwd_tiled(N, C, H, W, Hc, Wc, k) ? dtype : RCX_DTYPE_F32;
}

int rcx_upadd_dwconv_bwd(const void* x, const float* coarse, const void* gy, int gy_dtype, const float* w_kkc, const float* w_flipped_kkc,
                         void* gx, float* gcoarse, float* gw, float* gb, void* workspace, size_t workspace_bytes,
                         int N, int C, int H, int W, int Hc, int Wc, int k, int mode, int dtype, void* stream)
{
    if (!x || !coarse || !gy || !gw || !w_flipped_kkc) return fail(RCX_ERR_BAD_ARG, "rcx_upadd_dwconv_bwd: null pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent N=%d C=%d H=%d W=%d Hc=%d Wc=%d", N, C, H, W, Hc, Wc);
    if (k <= 0 || (k & 1) == 0) return fail(RCX_ERR_BAD_ARG, "kernel_size must be odd and positive, got %d", k);
    if (!known_dtype(dtype) || !known_dtype(gy_dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d / %d", dtype, gy_dtype);
    if (mode != RCX_MODE_BILINEAR && mode != RCX_MODE_NEAREST) return fail(RCX_ERR_BAD_ARG, "unknown mode %d", mode);
    if (C % 4) return fail(RCX_ERR_UNSUPPORTED, "the backward kernels need C %% 4 == 0, got C=%d", C);
    const size_t need = rcx_upadd_dwconv_bwd_workspace_bytes(N, C, H, W, Hc, Wc, k);
    if (!workspace || workspace_bytes < need) return fail(RCX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const bool tiled = upadd_bwd_tiled(N, C, H, W, Hc, Wc, k);
    if (gy_dtype != RCX_DTYPE_F32 && !(tiled && gy_dtype == dtype && dtype == RCX_DTYPE_BF16))
        return fail(RCX_ERR_UNSUPPORTED, "gy of dtype %d: this problem takes float32 (rcx_upadd_dwconv_bwd_gy_dtype)", gy_dtype);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    hipError_t e;
    if (tiled) {
        const int md = mode == RCX_MODE_NEAREST ? 1 : 0;
        if (gcoarse) {
            e = rcx::bwd_gc_cpt(gy, gy_dtype, gcoarse, w_flipped_kkc, N, C, H, md, s);
            if (e != hipSuccess) return hip_fail(e, "rcx_upadd_dwconv_bwd: coarse gradient");
        }
        if (gx) {
            e = rcx::bwd_gx_cpt(gy, gy_dtype, nullptr, gx, dtype, w_flipped_kkc, nullptr, N, C, H, s);
            if (e != hipSuccess) return hip_fail(e, "rcx_upadd_dwconv_bwd: input gradient");
        }
        int rows = 0;
        e = rcx::bwd_wgrad_k_cpt(x, dtype, coarse, gy, gy_dtype, part, N, C, H, md, s, &rows);
        if (e == hipSuccess) {
            rcx::WgradJobs J{};
            J.njobs = 1; J.kk = k * k; J.C = C;
            J.nslots[0] = 1; J.part[0][0] = part; J.rows[0][0] = rows; J.gw[0] = gw; J.gb[0] = gb;
            e = rcx::bwd_wgrad_reduce_jobs(J, s);
        }
        return e == hipSuccess ? 0 : hip_fail(e, "rcx_upadd_dwconv_bwd: weight gradient");
    }
    const float* gyf = (const float*)gy;
    float* gT = (float*)((char*)workspace + align256(rcx::wgrad_partial_bytes(C, k)));
    if (gcoarse) {
        e = step_dwconv(gyf, gT, w_flipped_kkc, nullptr, N, C, H, W, k, 1, RCX_DTYPE_F32, RCX_DTYPE_F32, s);
        if (e == hipSuccess) e = rcx::bwd_resize(gT, gcoarse, N, C, H, W, Hc, Wc, mode, s);
        if (e != hipSuccess) return hip_fail(e, "rcx_upadd_dwconv_bwd: coarse gradient");
    }
    if (gx) {
        if (gcoarse && dtype == RCX_DTYPE_F32) e = hipMemcpyAsync(gx, gT, sizeof(float) * (size_t)N * C * H * W, hipMemcpyDeviceToDevice, s);
        else e = step_dwconv(gyf, gx, w_flipped_kkc, nullptr, N, C, H, W, k, 1, RCX_DTYPE_F32, dtype, s);
        if (e != hipSuccess) return hip_fail(e, "rcx_upadd_dwconv_bwd: input gradient");
    }
    e = rcx::bwd_wgrad(x, dtype, coarse, gyf, part, gw, gb, N, C, H, W, Hc, Wc, H, W, k, 1, mode, 0, s);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_upadd_dwconv_bwd: weight gradient");
}

int rcx_dwconv2d_mult2_bwd(const void* x, const float* gy, const float* w_kkc, void* gx, float* gw, float* gb,
                           void* workspace, size_t workspace_bytes, int N, int Cin, int H, int W, int k, int dtype, void* stream)
{
    if (!x || !gy || !gw) return fail(RCX_ERR_BAD_ARG, "rcx_dwconv2d_mult2_bwd: null pointer");
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent N=%d C=%d H=%d W=%d", N, Cin, H, W);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (k != 3 && k != 5 && k != 7) return fail(RCX_ERR_UNSUPPORTED, "kernel_size %d not supported by the multiplier-2 backward (3, 5, 7)", k);
    if (Cin % 2) return fail(RCX_ERR_UNSUPPORTED, "the multiplier-2 backward needs an even channel count, got %d", Cin);
    if (gx && !w_kkc) return fail(RCX_ERR_BAD_ARG, "rcx_dwconv2d_mult2_bwd: weights are needed for the input gradient");
    const size_t need = rcx_dwconv2d_bwd_workspace_bytes(2 * Cin, k);
    if (!workspace || workspace_bytes < need) return fail(RCX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    hipError_t e = rcx::bwd_mult2(x, dtype, gy, w_kkc, gx, (float*)workspace, gw, gb, N, Cin, H, W, k, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_dwconv2d_mult2_bwd");
}

int rcx_linear_attention_fwd(const void* qpre, const void* kpre, const void* v, const void* pe, void* out,
                             int B, int n, int C, int heads, int dtype, void* stream)
{
    if (!qpre || !kpre || !v || !pe || !out) return fail(RCX_ERR_BAD_ARG, "rcx_linear_attention_fwd: null pointer");
    if (B <= 0 || n <= 0 || C <= 0 || heads <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent B=%d n=%d C=%d heads=%d", B, n, C, heads);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (C % heads) return fail(RCX_ERR_BAD_ARG, "C=%d is not a multiple of heads=%d", C, heads);
    const int D = C / heads;
    if (D > 64 || (D % 4 != 0 && D > 32)) return fail(RCX_ERR_UNSUPPORTED, "head dimension %d not supported (at most 64; at most 32 unless a multiple of 4)", D);
    hipError_t e = rcx::linattn_core(qpre, kpre, v, pe, out, B, n, C, heads, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_linear_attention_fwd");
}

int rcx_linear_attention_pe_fwd(const void* qpre, const void* kpre, const void* v, const float* w_pe_kkc, const float* b_pe, void* out,
                                int B, int H, int W, int C, int heads, int dtype, void* stream)
{
    if (!qpre || !kpre || !v || !w_pe_kkc || !out) return fail(RCX_ERR_BAD_ARG, "rcx_linear_attention_pe_fwd: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent B=%d H=%d W=%d C=%d heads=%d", B, H, W, C, heads);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (C % heads) return fail(RCX_ERR_BAD_ARG, "C=%d is not a multiple of heads=%d", C, heads);
    const int D = C / heads;
    if (D > 64 || !rcx::linattn_core_fuses_pe(H * W, C, heads, dtype))
        return fail(RCX_ERR_UNSUPPORTED, "head dimension %d, %d tokens: the fused form exists on the vector-pipe kernel of head dimensions that are multiples of "
                                         "four, at most 64 (use rcx_dwconv2d_fwd + rcx_linear_attention_fwd)", D, H * W);
    hipError_t e = rcx::linattn_core(qpre, kpre, v, nullptr, out, B, H * W, C, heads, dtype, (hipStream_t)stream, w_pe_kkc, b_pe, W);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_linear_attention_pe_fwd");
}

int rcx_recattn_qkcore_launches(int B, int H, int W, int C, int heads)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    return rcx::recattn_qkcore_launches(B, H, W, C, heads);
}

size_t rcx_recattn_qkcore_workspace_bytes(int B, int H, int W, int C, int heads)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    return rcx::recattn_qkcore_workspace_bytes(B, H, W, C, heads);
}

int rcx_recattn_qkcore_fwd(const float* d, const void* wqk_bf16, const float* bqk, const float* w_pe_kkc, const float* b_pe, float* out,
                           void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, void* stream)
{
    if (!d || !wqk_bf16 || !bqk || !w_pe_kkc || !out) return fail(RCX_ERR_BAD_ARG, "rcx_recattn_qkcore_fwd: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent B=%d H=%d W=%d C=%d heads=%d", B, H, W, C, heads);
    if (C % heads) return fail(RCX_ERR_BAD_ARG, "C=%d is not a multiple of heads=%d", C, heads);
    if (((size_t)d & 15) || ((size_t)wqk_bf16 & 15) || ((size_t)out & 15) || ((size_t)bqk & 15) || ((size_t)w_pe_kkc & 15) || ((size_t)b_pe & 15))
        return fail(RCX_ERR_BAD_ARG, "rcx_recattn_qkcore_fwd: d, wqk, bqk, w_pe_kkc, b_pe and out must be 16-byte aligned");
    if (!rcx::recattn_qkcore_applicable(B, H, W, C, heads))
        return fail(RCX_ERR_UNSUPPORTED, "rcx_recattn_qkcore_fwd: head dimension %d, %d heads, %d tokens, C=%d: the matrix-core form takes 32-wide heads, 1, 2, 4, 8 or "
                                         "16 of them; 16 only on planes of at most 32 tokens (rcx_recattn_qkcore_launches; use the projection GEMMs + "
                                         "rcx_linear_attention_pe_fwd)", C / heads, heads, H * W, C);
    const size_t need = rcx::recattn_qkcore_workspace_bytes(B, H, W, C, heads);
    if (need && (!workspace || workspace_bytes < need || ((size_t)workspace & 15)))
        return fail(RCX_ERR_WORKSPACE, "rcx_recattn_qkcore_fwd: needs a 16-byte-aligned workspace of %zu bytes, got %zu", need, workspace ? workspace_bytes : (size_t)0);
    hipError_t e = rcx::recattn_qkcore(d, wqk_bf16, bqk, w_pe_kkc, b_pe, out, workspace, B, H, W, C, heads, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_recattn_qkcore_fwd");
}

int rcx_recattn_down_qkcore_supported(int B, int H, int W, int C, int heads, int x_dtype)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    return rcx::recattn_down_qkcore_applicable(B, H, W, C, heads, x_dtype) ? 1 : 0;
}

int rcx_recattn_down_qkcore_fwd(const void* x, const float* w_down_kkc, const float* b_down, const void* wqk_bf16, const float* bqk,
                                const float* w_pe_kkc, const float* b_pe, float* out, int B, int H, int W, int C, int heads, int x_dtype, void* stream)
{
    if (!x || !w_down_kkc || !wqk_bf16 || !bqk || !w_pe_kkc || !out) return fail(RCX_ERR_BAD_ARG, "rcx_recattn_down_qkcore_fwd: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent B=%d H=%d W=%d C=%d heads=%d", B, H, W, C, heads);
    if (C % heads) return fail(RCX_ERR_BAD_ARG, "C=%d is not a multiple of heads=%d", C, heads);
    if (!known_dtype(x_dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", x_dtype);
    if (((size_t)wqk_bf16 & 15) || ((size_t)out & 15) || ((size_t)bqk & 15) || ((size_t)w_pe_kkc & 15) || ((size_t)b_pe & 15))
        return fail(RCX_ERR_BAD_ARG, "rcx_recattn_down_qkcore_fwd: wqk, bqk, w_pe_kkc, b_pe and out must be 16-byte aligned");
    if (!rcx::recattn_down_qkcore_applicable(B, H, W, C, heads, x_dtype))
        return fail(RCX_ERR_UNSUPPORTED, "rcx_recattn_down_qkcore_fwd: %d x %d plane, %d heads of %d, dtype %d: the one-launch form takes the 14 x 14 plane (1 .. 8 heads) "
                                         "and the 7 x 7 plane (1 .. 16 heads) of bf16 / f16 activations, 32-wide heads (use rcx_dwconv2d_fwd + rcx_recattn_qkcore_fwd)",
                    H, W, heads, C / heads, x_dtype);
    hipError_t e = rcx::recattn_down_qkcore(x, w_down_kkc, b_down, wqk_bf16, bqk, w_pe_kkc, b_pe, out, B, H, C, heads, x_dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_recattn_down_qkcore_fwd");
}

int rcx_recattn2d_fwd_supported(int B, int H, int W, int C, int heads, int mode, int dtype)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    return rcx::recattn2d_unit_applicable(B, H, W, C, heads, dtype, mode == RCX_MODE_NEAREST ? 1 : 0) ? 1 : 0;
}

int rcx_recattn2d_fwd(const void* x, void* y, const float* w_down_kkc, const float* b_down, const void* wqk_bf16, const float* bqk,
                      const float* w_pe_kkc, const float* b_pe, const float* w_conv_kkc, const float* b_conv,
                      int B, int H, int W, int C, int heads, int mode, int dtype, void* stream)
{
    if (!x || !y || !w_down_kkc || !wqk_bf16 || !bqk || !w_pe_kkc || !w_conv_kkc) return fail(RCX_ERR_BAD_ARG, "rcx_recattn2d_fwd: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent B=%d H=%d W=%d C=%d heads=%d", B, H, W, C, heads);
    if (C % heads) return fail(RCX_ERR_BAD_ARG, "C=%d is not a multiple of heads=%d", C, heads);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (mode != RCX_MODE_BILINEAR && mode != RCX_MODE_NEAREST) return fail(RCX_ERR_BAD_ARG, "unknown mode %d", mode);
    if (((size_t)wqk_bf16 & 15) || ((size_t)bqk & 15) || ((size_t)w_pe_kkc & 15) || ((size_t)b_pe & 15))
        return fail(RCX_ERR_BAD_ARG, "rcx_recattn2d_fwd: wqk, bqk, w_pe_kkc and b_pe must be 16-byte aligned");
    if (!rcx::recattn2d_unit_applicable(B, H, W, C, heads, dtype, mode == RCX_MODE_NEAREST ? 1 : 0))
        return fail(RCX_ERR_UNSUPPORTED, "rcx_recattn2d_fwd: %d x %d plane, %d heads of %d, mode %d, dtype %d: the one-launch unit takes the 14 x 14 (1 .. 8 heads) and 7 x 7 (1 .. 16 heads) planes of bf16 / f16 "
                                         "activations, heads of 32 or 4 .. 28 channels, nearest resize (use rcx_recattn_down_qkcore_fwd / rcx_dwconv2d_fwd + rcx_recattn_qkcore_fwd, then "
                                         "rcx_upadd_dwconv_fwd)", H, W, heads, C / heads, mode, dtype);
    hipError_t e = rcx::recattn2d_unit(x, w_down_kkc, b_down, wqk_bf16, bqk, w_pe_kkc, b_pe, w_conv_kkc, b_conv, y, B, H, C, heads, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_recattn2d_fwd");
}

int rcx_stem_supported(int N, int H, int W, int CM, int CO, int dtype) { return rcx::stem_applicable(N, H, W, CM, CO, dtype) ? 1 : 0; }

size_t rcx_stem_pack_bytes(int CM, int CO) { return rcx::stem_pack_bytes(CM, CO); }

int rcx_stem_fwd(const void* x, void* y, const void* w1frag, const float* b1, const void* w2frag, const float* b2, int N, int H, int W, int CM, int CO, int dtype, void* stream)
{
    if (!x || !y || !w1frag || !b1 || !w2frag || !b2) return fail(RCX_ERR_BAD_ARG, "rcx_stem_fwd: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || CM <= 0 || CO <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent N=%d H=%d W=%d CM=%d CO=%d", N, H, W, CM, CO);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (((size_t)x & 1) || ((size_t)y & 7) || ((size_t)w1frag & 15) || ((size_t)b1 & 15) || ((size_t)w2frag & 15) || ((size_t)b2 & 15))
        return fail(RCX_ERR_BAD_ARG, "rcx_stem_fwd: w1frag, b1, w2frag and b2 must be 16-byte aligned, y 8-byte aligned");
    if (!rcx::stem_applicable(N, H, W, CM, CO, dtype))
        return fail(RCX_ERR_UNSUPPORTED, "rcx_stem_fwd: no kernel for CM=%d CO=%d dtype %d (bf16; CM in {20, 24, 28, 32, 40}, CO %% 4 == 0, CO <= 96; one image of x below 2^31 bytes)", CM, CO, dtype);
    hipError_t e = rcx::stem_fwd(x, y, w1frag, b1, w2frag, b2, N, H, W, CM, CO, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_stem_fwd");
}

int rcx_channel_mlp_supported(int M, int C, int H, int dtype) { return rcx::channel_mlp_applicable(M, C, H, dtype) ? 1 : 0; }

size_t rcx_channel_mlp_pack_bytes(int C, int H) { return rcx::channel_mlp_pack_bytes(C, H); }

int rcx_channel_mlp_fwd(const void* z, const void* x, void* y, const void* wfrag, const float* bias, int M, int C, int H, int dtype, void* stream)
{
    if (!z || !x || !y || !wfrag || !bias) return fail(RCX_ERR_BAD_ARG, "rcx_channel_mlp_fwd: null pointer");
    if (M <= 0 || C <= 0 || H <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent M=%d C=%d H=%d", M, C, H);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (((size_t)z & 15) || ((size_t)x & 7) || ((size_t)y & 7) || ((size_t)wfrag & 15) || ((size_t)bias & 15))
        return fail(RCX_ERR_BAD_ARG, "rcx_channel_mlp_fwd: z, wfrag and bias must be 16-byte aligned, x and y 8-byte aligned");
    if (y == z || y == x) return fail(RCX_ERR_BAD_ARG, "rcx_channel_mlp_fwd: y must not alias its inputs");
    if (!rcx::channel_mlp_applicable(M, C, H, dtype))
        return fail(RCX_ERR_UNSUPPORTED, "rcx_channel_mlp_fwd: no kernel for M=%d C=%d H=%d dtype %d (bf16; (C, H) = (40 | 48, 96), (56 | 64, 128), (80, 160), (96, 192), (128, 256), (160, 320), "
                                         "(192, 384), (256, 512), (320, 640); M C 2 < 2^31)", M, C, H, dtype);
    hipError_t e = rcx::channel_mlp(z, x, y, wfrag, bias, M, C, H, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_channel_mlp_fwd");
}

int rcx_linear_attention_bwd(const void* qpre, const void* kpre, const void* v, const void* gout, void* gq, void* gk, void* gv,
                             int B, int n, int C, int heads, int dtype, void* stream)
{
    if (!qpre || !kpre || !v || !gout || !gq || !gk || !gv) return fail(RCX_ERR_BAD_ARG, "rcx_linear_attention_bwd: null pointer");
    if (B <= 0 || n <= 0 || C <= 0 || heads <= 0) return fail(RCX_ERR_BAD_ARG, "non-positive extent B=%d n=%d C=%d heads=%d", B, n, C, heads);
    if (!known_dtype(dtype)) return fail(RCX_ERR_BAD_ARG, "unknown dtype %d", dtype);
    if (C % heads) return fail(RCX_ERR_BAD_ARG, "C=%d is not a multiple of heads=%d", C, heads);
    if (C / heads > 64) return fail(RCX_ERR_UNSUPPORTED, "head dimension %d not supported (at most 64)", C / heads);
    hipError_t e = rcx::linattn_core_bwd(qpre, kpre, v, gout, gq, gk, gv, B, n, C, heads, dtype, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "rcx_linear_attention_bwd");
}

#ifdef RCX_STAMPS
/* diagnostic build only: not part of include/recnext_amd.h */
int rcx_debug_set_stamp_buffer(void* p) { int e = (int)rcx::set_stamp_buffer(p); return e ? e : (int)rcx::lanes::set_stamp_buffer(p); }
#endif

}  // extern "C"
