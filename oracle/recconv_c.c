/*
 * Plain-C restatement of the RecConv2d hot path -- TEST INFRASTRUCTURE ONLY (the oracle).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file; the product (recnext_amd/) never does.
 *
 * Parity status: PINNED -- checked against the golden vectors produced by importing the
 * reference itself (tests/golden/make_golden.py) in tests/test_oracle.py.
 *
 * Reference anchors (relative to the upstream repository root):
 *   model/recnext.py:24-34   control flow: down ladder, coarsest-first up recursion, final conv
 *   model/recnext.py:13-22   conv parameters: depthwise, pad k/2, stride 2 (down) or 1 (convs)
 *   model/recnext.py:33      resize to the recorded size of the finer level, mode bilinear|nearest
 * Conv / interpolate arithmetic is PyTorch ATen's (third party, unpinned: requirements.txt:1);
 * the published semantics restated here are: cross-correlation with zero padding;
 * bilinear align_corners=False with src = max(scale*(d+0.5)-0.5, 0) computed in float;
 * legacy nearest i = min(floor(d*scale), in-1).
 *
 * Memory layout: activations NHWC float32 (what a torch.channels_last tensor stores);
 * weights in the reference's own layout (C,1,k,k), biases (C).  Sums are carried in
 * double and rounded to float once per op, so the result is a slightly *better* float
 * answer than ATen's float accumulation; both agree to ~1e-6 on N(0,1) data.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RCX_ORACLE_MAX_LEVEL 16

static int down_size(int h, int k) { int p = k / 2; return (h + 2 * p - k) / 2 + 1; }

/* out[n,oy,ox,c] = b[c] + sum_{u,v} w[c,u,v] * in[n, s*oy+u-p, s*ox+v-p, c]   (zero outside) */
static void dwconv_nhwc(const float* in, float* out, const float* w, const float* b,
                        int N, int C, int H, int W, int k, int stride)
{
    const int p = k / 2;
    const int Ho = (H + 2 * p - k) / stride + 1, Wo = (W + 2 * p - k) / stride + 1;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            double* acc = (double*)malloc(sizeof(double) * (size_t)C);
            for (int ox = 0; ox < Wo; ++ox) {
                for (int c = 0; c < C; ++c) acc[c] = b ? (double)b[c] : 0.0;
                for (int u = 0; u < k; ++u) {
                    int iy = stride * oy + u - p;
                    if (iy < 0 || iy >= H) continue;
                    for (int v = 0; v < k; ++v) {
                        int ix = stride * ox + v - p;
                        if (ix < 0 || ix >= W) continue;
                        const float* px = in + (((size_t)n * H + iy) * W + ix) * C;
                        const float* pw = w + (size_t)u * k + v;           /* + c*k*k */
                        for (int c = 0; c < C; ++c) acc[c] += (double)pw[(size_t)c * k * k] * (double)px[c];
                    }
                }
                float* po = out + (((size_t)n * Ho + oy) * Wo + ox) * C;
                for (int c = 0; c < C; ++c) po[c] = (float)acc[c];
            }
            free(acc);
        }
}

static void bilinear_table(int n_in, int n_out, int* i0, int* i1, float* lam)
{
    const float scale = (float)n_in / (float)n_out;
    for (int d = 0; d < n_out; ++d) {
        float src = scale * ((float)d + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
        int a = (int)floorf(src);
        if (a > n_in - 1) a = n_in - 1;
        i0[d] = a;
        i1[d] = a + (a < n_in - 1 ? 1 : 0);
        lam[d] = src - (float)a;
    }
}

static void nearest_table(int n_in, int n_out, int* idx)
{
    const float scale = (float)n_in / (float)n_out;
    for (int d = 0; d < n_out; ++d) {
        int a = (int)floorf((float)d * scale);
        idx[d] = a < n_in - 1 ? a : n_in - 1;
    }
}

/* dst[n,y,x,c] = base[n,y,x,c] + resize(src -> (Ho,Wo))[n,y,x,c]; base may alias dst */
static void add_resized_nhwc(const float* base, const float* src, float* dst,
                             int N, int C, int Hi, int Wi, int Ho, int Wo, int mode)
{
    int *y0 = malloc(sizeof(int) * Ho), *y1 = malloc(sizeof(int) * Ho);
    int *x0 = malloc(sizeof(int) * Wo), *x1 = malloc(sizeof(int) * Wo);
    float *ly = malloc(sizeof(float) * Ho), *lx = malloc(sizeof(float) * Wo);
    if (mode == 0) { bilinear_table(Hi, Ho, y0, y1, ly); bilinear_table(Wi, Wo, x0, x1, lx); }
    else { nearest_table(Hi, Ho, y0); nearest_table(Wi, Wo, x0); }
    #pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                size_t o = (((size_t)n * Ho + y) * Wo + x) * C;
                if (mode == 0) {
                    const float* a = src + (((size_t)n * Hi + y0[y]) * Wi + x0[x]) * C;
                    const float* b = src + (((size_t)n * Hi + y0[y]) * Wi + x1[x]) * C;
                    const float* c_ = src + (((size_t)n * Hi + y1[y]) * Wi + x0[x]) * C;
                    const float* d = src + (((size_t)n * Hi + y1[y]) * Wi + x1[x]) * C;
                    double wy = ly[y], wx = lx[x];
                    for (int c = 0; c < C; ++c) {
                        double r = (1.0 - wy) * ((1.0 - wx) * a[c] + wx * b[c]) + wy * ((1.0 - wx) * c_[c] + wx * d[c]);
                        dst[o + c] = (float)((double)base[o + c] + (double)(float)r);
                    }
                } else {
                    const float* a = src + (((size_t)n * Hi + y0[y]) * Wi + x0[x]) * C;
                    for (int c = 0; c < C; ++c) dst[o + c] = (float)((double)base[o + c] + (double)a[c]);
                }
            }
    free(y0); free(y1); free(x0); free(x1); free(ly); free(lx);
}

int rcx_oracle_abi_version(void) { return 1; }

/* Stand-alone pieces, exported for unit tests. */
void rcx_oracle_dwconv2d_nhwc_f32(const float* in, float* out, const float* w, const float* b,
                                  int N, int C, int H, int W, int k, int stride)
{
    dwconv_nhwc(in, out, w, b, N, C, H, W, k, stride);
}

void rcx_oracle_add_resized_nhwc_f32(const float* base, const float* src, float* dst,
                                     int N, int C, int Hi, int Wi, int Ho, int Wo, int mode)
{
    add_resized_nhwc(base, src, dst, N, C, Hi, Wi, Ho, Wo, mode);
}

/*
 * RecConv2d.forward (model/recnext.py:24-34).
 * w_convs: (level+1, C,1,k,k) packed, convs[0] pairs with the coarsest feature; b_* nullable.
 * mode: 0 bilinear, 1 nearest.  threads <= 0 keeps the OpenMP default.  Returns 0 / -1 (bad args).
 */
int rcx_oracle_recconv2d_nhwc_f32(const float* x, float* y,
                                  const float* w_down, const float* b_down,
                                  const float* w_convs, const float* b_convs,
                                  int N, int C, int H, int W, int level, int k, int mode, int threads)
{
    if (level < 0 || level > RCX_ORACLE_MAX_LEVEL || (k & 1) == 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    int hs[RCX_ORACLE_MAX_LEVEL + 1], ws[RCX_ORACLE_MAX_LEVEL + 1];
    float* F[RCX_ORACLE_MAX_LEVEL + 1];
    hs[0] = H; ws[0] = W; F[0] = (float*)x;
    for (int l = 1; l <= level; ++l) {                      /* down ladder, shared weight (:27-29) */
        hs[l] = down_size(hs[l - 1], k); ws[l] = down_size(ws[l - 1], k);
        F[l] = (float*)malloc(sizeof(float) * (size_t)N * C * hs[l] * ws[l]);
        dwconv_nhwc(F[l - 1], F[l], w_down, b_down, N, C, hs[l - 1], ws[l - 1], k, 2);
    }
    const size_t wsz = (size_t)C * k * k;
    float* cv = NULL;                                       /* conv output of the level below, pre-resize */
    for (int l = level, j = 0; l >= 1; --l, ++j) {          /* coarsest first (:32-33) */
        size_t cnt = (size_t)N * C * hs[l] * ws[l];
        if (cv) {                                           /* T_l = F_l + resize(C_{l+1}) in place */
            add_resized_nhwc(F[l], cv, F[l], N, C, hs[l + 1], ws[l + 1], hs[l], ws[l], mode);
            free(cv);
        }
        cv = (float*)malloc(sizeof(float) * cnt);
        dwconv_nhwc(F[l], cv, w_convs + j * wsz, b_convs ? b_convs + (size_t)j * C : NULL, N, C, hs[l], ws[l], k, 1);
    }
    const float* t0 = x;
    float* tmp = NULL;
    if (cv) {
        tmp = (float*)malloc(sizeof(float) * (size_t)N * C * H * W);
        add_resized_nhwc(x, cv, tmp, N, C, hs[1], ws[1], H, W, mode);
        free(cv);
        t0 = tmp;
    }
    dwconv_nhwc(t0, y, w_convs + (size_t)level * wsz, b_convs ? b_convs + (size_t)level * C : NULL, N, C, H, W, k, 1);  /* (:34) */
    free(tmp);
    for (int l = 1; l <= level; ++l) free(F[l]);
    return 0;
}
