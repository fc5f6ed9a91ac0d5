/*
 * oflk_tolerant_model.c -- CPU model of the library's WITHIN-TOLERANCE arithmetic (OFLK_ARITH_TOLERANT).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT (same rules as oflk_oracle.c: only tests/, tools/ and
 * bench.py's checker legs load it).  It is NOT a restatement of the reference: it states, operation for
 * operation, the cheaper arithmetic the HIP kernels use when a plan is switched to the tolerant mode, with a
 * switch per stage x pyramid level x iteration, so that
 *   (1) tools/experiments/fast_mode_ablation.py can measure, on the CPU and with everything else bit-exact,
 *       what each relaxation costs in endpoint error against the reference's flow, cell by cell, and
 *   (2) tests can hold the tolerant HIP kernels to this model bit for bit -- the tolerance then only has to be
 *       established once, between this model and the reference-made dense flows of tests/golden/.
 * The reference functions it deviates from: python/lucas_kanade_core.py:110-133 (window sums, solve),
 * python/lucas_kanade_pyramidal.py:46-59 (pyramid), :88-96 (warp), :126-136 (flow upsample).
 *
 * Built with -ffp-contract=off: a fused multiply-add happens exactly where fma()/fmaf() is written.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OFLK_EXPORT __attribute__((visibility("default")))

/* exact pieces, from oflk_oracle.c */
void oflk_oracle_compute_gradients(const float *, const float *, int, int, float *, float *, float *);
void oflk_oracle_from_gradients(const float *, const float *, const float *, int, int, int, float *, float *);
void oflk_oracle_gaussian_filter(const float *, int, int, double, float *);
void oflk_oracle_resample_linspace(const float *, int, int, int, int, float *);
void oflk_oracle_pyramid_dims(int, int, int, double, int *);
void oflk_oracle_warp(const float *, const float *, const float *, int, int, float *);
void oflk_oracle_upsample_flow(const float *, const float *, int, int, int, int, float *, float *);
float oflk_oracle_mean_abs(const float *, size_t);
int oflk_oracle_gaussian_kernel1d(double, double *);

/* ---- variants ------------------------------------------------------------------------------------------- */
enum { WARP_EXACT = 0, WARP_LERP64 = 1, WARP_F32 = 2, WARP_FRAC32_LERP64 = 3 };
enum { SUMS_NUMPY = 0, SUMS_SEPARABLE = 1, SUMS_SEP_VFIRST = 2 };
enum { SOLVE_EXACT = 0, SOLVE_SHARED_RCP = 1, SOLVE_FMA_DET = 2 };
enum { PYR_EXACT = 0, PYR_CONTRACTED = 1, PYR_F32 = 2 };
enum { UP_EXACT = 0, UP_F32 = 1, UP_LERP64 = 2 };

static inline int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

static inline double linspace_at(int i, int S, int T)
{
    double delta = (double)(S - 1);
    if (T <= 1) return 0.0 * delta;
    if (i == T - 1) return delta;
    double div = (double)(T - 1);
    double step = delta / div;
    if (step == 0.0) return ((double)i / div) * delta;
    return (double)i * step;
}

/* ---- pyramid -------------------------------------------------------------------------------------------- */
/* PYR_CONTRACTED: SciPy's sums with the multiply and the add of a tap fused (k_pyr_down<PIX, true>):
 *   t = x[c]*w0; for k = r..1: t = fma(x[c-k] + x[c+k], w[k], t); fp32 store after each axis; the linspace
 *   sampling fuses its three additions too (resample_contracted). */
static void blur_axis(const float *in, float *out, int H, int W, int axis, const double *w, int radius, int variant)
{
    const int len = axis == 0 ? H : W, nl = axis == 0 ? W : H;
    double *line = (double *)malloc(sizeof(double) * (size_t)(len + 2 * radius));
    for (int l = 0; l < nl; l++) {
        for (int i = -radius; i < len + radius; i++) {
            int j = reflect_idx(i, len);
            line[i + radius] = (double)(axis == 0 ? in[(size_t)j * W + l] : in[(size_t)l * W + j]);
        }
        for (int c = 0; c < len; c++) {
            const double *p = line + c + radius;
            float r;
            if (variant == PYR_F32) {
                float t = (float)p[0] * (float)w[0];
                for (int k = radius; k >= 1; k--) t = fmaf((float)p[-k] + (float)p[k], (float)w[k], t);
                r = t;
            } else {
                double t = p[0] * w[0];
                for (int k = radius; k >= 1; k--) t = fma(p[-k] + p[k], w[k], t);
                r = (float)t;
            }
            if (axis == 0) out[(size_t)c * W + l] = r;
            else out[(size_t)l * W + c] = r;
        }
    }
    free(line);
}

static float lerp2_f32(float a, float b, float c, float d, float rx, float ry)
{
    const float top = fmaf(rx, b - a, a), bot = fmaf(rx, d - c, c);
    return fmaf(ry, bot - top, top);
}

static float lerp2_f64(float a, float b, float c, float d, double rx, double ry)
{
    const double top = fma(rx, (double)b - (double)a, (double)a), bot = fma(rx, (double)d - (double)c, (double)c);
    return (float)fma(ry, bot - top, top);
}

/* linspace sampling of `in` with fp32 / fused fp64 bilinear arithmetic (coordinates stay NumPy's fp64 linspace) */
static void resample_fast(const float *in, int H, int W, int Ho, int Wo, float *out, int f64)
{
    for (int i = 0; i < Ho; i++) {
        const double y = linspace_at(i, H, Ho);
        double fy = floor(y);
        if (fy > (double)(H - 2)) fy = (double)(H > 1 ? H - 2 : 0);
        const int y0 = (int)fy, y1 = H > 1 ? y0 + 1 : y0;
        const double ry = y - fy;
        for (int j = 0; j < Wo; j++) {
            const double x = linspace_at(j, W, Wo);
            double fx = floor(x);
            if (fx > (double)(W - 2)) fx = (double)(W > 1 ? W - 2 : 0);
            const int x0 = (int)fx, x1 = W > 1 ? x0 + 1 : x0;
            const double rx = x - fx;
            const float a = in[(size_t)y0 * W + x0], b = in[(size_t)y0 * W + x1], c = in[(size_t)y1 * W + x0],
                        d = in[(size_t)y1 * W + x1];
            out[(size_t)i * Wo + j] = f64 ? lerp2_f64(a, b, c, d, rx, ry) : lerp2_f32(a, b, c, d, (float)rx, (float)ry);
        }
    }
}

/* the linspace sampling of a contracted pyramid step (stage D of k_pyr_down<PIX, true>, k_resample<1, true>): SciPy's
 * weights and tap order, the three additions fused with the last multiply of their term */
static void resample_contracted(const float *in, int H, int W, int Ho, int Wo, float *out)
{
    for (int i = 0; i < Ho; i++) {
        const double y = linspace_at(i, H, Ho), fy = floor(y);
        const int y0 = (int)fy, y1 = (y0 + 1 < H) ? y0 + 1 : (H > 1 ? H - 2 : 0);
        const double wy0 = 1.0 - (y - fy), wy1 = 1.0 - wy0;
        for (int j = 0; j < Wo; j++) {
            const double x = linspace_at(j, W, Wo), fx = floor(x);
            const int x0 = (int)fx, x1 = (x0 + 1 < W) ? x0 + 1 : (W > 1 ? W - 2 : 0);
            const double wx0 = 1.0 - (x - fx), wx1 = 1.0 - wx0;
            double acc, c;
            c = (double)in[(size_t)y0 * W + x0]; c = c * wy0; acc = c * wx0;
            c = (double)in[(size_t)y0 * W + x1]; c = c * wy0; acc = fma(c, wx1, acc);
            c = (double)in[(size_t)y1 * W + x0]; c = c * wy1; acc = fma(c, wx0, acc);
            c = (double)in[(size_t)y1 * W + x1]; c = c * wy1; acc = fma(c, wx1, acc);
            out[(size_t)i * Wo + j] = (float)acc;
        }
    }
}

static void pyramid_step(const float *in, int H, int W, int Ho, int Wo, float *out, int variant)
{
    size_t n = (size_t)H * W;
    float *t1 = (float *)malloc(sizeof(float) * (n ? n : 1)), *t2 = (float *)malloc(sizeof(float) * (n ? n : 1));
    if (variant == PYR_EXACT) {
        oflk_oracle_gaussian_filter(in, H, W, 2.0, t2);
        oflk_oracle_resample_linspace(t2, H, W, Ho, Wo, out);
    } else {
        double w[65];
        int radius = oflk_oracle_gaussian_kernel1d(2.0, w);
        blur_axis(in, t1, H, W, 0, w, radius, variant);
        blur_axis(t1, t2, H, W, 1, w, radius, variant);
        if (variant == PYR_F32) resample_fast(t2, H, W, Ho, Wo, out, 0);
        else resample_contracted(t2, H, W, Ho, Wo, out);
    }
    free(t1);
    free(t2);
}

/* ---- warp ----------------------------------------------------------------------------------------------- */
static void warp_variant(const float *img, const float *u, const float *v, int H, int W, float *out, int variant)
{
    if (variant == WARP_EXACT) {
        oflk_oracle_warp(img, u, v, H, W, out);
        return;
    }
    const double Hm1 = (double)(H - 1), Wm1 = (double)(W - 1);
    const double Hm2 = (double)(H > 1 ? H - 2 : 0), Wm2 = (double)(W > 1 ? W - 2 : 0);
    for (int gy = 0; gy < H; gy++) {
        for (int gx = 0; gx < W; gx++) {
            const size_t i = (size_t)gy * W + gx;
            float r;
            if (variant == WARP_LERP64) {
                /* the exact path's coordinate, range test and capped floor (lean_frac_at); then three fused lerps */
                const double y = (double)gy + (double)v[i], x = (double)gx + (double)u[i];
                if (!(y >= 0.0 && y <= Hm1 && x >= 0.0 && x <= Wm1)) {
                    out[i] = 0.0f;
                    continue;
                }
                const double fy = fmin(floor(y), Hm2), fx = fmin(floor(x), Wm2);
                const double ry = y - fy, rx = x - fx;
                const int y0 = (int)fy, x0 = (int)fx, y1 = H > 1 ? y0 + 1 : y0, x1 = W > 1 ? x0 + 1 : x0;
                r = lerp2_f64(img[(size_t)y0 * W + x0], img[(size_t)y0 * W + x1], img[(size_t)y1 * W + x0],
                              img[(size_t)y1 * W + x1], rx, ry);
            } else {
                /* fractions from the flow alone, in fp32: frac(gx + u) = u - floor(u) (exact in fp32 unless u is a tiny
                 * negative number); integer cell = gx + (int)floor(u) */
                const float flu = floorf(u[i]), flv = floorf(v[i]);
                float rx = u[i] - flu, ry = v[i] - flv;
                long x0 = (long)gx + (long)flu, y0 = (long)gy + (long)flv;
                if (rx >= 1.0f) { rx = 0.0f; x0 += 1; }   /* -1e-9 - (-1) rounds to 1 */
                if (ry >= 1.0f) { ry = 0.0f; y0 += 1; }
                /* inside: 0 <= x0 + rx <= W-1 */
                const int in_x = x0 >= 0 && (x0 < W - 1 || (x0 == W - 1 && rx == 0.0f));
                const int in_y = y0 >= 0 && (y0 < H - 1 || (y0 == H - 1 && ry == 0.0f));
                if (!(in_x && in_y) || !(fabsf(u[i]) < 1e9f) || !(fabsf(v[i]) < 1e9f)) {
                    out[i] = 0.0f;
                    continue;
                }
                if (x0 > W - 2 && W > 1) { x0 = W - 2; rx = 1.0f; }
                if (y0 > H - 2 && H > 1) { y0 = H - 2; ry = 1.0f; }
                const long y1 = H > 1 ? y0 + 1 : y0, x1 = W > 1 ? x0 + 1 : x0;
                const float a = img[(size_t)y0 * W + x0], b = img[(size_t)y0 * W + x1], c = img[(size_t)y1 * W + x0],
                            d = img[(size_t)y1 * W + x1];
                r = variant == WARP_F32 ? lerp2_f32(a, b, c, d, rx, ry) : lerp2_f64(a, b, c, d, (double)rx, (double)ry);
            }
            out[i] = r;
        }
    }
}

/* ---- window sums + solve -------------------------------------------------------------------------------- */
static inline void solve_variant(float Sxx, float Syy, float Sxy, float Sxt, float Syt, float *u, float *v, int variant)
{
    const float b0 = -Sxt, b1 = -Syt;
    float det, nu, nv;
    if (variant == SOLVE_FMA_DET) {
        det = fmaf(Sxx, Syy, -(Sxy * Sxy));
        nu = fmaf(Syy, b0, -(Sxy * b1));
        nv = fmaf(Sxx, b1, -(Sxy * b0));
    } else {
        const float m0 = Sxx * Syy, m1 = Sxy * Sxy;
        det = m0 - m1;
        const float n0 = Syy * b0, n1 = Sxy * b1, n2 = Sxx * b1, n3 = Sxy * b0;
        nu = n0 - n1;
        nv = n2 - n3;
    }
    *u = 0.0f;
    *v = 0.0f;
    if (fabsf(det) > 1e-4f) {
        if (variant == SOLVE_SHARED_RCP) {
            const float r = 1.0f / det;
            *u = nu * r;
            *v = nv * r;
        } else {
            *u = nu / det;
            *v = nv / det;
        }
    }
}

/* 5-sum of a[0..4] in the separable kernels' order */
static inline float sum5(float a0, float a1, float a2, float a3, float a4)
{
    return ((a0 + a1) + (a2 + a3)) + a4;
}

static void lk_variant(const float *prev, const float *curr, int H, int W, int win, float *u, float *v, int sums,
                       int solve)
{
    const size_t N = (size_t)H * W;
    float *Ix = (float *)malloc(sizeof(float) * 3 * (N ? N : 1)), *Iy = Ix + N, *It = Iy + N;
    oflk_oracle_compute_gradients(prev, curr, H, W, Ix, Iy, It);   /* Sobel stays the reference's */
    if (sums == SUMS_NUMPY && solve == SOLVE_EXACT) {
        oflk_oracle_from_gradients(Ix, Iy, It, H, W, win, u, v);
        free(Ix);
        return;
    }
    memset(u, 0, sizeof(float) * N);
    memset(v, 0, sizeof(float) * N);
    const int hw = win / 2, side = 2 * hw + 1;
    if (H <= 2 * hw || W <= 2 * hw) {
        free(Ix);
        return;
    }
    float *P = (float *)malloc(sizeof(float) * 5 * N), *T = (float *)malloc(sizeof(float) * 5 * N);
    float *S = (float *)malloc(sizeof(float) * 5 * N);
    for (size_t i = 0; i < N; i++) {
        P[i] = Ix[i] * Ix[i];
        P[N + i] = Iy[i] * Iy[i];
        P[2 * N + i] = Ix[i] * Iy[i];
        P[3 * N + i] = Ix[i] * It[i];
        P[4 * N + i] = Iy[i] * It[i];
    }
    for (int p = 0; p < 5; p++) {
        const float *a = P + p * N;
        float *t = T + p * N, *s = S + p * N;
        if (sums == SUMS_NUMPY) {
            float buf[2048];
            for (int y = hw; y < H - hw; y++)
                for (int x = hw; x < W - hw; x++) {
                    int k = 0;
                    for (int dy = -hw; dy <= hw; dy++)
                        for (int dx = -hw; dx <= hw; dx++) buf[k++] = a[(size_t)(y + dy) * W + x + dx];
                    /* np.sum's pairwise block (n <= 128) */
                    float r[8], res;
                    int n = side * side, i;
                    if (n < 8) {
                        res = 0.0f;
                        for (i = 0; i < n; i++) res += buf[i];
                    } else {
                        for (int j = 0; j < 8; j++) r[j] = buf[j];
                        for (i = 8; i < n - (n % 8); i += 8)
                            for (int j = 0; j < 8; j++) r[j] += buf[i + j];
                        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                        for (; i < n; i++) res += buf[i];
                    }
                    s[(size_t)y * W + x] = 0.0f + res;
                }
        } else if (sums == SUMS_SEPARABLE) {
            /* horizontal first, then vertical */
            for (int y = 0; y < H; y++)
                for (int x = hw; x < W - hw; x++) {
                    const float *q = a + (size_t)y * W + x;
                    float acc;
                    if (hw == 2) acc = sum5(q[-2], q[-1], q[0], q[1], q[2]);
                    else {
                        acc = q[-hw];
                        for (int d = -hw + 1; d <= hw; d++) acc += q[d];
                    }
                    t[(size_t)y * W + x] = acc;
                }
            for (int y = hw; y < H - hw; y++)
                for (int x = hw; x < W - hw; x++) {
                    const float *q = t + (size_t)y * W + x;
                    float acc;
                    if (hw == 2) acc = sum5(q[-2 * W], q[-W], q[0], q[W], q[2 * W]);
                    else {
                        acc = q[-(ptrdiff_t)hw * W];
                        for (int d = -hw + 1; d <= hw; d++) acc += q[(ptrdiff_t)d * W];
                    }
                    s[(size_t)y * W + x] = acc;
                }
        } else {
            /* vertical first (a wave walking down rows keeps a register ring), then horizontal (lane shifts) */
            for (int y = hw; y < H - hw; y++)
                for (int x = 0; x < W; x++) {
                    const float *q = a + (size_t)y * W + x;
                    float acc;
                    if (hw == 2) acc = sum5(q[-2 * W], q[-W], q[0], q[W], q[2 * W]);
                    else {
                        acc = q[-(ptrdiff_t)hw * W];
                        for (int d = -hw + 1; d <= hw; d++) acc += q[(ptrdiff_t)d * W];
                    }
                    t[(size_t)y * W + x] = acc;
                }
            for (int y = hw; y < H - hw; y++)
                for (int x = hw; x < W - hw; x++) {
                    const float *q = t + (size_t)y * W + x;
                    float acc;
                    if (hw == 2) {
                        /* two columns per lane: an even column adds whole lane pairs first, an odd one straddles them */
                        if ((x & 1) == 0) acc = ((q[-2] + q[-1]) + (q[0] + q[1])) + q[2];
                        else acc = (q[-2] + (q[-1] + q[0])) + (q[1] + q[2]);
                    } else {
                        acc = q[-hw];
                        for (int d = -hw + 1; d <= hw; d++) acc += q[d];
                    }
                    s[(size_t)y * W + x] = acc;
                }
        }
    }
    for (int y = hw; y < H - hw; y++)
        for (int x = hw; x < W - hw; x++) {
            const size_t i = (size_t)y * W + x;
            solve_variant(S[i], S[N + i], S[2 * N + i], S[3 * N + i], S[4 * N + i], &u[i], &v[i], solve);
        }
    free(P);
    free(T);
    free(S);
    free(Ix);
}

/* ---- the pyramidal pass with a variant per cell ------------------------------------------------------------
 * pyr_v[l]            variant of the pyramid step that PRODUCES level l (l < levels-1), both frames
 * up_v[l]             variant of the flow upsample INTO level l (l >= 1)
 * warp_v / sums_v / solve_v [l*iters + k]   variants of iteration k of level l
 * Returns 0 or -1.  residual_log / iters_run as oflk_oracle_pyramidal. */
OFLK_EXPORT int oflk_model_pyramidal(const float *prev, const float *curr, int H, int W, int levels, int win, int iters,
                                     const int *pyr_v, const int *up_v, const int *warp_v, const int *sums_v,
                                     const int *solve_v, float *u_out, float *v_out, float *residual_log, int *iters_run)
{
    if (levels < 1 || levels > 32 || H < 1 || W < 1) return -1;
    int dims[64];
    oflk_oracle_pyramid_dims(H, W, levels, 0.5, dims);
    float *pp[32], *pc[32];
    for (int l = 0; l < levels; l++) {
        size_t n = (size_t)dims[2 * l] * (size_t)dims[2 * l + 1];
        pp[l] = (float *)malloc(sizeof(float) * (n ? n : 1));
        pc[l] = (float *)malloc(sizeof(float) * (n ? n : 1));
    }
    memcpy(pp[levels - 1], prev, sizeof(float) * (size_t)H * W);
    memcpy(pc[levels - 1], curr, sizeof(float) * (size_t)H * W);
    for (int l = levels - 2; l >= 0; l--) {
        pyramid_step(pp[l + 1], dims[2 * l + 2], dims[2 * l + 3], dims[2 * l], dims[2 * l + 1], pp[l], pyr_v[l]);
        pyramid_step(pc[l + 1], dims[2 * l + 2], dims[2 * l + 3], dims[2 * l], dims[2 * l + 1], pc[l], pyr_v[l]);
    }
    size_t n0 = (size_t)dims[0] * (size_t)dims[1];
    float *fu = (float *)calloc(n0 ? n0 : 1, sizeof(float)), *fv = (float *)calloc(n0 ? n0 : 1, sizeof(float));
    for (int l = 0; l < levels; l++) {
        const int h = dims[2 * l], w = dims[2 * l + 1];
        const size_t n = (size_t)h * w;
        if (l > 0) {
            float *nu = (float *)malloc(sizeof(float) * (n ? n : 1)), *nv = (float *)malloc(sizeof(float) * (n ? n : 1));
            const int hc = dims[2 * l - 2], wc = dims[2 * l - 1];
            if (up_v[l] == UP_EXACT) {
                oflk_oracle_upsample_flow(fu, fv, hc, wc, h, w, nu, nv);
            } else {
                resample_fast(fu, hc, wc, h, w, nu, up_v[l] == UP_LERP64);
                resample_fast(fv, hc, wc, h, w, nv, up_v[l] == UP_LERP64);
                const float sx = (float)((double)w / (double)wc), sy = (float)((double)h / (double)hc);
                for (size_t i = 0; i < n; i++) {
                    nu[i] = nu[i] * sx;
                    nv[i] = nv[i] * sy;
                }
            }
            free(fu);
            free(fv);
            fu = nu;
            fv = nv;
        }
        float *warped = (float *)malloc(sizeof(float) * 3 * (n ? n : 1)), *du = warped + n, *dv = du + n;
        if (iters_run) iters_run[l] = 0;
        for (int k = 0; k < iters; k++) {
            const int c = l * iters + k;
            warp_variant(pc[l], fu, fv, h, w, warped, warp_v[c]);
            lk_variant(pp[l], warped, h, w, win, du, dv, sums_v[c], solve_v[c]);
            for (size_t i = 0; i < n; i++) {
                fu[i] = fu[i] + du[i];
                fv[i] = fv[i] + dv[i];
            }
            const float mu = oflk_oracle_mean_abs(du, n), mv = oflk_oracle_mean_abs(dv, n);
            if (residual_log) {
                residual_log[((size_t)l * iters + k) * 2 + 0] = mu;
                residual_log[((size_t)l * iters + k) * 2 + 1] = mv;
            }
            if (iters_run) iters_run[l] = k + 1;
            if (mu < 0.01f && mv < 0.01f) break;
        }
        free(warped);
    }
    memcpy(u_out, fu, sizeof(float) * (size_t)H * W);
    memcpy(v_out, fv, sizeof(float) * (size_t)H * W);
    free(fu);
    free(fv);
    for (int l = 0; l < levels; l++) {
        free(pp[l]);
        free(pc[l]);
    }
    return 0;
}
