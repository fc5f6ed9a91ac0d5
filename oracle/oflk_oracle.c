/*
 * oflk_oracle.c -- CPU restatement of the reference's dense Lucas-Kanade path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped path
 * (optical-flow-fpga_amd/) never links, imports or calls it.
 *
 * What it restates (all citations relative to /root/reference):
 *   python/lucas_kanade_core.py:15-45    compute_gradients
 *   python/lucas_kanade_core.py:73-135   lucas_kanade_from_gradients
 *   python/lucas_kanade_core.py:48-70    lucas_kanade_single_scale
 *   python/lucas_kanade_pyramidal.py:23-63    build_gaussian_pyramid
 *   python/lucas_kanade_pyramidal.py:66-97    warp_image
 *   python/lucas_kanade_pyramidal.py:100-138  upsample_flow
 *   python/lucas_kanade_pyramidal.py:141-228  lucas_kanade_pyramidal
 *
 * The reference delegates its arithmetic to NumPy / SciPy (pyproject.toml:34-40,
 * unpinned ranges; validated here against numpy 2.2.6 / scipy 1.15.3).  The
 * published algorithms of those call sites are restated op for op:
 *   scipy.signal.convolve2d(mode="same", boundary="symm")  -> sobel()
 *   np.sum of a fresh contiguous fp32 array (pairwise, 8 accumulators) -> np_pairwise_sum_f32()
 *   scipy.ndimage.gaussian_filter (correlate1d, fp64 line buffers, fp32 store per axis)
 *   scipy.ndimage.map_coordinates(order=1, mode="constant", cval=0)  -> bilinear_f64()
 *   np.linspace, np.mean
 *
 * Parity pin: tests/golden/ holds vectors produced by importing the reference in
 * the build container (tests/golden/make_golden.py); tests/test_oracle_golden.py
 * checks this file against them bit for bit, and against the 26 metric sets of
 * the reference's python/verification_baseline.json.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: every fp32 op is
 * individually rounded, exactly as NumPy scalar arithmetic does).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define OFLK_EXPORT __attribute__((visibility("default")))

static int g_threads = 1;

OFLK_EXPORT void oflk_oracle_set_threads(int n)
{
    g_threads = n < 1 ? 1 : n;
}

OFLK_EXPORT int oflk_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* NumPy pairwise summation (numpy/_core/src/umath/loops_utils.h.src,
 * @TYPE@_pairwise_sum): used by np.sum (lucas_kanade_core.py:115-119) and by
 * np.mean (lucas_kanade_pyramidal.py:213-214).                              */
/* ------------------------------------------------------------------------- */
static float np_pairwise_sum_f32(const float *a, size_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
    }
}

static double np_pairwise_sum_f64(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_f64(a, n2) + np_pairwise_sum_f64(a + n2, n - n2);
    }
}

/* np.sum(x) of a contiguous fp32 array, as the add-reduction really runs it:
 * the output starts at the identity 0 and the inner loop is handed the data in
 * pieces of the ufunc buffer size (np.getbufsize() = 8192 elements); each piece
 * is pairwise-summed and added to the running output.  (Measured against
 * numpy 2.2.6 for n up to 2,073,600; a 25-element window is one piece.)      */
#define OFLK_NP_BUFSIZE 8192

static float np_sum_contig_f32(const float *a, size_t n)
{
    float s = 0.0f;
    for (size_t i = 0; i < n; i += OFLK_NP_BUFSIZE) {
        size_t m = n - i < OFLK_NP_BUFSIZE ? n - i : OFLK_NP_BUFSIZE;
        s = s + np_pairwise_sum_f32(a + i, m);
    }
    return s;
}

OFLK_EXPORT float oflk_oracle_np_sum_f32(const float *a, size_t n)
{
    return np_sum_contig_f32(a, n);
}

/* np.mean(np.abs(d)) as lucas_kanade_pyramidal.py:213-214 evaluates it:
 * fp32 |d| array, fp32 np.sum, then float32(float64(sum) / float64(n))
 * (numpy/_core/_methods.py _mean: the count is an intp scalar, so the quotient
 * is formed in double and cast back).                                        */
OFLK_EXPORT float oflk_oracle_mean_abs(const float *d, size_t n)
{
    float *t = (float *)malloc(sizeof(float) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) t[i] = fabsf(d[i]);
    float s = np_sum_contig_f32(t, n);
    free(t);
    return (float)((double)s / (double)n);
}

/* ------------------------------------------------------------------------- */
/* compute_gradients  (lucas_kanade_core.py:15-45)                           */
/* ------------------------------------------------------------------------- */

/* scipy.signal.convolve2d "symm" boundary for a 3x3 kernel: one ring of
 * edge-repeating reflection (index -1 -> 0, N -> N-1).                      */
static inline int symm1(int i, int n)
{
    if (i < 0) return 0;
    if (i >= n) return n - 1;
    return i;
}

OFLK_EXPORT void oflk_oracle_compute_gradients(const float *prev, const float *curr, int H, int W,
                                               float *Ix, float *Iy, float *It)
{
    size_t N = (size_t)H * (size_t)W;
    float *avg = (float *)malloc(sizeof(float) * (N ? N : 1));
    /* lucas_kanade_core.py:36  frame_avg = (prev + curr) / 2.0   (fp32) */
    for (size_t i = 0; i < N; i++) avg[i] = (prev[i] + curr[i]) / 2.0f;

    /* lucas_kanade_core.py:32-33: kernels as written, fp32 */
    const float sx[3][3] = {{-0.125f, 0.0f, 0.125f}, {-0.25f, 0.0f, 0.25f}, {-0.125f, 0.0f, 0.125f}};
    const float sy[3][3] = {{-0.125f, -0.25f, -0.125f}, {0.0f, 0.0f, 0.0f}, {0.125f, 0.25f, 0.125f}};

#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            /* lucas_kanade_core.py:39-40: true convolution (kernel flipped),
             * accumulated row-major over the kernel, each mul and add rounded
             * to fp32 (scipy/signal/_firfilter.c, float path). */
            float gx = 0.0f, gy = 0.0f;
            for (int j = 0; j < 3; j++) {
                int yy = symm1(y + 1 - j, H);
                for (int k = 0; k < 3; k++) {
                    int xx = symm1(x + 1 - k, W);
                    float p = avg[(size_t)yy * W + xx];
                    float tx = p * sx[j][k];
                    gx = gx + tx;
                    float ty = p * sy[j][k];
                    gy = gy + ty;
                }
            }
            Ix[(size_t)y * W + x] = gx;
            Iy[(size_t)y * W + x] = gy;
        }
    }
    /* lucas_kanade_core.py:43  It = prev - curr */
    for (size_t i = 0; i < N; i++) It[i] = prev[i] - curr[i];
    free(avg);
}

/* ------------------------------------------------------------------------- */
/* lucas_kanade_from_gradients  (lucas_kanade_core.py:73-135)                */
/* ------------------------------------------------------------------------- */
OFLK_EXPORT void oflk_oracle_from_gradients(const float *Ix, const float *Iy, const float *It, int H,
                                            int W, int window_size, float *u, float *v)
{
    size_t N = (size_t)H * (size_t)W;
    memset(u, 0, sizeof(float) * N); /* :101-102 */
    memset(v, 0, sizeof(float) * N);
    int hw = window_size / 2; /* :104 */
    int side = 2 * hw + 1;
    int n = side * side;
    /* abs(det) > 1e-4 with det an np.float32 scalar: NumPy 2 compares in fp32
     * (float32(1e-4) = 9.99999975e-05); NumPy 1 compared in double.  No fp32
     * value lies between the two thresholds, so both give the same decision. */
    const float thr = 1e-4f;

#pragma omp parallel num_threads(g_threads)
    {
        float *pxx = (float *)malloc(sizeof(float) * 5 * (size_t)(n ? n : 1));
        float *pyy = pxx + n, *pxy = pyy + n, *pxt = pxy + n, *pyt = pxt + n;
#pragma omp for schedule(static)
        for (int y = hw; y < H - hw; y++) { /* :107 */
            for (int x = hw; x < W - hw; x++) { /* :108 */
                int t = 0;
                for (int dy = -hw; dy <= hw; dy++) {
                    const size_t row = (size_t)(y + dy) * W;
                    for (int dx = -hw; dx <= hw; dx++, t++) {
                        float gx = Ix[row + x + dx], gy = Iy[row + x + dx], gt = It[row + x + dx];
                        pxx[t] = gx * gx; /* :115-119: win_a * win_b is rounded to fp32 first */
                        pyy[t] = gy * gy;
                        pxy[t] = gx * gy;
                        pxt[t] = gx * gt;
                        pyt[t] = gy * gt;
                    }
                }
                float Sxx = 0.0f + np_pairwise_sum_f32(pxx, (size_t)n);
                float Syy = 0.0f + np_pairwise_sum_f32(pyy, (size_t)n);
                float Sxy = 0.0f + np_pairwise_sum_f32(pxy, (size_t)n);
                float Sxt = 0.0f + np_pairwise_sum_f32(pxt, (size_t)n);
                float Syt = 0.0f + np_pairwise_sum_f32(pyt, (size_t)n);
                float b0 = -Sxt, b1 = -Syt; /* :125 */
                float m0 = Sxx * Syy;       /* :128, each op rounded */
                float m1 = Sxy * Sxy;
                float det = m0 - m1;
                if (fabsf(det) > thr) { /* :131 */
                    float n0 = Syy * b0, n1 = Sxy * b1;
                    float n2 = Sxx * b1, n3 = Sxy * b0;
                    float nu = n0 - n1, nv = n2 - n3;
                    u[(size_t)y * W + x] = nu / det; /* :132 */
                    v[(size_t)y * W + x] = nv / det; /* :133 */
                }
            }
        }
        free(pxx);
    }
}

/* lucas_kanade_single_scale  (lucas_kanade_core.py:48-70) */
OFLK_EXPORT void oflk_oracle_single_scale(const float *prev, const float *curr, int H, int W,
                                          int window_size, float *u, float *v)
{
    size_t N = (size_t)H * (size_t)W;
    float *Ix = (float *)malloc(sizeof(float) * 3 * (N ? N : 1));
    float *Iy = Ix + N, *It = Iy + N;
    oflk_oracle_compute_gradients(prev, curr, H, W, Ix, Iy, It);
    oflk_oracle_from_gradients(Ix, Iy, It, H, W, window_size, u, v);
    free(Ix);
}

/* ------------------------------------------------------------------------- */
/* scipy.ndimage.gaussian_filter  (lucas_kanade_pyramidal.py:46-47)          */
/* ------------------------------------------------------------------------- */

/* scipy/ndimage/_filters.py _gaussian_kernel1d(sigma, 0, radius):
 *   x = arange(-r, r+1); phi = exp(-0.5 / sigma**2 * x**2); phi /= phi.sum()
 * radius = int(truncate * sigma + 0.5), truncate = 4.0.
 * Returns the radius; w[k] (k = 0..radius) holds the weight at distance k.
 * For the reference's only sigma (2.0 = 1/scale_factor) the table below is the
 * one SciPy produces here (hex floats; checked by tests/test_oracle_golden.py);
 * other sigmas go through libm exp(), which may differ from NumPy's exp in the
 * last ulp -- "parity unpinned" for scale_factor != 0.5.                     */
#define OFLK_MAX_RADIUS 64

static const double k_sigma2_w[9] = {
    0x1.98862a07ae7b4p-3, 0x1.68856f9ab1982p-3, 0x1.ef9093fc46e5ap-4, 0x1.0941b71ceef37p-4,
    0x1.ba4d4125ffd2ap-6, 0x1.1f30504e20207p-7, 0x1.227362b5fc92dp-9, 0x1.c98b8c5d0dda5p-12,
    0x1.18aad19e4159bp-14};

OFLK_EXPORT int oflk_oracle_gaussian_kernel1d(double sigma, double *w)
{
    int radius = (int)(4.0 * sigma + 0.5);
    if (radius > OFLK_MAX_RADIUS) radius = OFLK_MAX_RADIUS;
    if (sigma == 2.0) {
        for (int k = 0; k <= 8; k++) w[k] = k_sigma2_w[k];
        return 8;
    }
    double phi[2 * OFLK_MAX_RADIUS + 1];
    double sigma2 = sigma * sigma;
    double c = -0.5 / sigma2;
    for (int i = -radius; i <= radius; i++) phi[i + radius] = exp(c * (double)(i * i));
    double s = 0.0 + np_pairwise_sum_f64(phi, (size_t)(2 * radius + 1));
    for (int k = 0; k <= radius; k++) w[k] = phi[radius + k] / s;
    return radius;
}

/* scipy.ndimage "reflect" extension (d c b a | a b c d | d c b a), any distance */
static inline int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

/* One correlate1d pass of a symmetric kernel (scipy/ndimage/src/ni_filters.c
 * NI_Correlate1D, symmetric branch): double line buffer,
 *   tmp = line[c]*w0;  for k = radius..1: tmp += (line[c-k] + line[c+k]) * w[k]
 * result stored to the fp32 output array. `stride` selects the axis.          */
static void correlate1d_sym(const float *in, float *out, int nlines, int len, size_t line_stride,
                            size_t elem_stride, const double *w, int radius)
{
#pragma omp parallel num_threads(g_threads)
    {
        double *line = (double *)malloc(sizeof(double) * (size_t)(len + 2 * radius));
#pragma omp for schedule(static)
        for (int l = 0; l < nlines; l++) {
            const float *src = in + (size_t)l * line_stride;
            float *dst = out + (size_t)l * line_stride;
            for (int i = -radius; i < len + radius; i++)
                line[i + radius] = (double)src[(size_t)reflect_idx(i, len) * elem_stride];
            for (int c = 0; c < len; c++) {
                const double *p = line + c + radius;
                double tmp = p[0] * w[0];
                for (int k = radius; k >= 1; k--) tmp += (p[-k] + p[k]) * w[k];
                dst[(size_t)c * elem_stride] = (float)tmp;
            }
        }
        free(line);
    }
}

/* gaussian_filter with the caller's weights (w[k] at distance k, k = 0..radius): what SciPy computes once it has
 * its kernel.  The Python wrapper forms the kernel with NumPy exactly as scipy/ndimage/_filters.py
 * _gaussian_kernel1d does (np.exp, not libm's exp), which pins every sigma, not only the embedded sigma = 2. */
OFLK_EXPORT void oflk_oracle_gaussian_filter_w(const float *in, int H, int W, const double *w, int radius, float *out)
{
    size_t N = (size_t)H * (size_t)W;
    float *tmp = (float *)malloc(sizeof(float) * (N ? N : 1));
    correlate1d_sym(in, tmp, W, H, 1, (size_t)W, w, radius);
    correlate1d_sym(tmp, out, H, W, (size_t)W, 1, w, radius);
    free(tmp);
}

/* gaussian_filter(image_f32, sigma): axis 0 first, fp32 store, then axis 1. */
OFLK_EXPORT void oflk_oracle_gaussian_filter(const float *in, int H, int W, double sigma, float *out)
{
    double w[OFLK_MAX_RADIUS + 1];
    int radius = oflk_oracle_gaussian_kernel1d(sigma, w);
    size_t N = (size_t)H * (size_t)W;
    float *tmp = (float *)malloc(sizeof(float) * (N ? N : 1));
    /* axis 0: lines are columns */
    correlate1d_sym(in, tmp, W, H, 1, (size_t)W, w, radius);
    /* axis 1: lines are rows */
    correlate1d_sym(tmp, out, H, W, (size_t)W, 1, w, radius);
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* scipy.ndimage.map_coordinates(order=1, mode="constant", cval=0.0)         */
/* (scipy/ndimage/src/ni_interpolation.c NI_GeometricTransform)              */
/* ------------------------------------------------------------------------- */
static inline int mirror_tap(int idx, int len)
{
    /* taps that fall outside carry weight exactly 0 for order 1; SciPy still
     * reads a mirrored in-range element, so do the same (0 * finite = 0). */
    if (len <= 1) return 0;
    int s2 = 2 * len - 2;
    if (idx < 0) {
        idx = s2 * (-idx / s2) + idx;
        idx = idx <= 1 - len ? idx + s2 : -idx;
    } else if (idx >= len) {
        idx -= s2 * (idx / s2);
        if (idx >= len) idx = s2 - idx;
    }
    return idx;
}

static inline float bilinear_f64(const float *img, int H, int W, double y, double x)
{
    /* map_coordinate(): constant mode -> out of [0, len-1] means cval */
    if (y < 0.0 || y > (double)(H - 1) || x < 0.0 || x > (double)(W - 1)) return 0.0f;
    /* NaN coordinates: every comparison above is false; floor(NaN) is NaN and
     * the cast below is undefined in C; SciPy has the same hole.  The path
     * never produces NaN coordinates from finite inputs. */
    double fy = floor(y), fx = floor(x);
    int y0 = (int)fy, x0 = (int)fx;
    /* get_spline_interpolation_weights(order 1): w0 = 1 - frac, w1 = 1 - w0 */
    double ry = y - fy, rx = x - fx;
    double wy0 = 1.0 - ry, wx0 = 1.0 - rx;
    double wy1 = 1.0 - wy0, wx1 = 1.0 - wx0;
    int y1 = mirror_tap(y0 + 1, H), x1 = mirror_tap(x0 + 1, W);
    double t = 0.0, c;
    c = (double)img[(size_t)y0 * W + x0]; c *= wy0; c *= wx0; t += c;
    c = (double)img[(size_t)y0 * W + x1]; c *= wy0; c *= wx1; t += c;
    c = (double)img[(size_t)y1 * W + x0]; c *= wy1; c *= wx0; t += c;
    c = (double)img[(size_t)y1 * W + x1]; c *= wy1; c *= wx1; t += c;
    return (float)t;
}

/* np.linspace(0, S-1, T)[i]  (numpy/_core/function_base.py) */
static inline double linspace_at(int i, int S, int T)
{
    double delta = (double)(S - 1);
    if (T <= 1) return 0.0 * delta; /* div == 0: y = arange * delta + start */
    if (i == T - 1) return delta;   /* endpoint forced */
    double div = (double)(T - 1);
    double step = delta / div;
    if (step == 0.0) return ((double)i / div) * delta;
    return (double)i * step;
}

/* sample `in` on the linspace(0,H-1,Ho) x linspace(0,W-1,Wo) grid
 * (lucas_kanade_pyramidal.py:55-59 and :126-132) */
OFLK_EXPORT void oflk_oracle_resample_linspace(const float *in, int H, int W, int Ho, int Wo,
                                               float *out)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < Ho; i++) {
        double y = linspace_at(i, H, Ho);
        for (int j = 0; j < Wo; j++) {
            double x = linspace_at(j, W, Wo);
            out[(size_t)i * Wo + j] = bilinear_f64(in, H, W, y, x);
        }
    }
}

/* dims_out[2*l], dims_out[2*l+1] = H, W of pyramid level l, l = 0 coarsest
 * (lucas_kanade_pyramidal.py:51-52: int(height * scale_factor)) */
OFLK_EXPORT void oflk_oracle_pyramid_dims(int H, int W, int levels, double scale, int *dims_out)
{
    int h = H, w = W;
    for (int l = levels - 1; l >= 0; l--) {
        dims_out[2 * l] = h;
        dims_out[2 * l + 1] = w;
        h = (int)((double)h * scale);
        w = (int)((double)w * scale);
    }
}

/* build_gaussian_pyramid  (lucas_kanade_pyramidal.py:23-63).
 * out[l] must hold dims(l) floats; out[levels-1] receives a copy of `img`. */
OFLK_EXPORT void oflk_oracle_build_pyramid(const float *img, int H, int W, int levels, double scale,
                                           float **out)
{
    int dims[2 * 32];
    if (levels > 32) levels = 32;
    oflk_oracle_pyramid_dims(H, W, levels, scale, dims);
    memcpy(out[levels - 1], img, sizeof(float) * (size_t)H * (size_t)W); /* :40 */
    double sigma = 1.0 / scale;                                          /* :46 */
    for (int l = levels - 2; l >= 0; l--) {
        int h = dims[2 * (l + 1)], w = dims[2 * (l + 1) + 1];
        int ho = dims[2 * l], wo = dims[2 * l + 1];
        size_t n = (size_t)h * (size_t)w;
        float *sm = (float *)malloc(sizeof(float) * (n ? n : 1));
        oflk_oracle_gaussian_filter(out[l + 1], h, w, sigma, sm);    /* :47 */
        oflk_oracle_resample_linspace(sm, h, w, ho, wo, out[l]);     /* :55-59 */
        free(sm);
    }
}

/* build_gaussian_pyramid with the caller's Gaussian weights (see oflk_oracle_gaussian_filter_w) */
OFLK_EXPORT void oflk_oracle_build_pyramid_w(const float *img, int H, int W, int levels, double scale, const double *w,
                                             int radius, float **out)
{
    int dims[2 * 32];
    if (levels > 32) levels = 32;
    oflk_oracle_pyramid_dims(H, W, levels, scale, dims);
    memcpy(out[levels - 1], img, sizeof(float) * (size_t)H * (size_t)W);
    for (int l = levels - 2; l >= 0; l--) {
        int h = dims[2 * (l + 1)], wd = dims[2 * (l + 1) + 1];
        size_t n = (size_t)h * (size_t)wd;
        float *sm = (float *)malloc(sizeof(float) * (n ? n : 1));
        oflk_oracle_gaussian_filter_w(out[l + 1], h, wd, w, radius, sm);
        oflk_oracle_resample_linspace(sm, h, wd, dims[2 * l], dims[2 * l + 1], out[l]);
        free(sm);
    }
}

/* warp_image  (lucas_kanade_pyramidal.py:66-97): x + (double)u, y + (double)v */
OFLK_EXPORT void oflk_oracle_warp(const float *img, const float *u, const float *v, int H, int W,
                                  float *out)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            size_t i = (size_t)y * W + x;
            double xs = (double)x + (double)u[i]; /* :88  int64 + float32 -> float64 */
            double ys = (double)y + (double)v[i]; /* :89 */
            out[i] = bilinear_f64(img, H, W, ys, xs);
        }
    }
}

/* upsample_flow  (lucas_kanade_pyramidal.py:100-138) */
OFLK_EXPORT void oflk_oracle_upsample_flow(const float *u, const float *v, int Hc, int Wc, int Ht,
                                           int Wt, float *uo, float *vo)
{
    double scale_y = (double)Ht / (double)Hc; /* :122 */
    double scale_x = (double)Wt / (double)Wc; /* :123 */
    /* :135-136: fp32 array * Python float -> fp32 multiply by float32(scale) */
    float sx = (float)scale_x, sy = (float)scale_y;
    oflk_oracle_resample_linspace(u, Hc, Wc, Ht, Wt, uo);
    oflk_oracle_resample_linspace(v, Hc, Wc, Ht, Wt, vo);
    size_t N = (size_t)Ht * (size_t)Wt;
    for (size_t i = 0; i < N; i++) {
        uo[i] = uo[i] * sx;
        vo[i] = vo[i] * sy;
    }
}

/* lucas_kanade_pyramidal  (lucas_kanade_pyramidal.py:141-228).
 * residual_log[(l*iters + k)*2 + {0,1}] = mean|du|, mean|dv| of iteration k at
 * level l (only the first iters_run[l] entries of a level are written).
 * Returns 0, or -1 on bad arguments.                                         */
OFLK_EXPORT int oflk_oracle_pyramidal(const float *prev, const float *curr, int H, int W,
                                      int levels, int window_size, int iters, float *u_out,
                                      float *v_out, float *residual_log, int *iters_run)
{
    if (levels < 1 || levels > 32 || H < 1 || W < 1) return -1;
    int dims[64];
    const double scale = 0.5; /* build_gaussian_pyramid default, :24 */
    oflk_oracle_pyramid_dims(H, W, levels, scale, dims);
    float *pp[32], *pc[32];
    for (int l = 0; l < levels; l++) {
        size_t n = (size_t)dims[2 * l] * (size_t)dims[2 * l + 1];
        pp[l] = (float *)malloc(sizeof(float) * (n ? n : 1));
        pc[l] = (float *)malloc(sizeof(float) * (n ? n : 1));
    }
    oflk_oracle_build_pyramid(prev, H, W, levels, scale, pp); /* :173 */
    oflk_oracle_build_pyramid(curr, H, W, levels, scale, pc); /* :174 */

    size_t n0 = (size_t)dims[0] * (size_t)dims[1];
    float *fu = (float *)calloc(n0 ? n0 : 1, sizeof(float)); /* :182-184 */
    float *fv = (float *)calloc(n0 ? n0 : 1, sizeof(float));

    for (int l = 0; l < levels; l++) { /* :187 */
        int h = dims[2 * l], w = dims[2 * l + 1];
        size_t n = (size_t)h * (size_t)w;
        if (l > 0) { /* :195-197 */
            float *nu = (float *)malloc(sizeof(float) * (n ? n : 1));
            float *nv = (float *)malloc(sizeof(float) * (n ? n : 1));
            oflk_oracle_upsample_flow(fu, fv, dims[2 * (l - 1)], dims[2 * (l - 1) + 1], h, w, nu, nv);
            free(fu);
            free(fv);
            fu = nu;
            fv = nv;
        }
        float *warped = (float *)malloc(sizeof(float) * (n ? n : 1));
        float *du = (float *)malloc(sizeof(float) * (n ? n : 1));
        float *dv = (float *)malloc(sizeof(float) * (n ? n : 1));
        if (iters_run) iters_run[l] = 0;
        for (int k = 0; k < iters; k++) { /* :201 */
            oflk_oracle_warp(pc[l], fu, fv, h, w, warped);                      /* :203 */
            oflk_oracle_single_scale(pp[l], warped, h, w, window_size, du, dv); /* :206 */
            for (size_t i = 0; i < n; i++) { /* :209-210 */
                fu[i] = fu[i] + du[i];
                fv[i] = fv[i] + dv[i];
            }
            float mu = oflk_oracle_mean_abs(du, n); /* :213-214 */
            float mv = oflk_oracle_mean_abs(dv, n);
            if (residual_log) {
                residual_log[((size_t)l * iters + k) * 2 + 0] = mu;
                residual_log[((size_t)l * iters + k) * 2 + 1] = mv;
            }
            if (iters_run) iters_run[l] = k + 1;
            if (mu < 0.01f && mv < 0.01f) break; /* :221-223, fp32 vs float32(0.01) */
        }
        free(warped);
        free(du);
        free(dv);
    }
    size_t N = (size_t)H * (size_t)W;
    memcpy(u_out, fu, sizeof(float) * N);
    memcpy(v_out, fv, sizeof(float) * N);
    free(fu);
    free(fv);
    for (int l = 0; l < levels; l++) {
        free(pp[l]);
        free(pc[l]);
    }
    return 0;
}
