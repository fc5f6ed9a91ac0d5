// oflk.hip -- host side of liboflk.so: C ABI (include/oflk.h), plans, launch
// orchestration.  Device code lives in oflk_kernels.hpp.
//
// Build (see optical-flow-fpga_amd/csrc/Makefile):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ...
//
// There is deliberately no CPU path in this library: with no usable GPU every
// compute entry point returns OFLK_ERR_NO_DEVICE.
#include "oflk_kernels.hpp"
#include "oflk_stream.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/oflk.h"

#define OFLK_API extern "C" __attribute__((visibility("default")))

namespace {

using namespace oflk;

thread_local std::string t_err;
thread_local int t_resolved = 0;   // pairs the last host pyramidal call of this thread redid exactly

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(OFLK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

std::atomic<int> g_device{0};   // device of the host-pointer entry points (oflk_set_device)
std::atomic<int> g_host_arith{OFLK_ARITH_EXACT};   // arithmetic of the host-pointer entry points (oflk_set_host_arithmetic)
std::atomic<int> g_multi_workers{0};   // oflk_multi_rehearsal: queue workers of the *_multi entry points (0 = one per device)

int ensure_device(int dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(OFLK_ERR_NO_DEVICE,
                    "no usable HIP device (hipGetDeviceCount: %s, count %d); liboflk has no CPU path",
                    hipGetErrorString(e), n);
    }
    if (dev < 0 || dev >= n) return fail(OFLK_ERR_INVALID, "device %d out of range [0,%d)", dev, n);
    HIP_TRY(hipSetDevice(dev));
    return OFLK_OK;
}

// scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, int(4*sigma+0.5)).  For the
// reference's sigma (2.0 = 1/scale_factor, lucas_kanade_pyramidal.py:46) the
// table is SciPy's own output (NumPy's exp differs from libm's in the last ulp
// for some taps); other sigmas use libm and NumPy's pairwise normalisation order.
const double kSigma2[9] = {0x1.98862a07ae7b4p-3,  0x1.68856f9ab1982p-3,  0x1.ef9093fc46e5ap-4,
                           0x1.0941b71ceef37p-4,  0x1.ba4d4125ffd2ap-6,  0x1.1f30504e20207p-7,
                           0x1.227362b5fc92dp-9,  0x1.c98b8c5d0dda5p-12, 0x1.18aad19e4159bp-14};

double np_pairwise_f64(const double *a, size_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_f64(a, n2) + np_pairwise_f64(a + n2, n - n2);
}

int make_gauss(double sigma, GaussW *g)
{
    if (!(sigma > 0.0)) return fail(OFLK_ERR_INVALID, "sigma must be positive");
    int radius = (int)(4.0 * sigma + 0.5);
    if (radius > kMaxRadius)
        return fail(OFLK_ERR_UNSUPPORTED, "gaussian radius %d > %d (scale_factor too small)", radius,
                    kMaxRadius);
    g->radius = radius;
    if (sigma == 2.0) {
        for (int k = 0; k <= 8; k++) g->w[k] = kSigma2[k];
        return OFLK_OK;
    }
    double phi[2 * kMaxRadius + 1];
    double c = -0.5 / (sigma * sigma);
    for (int i = -radius; i <= radius; i++) phi[i + radius] = std::exp(c * (double)(i * i));
    double s = 0.0 + np_pairwise_f64(phi, (size_t)(2 * radius + 1));
    for (int k = 0; k <= radius; k++) g->w[k] = phi[radius + k] / s;
    return OFLK_OK;
}

Linspace make_linspace(int S, int T)
{
    Linspace l;
    l.T = T;
    l.last = (double)(S - 1);
    l.step = T > 1 ? (double)(S - 1) / (double)(T - 1) : 0.0;
    return l;
}

int level_dims(int H, int W, int levels, double scale, int *dims)
{
    if (H < 1 || W < 1) return fail(OFLK_ERR_INVALID, "H and W must be >= 1 (got %d x %d)", H, W);
    if (levels < 1 || levels > OFLK_MAX_LEVELS)
        return fail(OFLK_ERR_INVALID, "levels must be in [1,%d] (got %d)", OFLK_MAX_LEVELS, levels);
    if (!(scale > 0.0 && scale <= 1.0))
        return fail(OFLK_ERR_INVALID, "scale_factor must be in (0,1] (got %g)", scale);
    int h = H, w = W;
    for (int l = levels - 1; l >= 0; l--) {
        if (h < 1 || w < 1)
            return fail(OFLK_ERR_INVALID, "pyramid level %d of %dx%d would be empty", l, W, H);
        dims[2 * l] = h;
        dims[2 * l + 1] = w;
        h = (int)((double)h * scale);  // int(height * scale_factor), lucas_kanade_pyramidal.py:51
        w = (int)((double)w * scale);
    }
    return OFLK_OK;
}

// windows with a tiled kernel (3x3 ... 11x11); every other admissible size runs the generic one-thread-per-pixel kernel
inline bool tiled_window(int hw) { return hw >= 1 && hw <= 5; }

int window_hw(int window_size, int *hw)
{
    if (window_size < 1) return fail(OFLK_ERR_INVALID, "window_size must be >= 1");
    int h = window_size / 2;  // even sizes round down, lucas_kanade_core.py:104
    if (window_size > OFLK_MAX_WINDOW || pairwise_depth((2 * h + 1) * (2 * h + 1)) > kGenericDepth)
        return fail(OFLK_ERR_UNSUPPORTED, "window_size %d not built (windows of up to %d x %d)", window_size, OFLK_MAX_WINDOW,
                    OFLK_MAX_WINDOW);
    *hw = h;
    return OFLK_OK;
}

// ---- kernel classes for the per-kernel event timing --------------------------
enum KClass { KC_LK_SINGLE = 0, KC_LK_ITER, KC_LK_ITER_FINEST, KC_BLUR, KC_RESAMPLE,
              KC_UPSAMPLE, KC_EXPORT, KC_INIT, KC_PYR_FUSED, KC_LK_REDO, KC_COUNT };
const char *kClassNames[KC_COUNT] = {"lk_single", "lk_iter", "lk_iter_finest", "blur",
                                     "pyr_resample", "flow_upsample", "export_fixup", "call_init",
                                     "pyr_down_fused", "lk_single_redo"};

}  // namespace

struct oflk_plan {
    int device = 0;
    int B = 0, H = 0, W = 0, L = 0, win = 0, hw = 0, K = 0;
    int dims[2 * OFLK_MAX_LEVELS] = {0};
    size_t ws_bytes = 0;
    // workspace
    float *pyr[OFLK_MAX_LEVELS] = {nullptr};      // l < L-1: [2B][h][w] (prev then curr)
    float *tmpA = nullptr, *tmpB = nullptr;       // blur temporaries [2B][H][W]
    // per level one block: [slot 0..1][B][h][w] of interleaved float2 {u, v} (see LkArgs)
    float *flow[OFLK_MAX_LEVELS] = {nullptr};
    // per-call state, one allocation, zeroed by k_call_init at the start of every call:
    //   acc[B][L][K][kAccShards][kAccStride] (u64) | iters_run[B][L] (i32) | uncertain[B][L] (i32) | log[B][L][K][2] (f32)
    unsigned long long *state = nullptr;
    // uint8 plans only, and only when the fused pyramid kernel cannot take the frames (it always can for
    // scale 0.5 unless a level is tiny): float32 copies of the caller's frames, allocated on first need
    float *u8_stage[2] = {nullptr, nullptr};
    // scratch of oflk_plan_resolve_uncertain (one pair, unfused, planar), allocated on first use
    struct Exact {
        float *pyr[OFLK_MAX_LEVELS] = {nullptr};   // l < L-1: [2][h][w]
        float *u[OFLK_MAX_LEVELS] = {nullptr}, *v[OFLK_MAX_LEVELS] = {nullptr};
        float *warped = nullptr, *du = nullptr, *dv = nullptr, *f32[2] = {nullptr, nullptr}, *pieces = nullptr;
        bool ready = false;
    } exact;
    GaussW gauss;
    int arith = OFLK_ARITH_EXACT;   // oflk_plan_set_arithmetic
    int kernels = OFLK_KERNELS_AUTO;   // oflk_plan_set_kernels
    // single-scale 5x5 on integer-valued frames: the streaming kernel's list of doubtful tiles (LkArgs::redo)
    unsigned *redo = nullptr;
#ifdef OFLK_STAMPS
    unsigned *stamps = nullptr;      // diagnostic build: per-wave section cycle sums of the last finest-level launch
    size_t stamps_blocks = 0;
#endif
    // profiling
    bool prof = false;
    int prof_only = -1;  // >= 0: bracket only launches of this kernel class
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double acc_ms[KC_COUNT] = {0};
    long acc_n[KC_COUNT] = {0};

    size_t npix(int l) const { return (size_t)dims[2 * l] * (size_t)dims[2 * l + 1]; }
    float2 *fl(int l, int slot) const { return reinterpret_cast<float2 *>(flow[l]) + (size_t)slot * B * npix(l); }
    int Kc() const { return std::max(K, 1); }
    size_t n_acc() const { return (size_t)B * L * Kc() * kAccShards * kAccStride; }
    unsigned long long *acc() const { return state; }
    int *iters_run() const { return reinterpret_cast<int *>(state + n_acc()); }
    int *uncertain() const { return iters_run() + (size_t)B * L; }
    float *log() const { return reinterpret_cast<float *>(uncertain() + (size_t)B * L); }
    size_t n_log() const { return (size_t)B * L * Kc() * 2; }
    // 32-bit words of the whole state block (rounded up to a multiple of 4)
    size_t state_words() const { return ((2 * n_acc() + 2 * (size_t)B * L + n_log()) + 3) & ~(size_t)3; }
};

namespace {

struct Prof {
    oflk_plan *p;
    hipStream_t s;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(oflk_plan *p_, hipStream_t s_, int cls_) : p(p_), s(s_), cls(cls_)
    {
        if (!p || !p->prof) return;
        if (p->prof_only >= 0 && cls != p->prof_only) return;
        if (p->pool.empty()) {
            if (hipEventCreate(&a) != hipSuccess) { a = nullptr; return; }   // this launch goes untimed
            if (hipEventCreate(&b) != hipSuccess) {
                (void)hipEventDestroy(a);
                a = b = nullptr;
                return;
            }
        } else {
            a = p->pool.back().first;
            b = p->pool.back().second;
            p->pool.pop_back();
        }
        if (hipEventRecord(a, s) != hipSuccess) drop();
    }
    void drop()
    {
        (void)hipGetLastError();
        p->pool.push_back({a, b});
        a = b = nullptr;
    }
    ~Prof()
    {
        if (!a) return;
        if (hipEventRecord(b, s) != hipSuccess) {
            drop();
            return;
        }
        p->pending.push_back({a, b, cls});
    }
};

// smallest |d| total (fixed point) whose mean reads >= float32(0.01) for a level of `count`
// pixels: lk_mean_of is non-decreasing in the total, so "mean < 0.01" <=> total < threshold
unsigned long long conv_threshold(double count)
{
    unsigned long long lo = 0, hi = 1ull << 62;   // mean(lo) < 0.01 <= mean(hi)
    while (hi - lo > 1) {
        const unsigned long long mid = lo + (hi - lo) / 2;
        if (lk_mean_of(mid, count) < 0.01f) lo = mid;
        else hi = mid;
    }
    return hi;
}

// resident blocks of the redo pass after the streaming single-scale kernel: one per CU -- an empty list (the common case)
// then costs 7 us instead of the 22 us of a full round of 1024 blocks, a long one is walked four times slower
#ifndef OFLK_REDO_BLOCKS
#define OFLK_REDO_BLOCKS 256
#endif

// u8: a.prev / a.curr point at uint8 frames (finest level of a uint8 plan); never with MODE_GRADS
template <int MODE>
int launch_lk(oflk_plan *plan, hipStream_t s, int cls, int hw, const LkArgs &a_in, int B, bool u8 = false)
{
    LkArgs a = a_in;
    a.B = B;
    Prof pr(plan, s, cls);
    // 1-D grid.  Large launches chain vertically adjacent tiles in one block (they share 2R
    // staging rows, the expensive part of stage 1): tile rows are cut into segments of `cap`
    // rows, then ever shorter ones (each at most half of what is left), and every XCD runs its segments longest first so that the
    // drain of the grid is made of single tiles.  Small launches keep one tile per block.
    const int tiles_x = (a.W + k5TX - 1) / k5TX, tiles_y = (a.H + k5TY - 1) / k5TY;
    if (a.W <= 2 * hw || a.H <= 2 * hw) {
        // nothing has a full window: all-zero d (and k_lkw may assume H, W > 2*hw)
        const unsigned nb = (unsigned)(((size_t)a.H * (size_t)a.W + 255) / 256);
        hipLaunchKernelGGL((k_lk_degenerate<MODE>), dim3(nb, (unsigned)B), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        return OFLK_OK;
    }
    if (!tiled_window(hw)) {
        // 1x1, 13x13 and larger windows: one thread per output pixel, np.sum's pairwise order for any length
        if constexpr (MODE == MODE_ITER) {
            return fail(OFLK_ERR_UNSUPPORTED, "the fused iteration has no kernel for half window %d", hw);   // (plans of such windows run unfused)
        } else {
            const dim3 grid((a.W + 63) / 64, (a.H + 3) / 4, (unsigned)B);
            if constexpr (MODE == MODE_SINGLE) {
                if (u8) hipLaunchKernelGGL((k_lk_generic<MODE_SINGLE, unsigned char>), grid, dim3(256), 0, s, a, hw);
                else hipLaunchKernelGGL((k_lk_generic<MODE_SINGLE, float>), grid, dim3(256), 0, s, a, hw);
            } else {
                hipLaunchKernelGGL((k_lk_generic<MODE_GRADS, float>), grid, dim3(256), 0, s, a, hw);
            }
            HIP_TRY(hipGetLastError());
            return OFLK_OK;
        }
    }
    constexpr int cap_env = 8;   // tiles per chained block at most (12 / 16 were measured: no change)
    const long resident = 1024;  // 256 CUs x 4 blocks
    const long per_slot = (long)B * tiles_x * tiles_y / resident;
    int cap = MODE == MODE_GRADS ? 1 : (int)std::min<long>(std::max(1, cap_env), per_slot / 5);
    cap = std::max(cap, (tiles_y + kMaxSegs / 2 - 1) / (kMaxSegs / 2));  // keep the table short
    // seg_row holds tile rows as unsigned short
    if (MODE == MODE_GRADS || hw > 2 || cap_env <= 1 || per_slot < 10 || tiles_y > 65535) cap = 1;   // kLkChain
    unsigned nblocks;
    if (MODE == MODE_SINGLE && a.redo_pass) {
        // redo pass of the streaming kernel: resident blocks that walk the device-side list of flagged tiles
        a.nseg = 0;
        nblocks = (unsigned)std::min<long>((long)B * tiles_x * tiles_y, OFLK_REDO_BLOCKS);
    } else if (cap <= 1) {
        a.nseg = 0;
        nblocks = (unsigned)(tiles_x * tiles_y * B);
    } else {
        int row = 0, n = 0;
        while (row < tiles_y) {
            const int rest = tiles_y - row;
            int len = std::min(cap, std::max(1, rest / 2));   // halving tail: 8,8,8,8,6,3,2,1,1 for 45 rows
            if (n == kMaxSegs - 1) len = rest;  // the halving tail of a very tall frame can get here: last segment takes the rest
            a.seg_row[n++] = (unsigned short)row;
            row += len;
        }
        a.seg_row[n] = (unsigned short)tiles_y;
        a.nseg = n;
        const int strips = B * tiles_x;
        nblocks = 8u * (unsigned)((strips + 7) / 8) * (unsigned)n;
    }
    dim3 grid(nblocks);
#ifdef OFLK_STAMPS
    a.stamps = nullptr;
    if (plan && cls == KC_LK_ITER_FINEST) {
        if (plan->stamps_blocks < nblocks) {
            if (plan->stamps) (void)hipFree(plan->stamps);
            plan->stamps = nullptr;
            HIP_TRY(hipMalloc((void **)&plan->stamps, (size_t)nblocks * 520 * sizeof(unsigned)));
            plan->stamps_blocks = nblocks;
        }
        HIP_TRY(hipMemsetAsync(plan->stamps, 0, (size_t)nblocks * 520 * sizeof(unsigned), s));
        a.stamps = plan->stamps;
    }
#endif
    // the VEC instantiation moves 16 / 8 bytes per lane: every plane must start 16-byte aligned
    // (each row then does, W % 4 == 0); anything else takes the element-wise instantiation
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    auto al4 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 3u) == 0; };
    const bool frames_ok = u8 ? (al4(a.prev) && al4(a.curr)) : (al16(a.prev) && al16(a.curr));   // uint8: 4 pixels per dword
    const bool vec = (a.W & 3) == 0 && frames_ok && al16(a.aux) && al16(a.fl[0]) && al16(a.fl[1]) && al16(a.ou) && al16(a.ov);
#define OFLK_LAUNCH_LKW(HWV)                                                                          \
    do {                                                                                              \
        if constexpr (MODE != MODE_GRADS) {                                                           \
            if (u8) {                                                                                 \
                if (vec) hipLaunchKernelGGL((k_lkw<HWV, MODE, true, unsigned char>), grid, dim3(256), 0, s, a);  \
                else hipLaunchKernelGGL((k_lkw<HWV, MODE, false, unsigned char>), grid, dim3(256), 0, s, a);     \
                break;                                                                                \
            }                                                                                         \
        }                                                                                             \
        if (vec) hipLaunchKernelGGL((k_lkw<HWV, MODE, true>), grid, dim3(256), 0, s, a);                  \
        else hipLaunchKernelGGL((k_lkw<HWV, MODE, false>), grid, dim3(256), 0, s, a);                     \
    } while (0)
    if constexpr (MODE == MODE_SINGLE) {
        if (a.redo_pass && hw == 3) {   // the 7x7 kernel's list-walking instantiation
            if (u8) {
                if (vec) hipLaunchKernelGGL((k_lkw<3, MODE_SINGLE, true, unsigned char, true>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((k_lkw<3, MODE_SINGLE, false, unsigned char, true>), grid, dim3(256), 0, s, a);
            } else {
                if (vec) hipLaunchKernelGGL((k_lkw<3, MODE_SINGLE, true, float, true>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((k_lkw<3, MODE_SINGLE, false, float, true>), grid, dim3(256), 0, s, a);
            }
            HIP_TRY(hipGetLastError());
            return OFLK_OK;
        }
    }
    switch (hw) {
        case 1: OFLK_LAUNCH_LKW(1); break;
        case 2: OFLK_LAUNCH_LKW(2); break;
        case 3: OFLK_LAUNCH_LKW(3); break;
        case 4: OFLK_LAUNCH_LKW(4); break;
        case 5: OFLK_LAUNCH_LKW(5); break;
        default: return fail(OFLK_ERR_UNSUPPORTED, "half window %d not built", hw);
    }
#undef OFLK_LAUNCH_LKW
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

// k_lks (oflk_stream.hpp): the streaming form of the 5x5 kernel (single-scale: 7x7 too), for passes whose window sums need
// not be in NumPy's order (the tolerant mode's fine levels; single-scale on integer-valued frames, where any order is
// exact).  One wave per (strip of 120 output columns, segment of Hs rows); segments are sized so that the launch is a whole
// number of rounds of the chip's wave slots at the kernel's occupancy, ~128 rows each when there are rounds to spare (a
// segment pays 2 (hw + 1) extra rows), shorter ones when the launch would otherwise leave wave slots empty.
#ifndef OFLK_LKS_SEG_ROWS
#define OFLK_LKS_SEG_ROWS 128
#endif
template <int MODE>
int launch_lks(oflk_plan *plan, hipStream_t s, int cls, const LkArgs &a_in, int B, bool u8, int warp, int hw = 2)
{
    LkArgs a = a_in;
    a.B = B;
    Prof pr(plan, s, cls);
    const long strips = ((long)a.W + kLksOutW - 1) / kLksOutW * B;
    const long slots = 256 * 4 * (MODE == MODE_SINGLE ? 4 : OFLK_LKS_WAVES);   // wave slots of the chip at the kernel's occupancy (SINGLE: ~100 VGPRs)
    long segs = ((long)a.H + OFLK_LKS_SEG_ROWS - 1) / OFLK_LKS_SEG_ROWS;   // a segment pays 6 extra rows: ~128 rows each when the launch has rounds to spare
    const double rounds = (double)(strips * segs) / (double)slots;
    if (rounds > 0.75) segs = std::max<long>(1, (long)std::ceil(rounds - 0.25) * slots / strips);
    else segs = std::max(segs, std::min(slots / std::max<long>(strips, 1), std::max<long>(1, a.H / 40)));   // fill the one round, >= 40 rows each
    // a launch that cannot fill the slots even so (a single pair in the tolerant mode) trades rows per segment for parallel waves
    if (strips * segs < slots / 2) segs = std::max(segs, std::min(slots / 2 / std::max<long>(strips, 1), std::max<long>(1, a.H / 12)));
    segs = std::min<long>(segs, std::max<long>(1, a.H / 8));
    a.Hs = (int)(((long)a.H + segs - 1) / segs);
    a.segs = (a.H + a.Hs - 1) / a.Hs;
    const long nwave = strips * a.segs;
    dim3 grid((unsigned)((nwave + 3) / 4)), block(256);
    auto al = [](const void *q, unsigned m) { return (reinterpret_cast<uintptr_t>(q) & (m - 1)) == 0; };
    const bool frames_ok = u8 ? (al(a.prev, 2) && al(a.curr, 2)) : (al(a.prev, 8) && al(a.curr, 8));
    const bool vec = (a.W & 1) == 0 && a.W >= 2 && frames_ok && al(a.ou, 8) && al(a.ov, 8) &&
                     (MODE != MODE_ITER || (al(a.fl[0], 16) && al(a.fl[1], 16)));
#define OFLK_LAUNCH_LKS(WV)                                                                                   \
    do {                                                                                                      \
        if (u8) {                                                                                             \
            if (vec) hipLaunchKernelGGL((k_lks<MODE, true, WV, unsigned char>), grid, block, 0, s, a);        \
            else hipLaunchKernelGGL((k_lks<MODE, false, WV, unsigned char>), grid, block, 0, s, a);           \
        } else {                                                                                              \
            if (vec) hipLaunchKernelGGL((k_lks<MODE, true, WV, float>), grid, block, 0, s, a);                \
            else hipLaunchKernelGGL((k_lks<MODE, false, WV, float>), grid, block, 0, s, a);                   \
        }                                                                                                     \
    } while (0)
    if constexpr (MODE == MODE_ITER) {
        if (a.up_src != nullptr) {
            // first iteration of a level with the flow upsampling fused in (tolerant mode)
            if (u8) {
                if (vec) hipLaunchKernelGGL((k_lks<MODE_ITER, true, WARP_LERP64, unsigned char, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((k_lks<MODE_ITER, false, WARP_LERP64, unsigned char, true>), grid, block, 0, s, a);
            } else {
                if (vec) hipLaunchKernelGGL((k_lks<MODE_ITER, true, WARP_LERP64, float, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((k_lks<MODE_ITER, false, WARP_LERP64, float, true>), grid, block, 0, s, a);
            }
        } else if (warp == WARP_LERP64) OFLK_LAUNCH_LKS(WARP_LERP64);
        else OFLK_LAUNCH_LKS(WARP_SCIPY);
    } else if (hw == 3) {
        if (u8) {
            if (vec) hipLaunchKernelGGL((k_lks<MODE_SINGLE, true, WARP_SCIPY, unsigned char, false, 3>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_lks<MODE_SINGLE, false, WARP_SCIPY, unsigned char, false, 3>), grid, block, 0, s, a);
        } else {
            if (vec) hipLaunchKernelGGL((k_lks<MODE_SINGLE, true, WARP_SCIPY, float, false, 3>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_lks<MODE_SINGLE, false, WARP_SCIPY, float, false, 3>), grid, block, 0, s, a);
        }
    } else {
        OFLK_LAUNCH_LKS(WARP_SCIPY);
    }
#undef OFLK_LAUNCH_LKS
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

inline dim3 grid2d(int W, int H, int n) { return dim3((W + 63) / 64, (H + 3) / 4, n); }
// k_resample: 4 outputs per thread along x
inline dim3 grid_resample(int W, int H, int n) { return dim3((W + 255) / 256, (H + 3) / 4, n); }

// does every 32 x 16 coarse tile's source span fit the fused kernel's LDS tile?
// (exactly the index arithmetic of k_pyr_down, evaluated for each tile row / column)
bool pyr_fused_fits(int h, int w, int ho, int wo, const GaussW &g)
{
    if (g.radius != 8) return false;
    auto span_ok = [](int S, int T, int tile, int cap) {
        Linspace l = make_linspace(S, T);
        auto at = [&](int i) { return T <= 1 ? 0.0 : (i == T - 1 ? l.last : (double)i * l.step); };
        for (int t0 = 0; t0 < T; t0 += tile) {
            int last = std::min(t0 + tile, T) - 1;
            int lo = (int)std::floor(at(t0));
            if (t0 + tile >= T) lo = std::min(lo, std::max(S - 2, 0));
            int hi = std::min((int)std::floor(at(last)) + 1, S - 1);
            if (hi - lo + 1 > cap) return false;
        }
        return true;
    };
    return span_ok(h, ho, kPTH, kPBH) && span_ok(w, wo, kPTW, kPBW);
}

// does every 256 x 4 output block of k_upsample find its coarse source span inside the
// staged LDS tile?  (same index arithmetic as the kernel)
bool upsample_fits(int hc, int wc, int ht, int wt)
{
    auto span_ok = [](int S, int T, int tile, int cap) {
        Linspace l = make_linspace(S, T);
        auto at = [&](int i) { return T <= 1 ? 0.0 : (i == T - 1 ? l.last : (double)i * l.step); };
        for (int t0 = 0; t0 < T; t0 += tile) {
            int last = std::min(t0 + tile, T) - 1;
            int lo = (int)std::floor(at(t0));
            if (t0 + tile >= T) lo = std::min(lo, std::max(S - 2, 0));
            int hi = std::min((int)std::floor(at(last)) + 1, S - 1);
            if (hi - lo + 1 > cap) return false;
        }
        return true;
    };
    return span_ok(hc, ht, kUTH, kUSH) && span_ok(wc, wt, kUTW, kUSW);
}

// upsample_flow of `nimg` flow fields (both planes); r is fully populated by the caller
int launch_upsample(oflk_plan *plan, hipStream_t s, const ResampleArgs &r_in, int nimg)
{
    ResampleArgs r = r_in;
    // 16-byte stores want Wo % 4 == 0 and 16-byte aligned output planes (hipMalloc / torch give that)
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    r.vec_store = (r.Wo & 3) == 0 && al16(r.out[0]) && (r.interleaved || al16(r.out[1]));
    Prof pr(plan, s, KC_UPSAMPLE);
    if (upsample_fits(r.H, r.W, r.Ho, r.Wo)) {
        dim3 grid((r.Wo + kUTW - 1) / kUTW, (r.Ho + kUTH - 1) / kUTH, nimg);
        hipLaunchKernelGGL(k_upsample, grid, dim3(256), 0, s, r);
    } else {
        hipLaunchKernelGGL(k_resample<2>, grid_resample(r.Wo, r.Ho, nimg), dim3(256), 0, s, r);
    }
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

// what the first pyramid launch of a pyramidal call carries besides its own work
struct PyrExtra {
    const float *in2 = nullptr;   // images nsplit .. nimg-1 (the second caller buffer)
    int nsplit = 0;
    unsigned *zero_words = nullptr;   // per-call state to clear
    size_t n_zero_words = 0;
    float *zero_u = nullptr, *zero_v = nullptr;   // coarsest-level flow planes to clear
    size_t n_zero_flow = 0;
    bool u8 = false;                  // the input images are uint8 (the caller's frames of a uint8 plan)
};

int launch_call_init(oflk_plan *plan, hipStream_t s, const PyrExtra &x)
{
    Prof pr(plan, s, KC_INIT);
    hipLaunchKernelGGL(k_call_init, dim3(256), dim3(256), 0, s, x.zero_words, x.n_zero_words, x.zero_u, x.zero_v,
                       x.n_zero_flow);
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

// gaussian blur + linspace resample of `nimg` images: in [nimg][h][w] -> out [nimg][ho][wo]
int launch_pyr_down(oflk_plan *plan, const GaussW &gauss, hipStream_t s, const float *in, float *out,
                    float *tmpA, float *tmpB, int nimg, int h, int w, int ho, int wo,
                    const PyrExtra *extra = nullptr)
{
    if (pyr_fused_fits(h, w, ho, wo, gauss)) {
        PyrArgs a{};
        a.in = in;
        a.in2 = in;
        a.nsplit = nimg;
        if (extra) {
            if (extra->in2) {
                a.in2 = extra->in2;
                a.nsplit = extra->nsplit;
            }
            a.zero_words = extra->zero_words;
            a.n_zero_words = extra->n_zero_words;
            a.zero_u = extra->zero_u;
            a.zero_v = extra->zero_v;
            a.n_zero_flow = extra->n_zero_flow;
        }
        a.out = out;
        a.H = h; a.W = w; a.Ho = ho; a.Wo = wo;
        a.ly = make_linspace(h, ho);
        a.lx = make_linspace(w, wo);
        for (int k = 0; k <= 8; k++) a.w[k] = gauss.w[k];
        dim3 grid((wo + kPTW - 1) / kPTW, (ho + kPTH - 1) / kPTH, nimg);
        Prof pr(plan, s, KC_PYR_FUSED);
        const bool fma = plan && plan->arith != OFLK_ARITH_EXACT;   // opt-in (contracted / tolerant); never the default
        if (extra && extra->u8) {
            if (fma) hipLaunchKernelGGL((k_pyr_down<unsigned char, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_pyr_down<unsigned char, false>), grid, dim3(256), 0, s, a);
        } else {
            if (fma) hipLaunchKernelGGL((k_pyr_down<float, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_pyr_down<float, false>), grid, dim3(256), 0, s, a);
        }
        HIP_TRY(hipGetLastError());
        return OFLK_OK;
    }
    if (extra) {
        // unfused path: the extras become launches of their own
        if (extra->zero_words) {
            int rc = launch_call_init(plan, s, *extra);
            if (rc) return rc;
        }
        if (extra->in2) {
            int rc = launch_pyr_down(plan, gauss, s, in, out, tmpA, tmpB, extra->nsplit, h, w, ho, wo);
            if (rc) return rc;
            return launch_pyr_down(plan, gauss, s, extra->in2, out + (size_t)extra->nsplit * ho * wo, tmpA, tmpB,
                                   nimg - extra->nsplit, h, w, ho, wo);
        }
    }
    {
        Prof pr(plan, s, KC_BLUR);
        const bool fma = plan && plan->arith != OFLK_ARITH_EXACT;   // the unfused chain keeps the plan's arithmetic
        if (fma) hipLaunchKernelGGL((k_blur<0, true>), grid2d(w, h, nimg), dim3(256), 0, s, in, tmpA, h, w, gauss);
        else hipLaunchKernelGGL((k_blur<0, false>), grid2d(w, h, nimg), dim3(256), 0, s, in, tmpA, h, w, gauss);
        HIP_TRY(hipGetLastError());
        if (fma) hipLaunchKernelGGL((k_blur<1, true>), grid2d(w, h, nimg), dim3(256), 0, s, (const float *)tmpA, tmpB, h, w, gauss);
        else hipLaunchKernelGGL((k_blur<1, false>), grid2d(w, h, nimg), dim3(256), 0, s, (const float *)tmpA, tmpB, h, w, gauss);
        HIP_TRY(hipGetLastError());
    }
    ResampleArgs r{};
    r.in[0] = tmpB;
    r.out[0] = out;
    r.acc = nullptr;
    r.in_sel_stride = 0;
    r.H = h; r.W = w; r.Ho = ho; r.Wo = wo;
    r.ly = make_linspace(h, ho);
    r.lx = make_linspace(w, wo);
    r.nplanes = 1;
    r.apply_scale = 0;
    r.vec_store = (wo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    {
        Prof pr(plan, s, KC_RESAMPLE);
        if (plan && plan->arith != OFLK_ARITH_EXACT) hipLaunchKernelGGL((k_resample<1, true>), grid_resample(wo, ho, nimg), dim3(256), 0, s, r);
        else hipLaunchKernelGGL((k_resample<1, false>), grid_resample(wo, ho, nimg), dim3(256), 0, s, r);
    }
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

template <typename T>
int dmalloc(T **p, size_t n, size_t *total)
{
    size_t bytes = std::max<size_t>(n * sizeof(T), 256);
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(OFLK_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    *total += bytes;
    return OFLK_OK;
}

void plan_free(oflk_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (int l = 0; l < OFLK_MAX_LEVELS; l++) {
        if (p->pyr[l]) (void)hipFree(p->pyr[l]);
        if (p->flow[l]) (void)hipFree(p->flow[l]);
    }
    if (p->tmpA) (void)hipFree(p->tmpA);
    if (p->tmpB) (void)hipFree(p->tmpB);
    if (p->state) (void)hipFree(p->state);
    if (p->redo) (void)hipFree(p->redo);
    for (auto &q : p->u8_stage)
        if (q) (void)hipFree(q);
    for (int l = 0; l < OFLK_MAX_LEVELS; l++)
        for (float *q : {p->exact.pyr[l], p->exact.u[l], p->exact.v[l]})
            if (q) (void)hipFree(q);
    for (float *q : {p->exact.warped, p->exact.du, p->exact.dv, p->exact.f32[0], p->exact.f32[1], p->exact.pieces})
        if (q) (void)hipFree(q);
#ifdef OFLK_STAMPS
    if (p->stamps) (void)hipFree(p->stamps);
#endif
    for (auto &e : p->pending) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto &e : p->pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete p;
}

}  // namespace

// =============================================================================
// library
// =============================================================================
OFLK_API const char *oflk_version(void) { return "oflk 0.4.0 (gfx950)"; }

OFLK_API int oflk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

OFLK_API const char *oflk_last_error(void) { return t_err.c_str(); }

OFLK_API int oflk_set_device(int device)
{
    int rc = ensure_device(device);
    if (rc == OFLK_OK) g_device = device;
    return rc;
}

OFLK_API int oflk_pyramid_level_dims(int H, int W, int levels, double scale_factor, int *dims_out)
{
    if (!dims_out) return fail(OFLK_ERR_INVALID, "dims_out is NULL");
    return level_dims(H, W, levels, scale_factor, dims_out);
}

// =============================================================================
// plan API
// =============================================================================
OFLK_API int oflk_plan_create(oflk_plan **out, int device, int B, int H, int W, int levels,
                              int window_size, int iters)
{
    if (!out) return fail(OFLK_ERR_INVALID, "plan pointer is NULL");
    *out = nullptr;
    if (B < 1) return fail(OFLK_ERR_INVALID, "B must be >= 1");
    if (iters < 0) return fail(OFLK_ERR_INVALID, "iters must be >= 0");
    if ((size_t)H * (size_t)W >= ((size_t)1 << 29) || H >= (1 << 24) || W >= (1 << 24))
        return fail(OFLK_ERR_UNSUPPORTED, "frames of 2^29 pixels or more are not supported");  // 32-bit byte offsets into float2 planes
    int hw = 0;
    int rc = window_hw(window_size, &hw);
    if (rc) return rc;
    int dims[2 * OFLK_MAX_LEVELS];
    rc = level_dims(H, W, levels, 0.5, dims);
    if (rc) return rc;
    rc = ensure_device(device);
    if (rc) return rc;

    oflk_plan *p = new oflk_plan();
    p->device = device;
    p->B = B; p->H = H; p->W = W; p->L = levels; p->win = window_size; p->hw = hw; p->K = iters;
    std::memcpy(p->dims, dims, sizeof(int) * 2 * levels);
    rc = make_gauss(2.0, &p->gauss);  // sigma = 1/scale_factor, scale_factor = 0.5 (:24, :46)
    const size_t N = (size_t)H * W;
    if (!rc && levels > 1) {
        rc = dmalloc(&p->tmpA, 2 * (size_t)B * N, &p->ws_bytes);
        if (!rc) rc = dmalloc(&p->tmpB, 2 * (size_t)B * N, &p->ws_bytes);
    }
    for (int l = 0; l < levels && !rc; l++) {
        size_t n = (size_t)dims[2 * l] * dims[2 * l + 1];
        if (l < levels - 1) rc = dmalloc(&p->pyr[l], 2 * (size_t)B * n, &p->ws_bytes);
        if (!rc && (iters > 0 || levels > 1)) rc = dmalloc(&p->flow[l], (size_t)2 * 2 * B * n, &p->ws_bytes);   // two interleaved slots
    }
    if (!rc) rc = dmalloc(&p->state, p->state_words() / 2, &p->ws_bytes);
    if (!rc && (p->hw == 2 || p->hw == 3)) {
        // redo list of the single-scale streaming kernel (LkArgs::redo): all zero between calls
        const size_t n = 2 + 2 * (size_t)B * ((W + k5TX - 1) / k5TX) * ((H + k5TY - 1) / k5TY);
        rc = dmalloc(&p->redo, n, &p->ws_bytes);
        if (!rc && hipMemset(p->redo, 0, std::max<size_t>(n * sizeof(unsigned), 256)) != hipSuccess) rc = fail(OFLK_ERR_HIP, "hipMemset of the redo list failed");
    }
    if (rc) {
        std::string keep = t_err;
        plan_free(p);
        t_err = keep;
        return rc;
    }
    *out = p;
    return OFLK_OK;
}

OFLK_API int oflk_plan_destroy(oflk_plan *plan)
{
    plan_free(plan);
    return OFLK_OK;
}

OFLK_API size_t oflk_plan_workspace_bytes(const oflk_plan *plan) { return plan ? plan->ws_bytes : 0; }

namespace {
int plan_single_scale(oflk_plan *p, const void *d_prev, const void *d_curr, bool u8, float *d_u, float *d_v,
                      hipStream_t s)
{
    if (!p || !d_prev || !d_curr || !d_u || !d_v) return fail(OFLK_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    LkArgs a{};
    a.prev = static_cast<const float *>(d_prev);   // element type is the kernel's PIX (launch_lk, u8)
    a.curr = static_cast<const float *>(d_curr);
    a.ou = d_u; a.ov = d_v;
    a.H = p->H; a.W = p->W;
    // The streaming kernel walks rows one after the other inside a wave: a launch too small to fill the chip's wave slots with
    // segments of ~40 rows is latency-bound there (one 640x480 pair: 37 us against the tile kernel's 6), so small launches
    // keep the tile kernel -- both are exact, the choice is speed only.
    const long stream_waves = ((long)p->W + kLksOutW - 1) / kLksOutW * p->B * std::max(1, p->H / 40);
    if ((p->hw == 2 || p->hw == 3) && p->H > 2 * p->hw && p->W > 2 * p->hw &&
        ((p->kernels == OFLK_KERNELS_AUTO && stream_waves >= 2048) || p->kernels == OFLK_KERNELS_STREAM)) {
        // 5x5 window: the streaming kernel, whose order-free sums are NumPy's wherever the frames are integers in [0, 255]
        // and a window's Sxx, Syy stay below 2^16 (proof at kLksExactBound); it flags the tiles where that is in doubt and
        // the tile kernel redoes exactly those in NumPy's order.  Results are the reference's either way.
        a.redo = p->redo;   // allocated and zeroed with the plan
        int rc = launch_lks<MODE_SINGLE>(p, s, KC_LK_SINGLE, a, p->B, u8, WARP_SCIPY, p->hw);
        if (rc) return rc;
        a.redo_pass = 1;
        return launch_lk<MODE_SINGLE>(p, s, KC_LK_REDO, p->hw, a, p->B, u8);
    }
    return launch_lk<MODE_SINGLE>(p, s, KC_LK_SINGLE, p->hw, a, p->B, u8);
}
int plan_pyramidal(oflk_plan *p, const void *d_prev, const void *d_curr, bool u8, float *d_u, float *d_v, hipStream_t s);
int resolve_pair(oflk_plan *p, int b, const void *d_prev_in, const void *d_curr_in, bool u8, float *d_u, float *d_v, hipStream_t s);
}  // namespace

OFLK_API int oflk_plan_single_scale(oflk_plan *p, const float *d_prev, const float *d_curr,
                                    float *d_u, float *d_v, void *stream)
{
    return plan_single_scale(p, d_prev, d_curr, false, d_u, d_v, (hipStream_t)stream);
}

// BASELINE config 5: fp16 gradients / accumulators (k_lk16d).  pixel_max bounds the frame values.
OFLK_API int oflk_plan_single_scale_fp16(oflk_plan *p, const float *d_prev, const float *d_curr, float *d_u, float *d_v,
                                         float pixel_max, void *stream)
{
    if (!p || !d_prev || !d_curr || !d_u || !d_v) return fail(OFLK_ERR_INVALID, "NULL argument");
    if (!(pixel_max > 0.0f) || !std::isfinite(pixel_max)) return fail(OFLK_ERR_INVALID, "pixel_max must be positive and finite");
    HIP_TRY(hipSetDevice(p->device));
    const int hw = p->hw, taps = (2 * hw + 1) * (2 * hw + 1);
    // s_g = 2^-k, the largest power of two (<= 1) with taps * (pixel_max / 2 * s_g)^2 <= 60000
    int k = 0;
    while ((double)taps * std::pow(0.5 * (double)pixel_max * std::ldexp(1.0, -k), 2.0) > 60000.0) k++;
    // one wave per (strip of 128 - 4 ceil(R/2) output columns, segment of Hs rows): two columns per lane (k_lk16d;
    // 7x7: 77 VGPRs = 6 waves per SIMD, halo 6 %).  Segments are sized so that the launch is a whole number of
    // rounds of the chip's wave slots at the kernel's occupancy, with ~64 rows each (a segment pays 2R extra rows
    // of loads).
    Lk16sArgs g{};
    g.prev = d_prev; g.curr = d_curr; g.u = d_u; g.v = d_v;
    g.H = p->H; g.W = p->W; g.B = p->B;
    g.s_g = (float)std::ldexp(1.0, -k);
    g.s_t = 0.5f * g.s_g;
    g.det_thr = (float)(1e-4 * std::ldexp(1.0, -4 * k));
    hipStream_t s = (hipStream_t)stream;
    Prof pr(p, s, KC_LK_SINGLE);
    const int outw = 2 * (64 - 2 * ((hw + 2) / 2));
    const long strips = ((long)g.W + outw - 1) / outw * g.B;
    const long slots = hw <= 2 ? 8192 : hw == 3 ? 6144 : hw == 4 ? 5120 : 4096;   // wave slots of the chip at 8 / 8 / 6 / 5 / 4 waves per SIMD
    long segs = ((long)g.H + 63) / 64;
    const double rounds = (double)(strips * segs) / (double)slots;
    if (rounds > 0.75) segs = std::max<long>(1, (long)std::ceil(rounds - 0.25) * slots / strips);
    else segs = std::max(segs, std::min(slots / std::max<long>(strips, 1), std::max<long>(1, g.H / 40)));   // fill the one round, >= 40 rows each
    segs = std::min<long>(segs, std::max<long>(1, g.H / 8));
    g.Hs = (int)(((long)g.H + segs - 1) / segs);
    g.segs = (g.H + g.Hs - 1) / g.Hs;
    const long nwave = strips * g.segs;
    dim3 sgrid((unsigned)((nwave + 3) / 4)), sblock(256);
    auto al8 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 7u) == 0; };
    const bool vec8 = (g.W & 1) == 0 && g.W >= 2 && al8(d_prev) && al8(d_curr) && al8(d_u) && al8(d_v);   // 8-byte column pairs
#define OFLK_LAUNCH_LK16D(HWV)                                                                 \
    do {                                                                                       \
        if (vec8) hipLaunchKernelGGL((k_lk16d<HWV, true>), sgrid, sblock, 0, s, g);            \
        else hipLaunchKernelGGL((k_lk16d<HWV, false>), sgrid, sblock, 0, s, g);                \
    } while (0)
    switch (hw) {
        case 1: OFLK_LAUNCH_LK16D(1); break;
        case 2: OFLK_LAUNCH_LK16D(2); break;
        case 3: OFLK_LAUNCH_LK16D(3); break;
        case 4: OFLK_LAUNCH_LK16D(4); break;
        case 5: OFLK_LAUNCH_LK16D(5); break;
        default: return fail(OFLK_ERR_UNSUPPORTED, "half window %d not built", hw);
    }
#undef OFLK_LAUNCH_LK16D
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

OFLK_API int oflk_plan_single_scale_u8(oflk_plan *p, const unsigned char *d_prev, const unsigned char *d_curr,
                                       float *d_u, float *d_v, void *stream)
{
    return plan_single_scale(p, d_prev, d_curr, true, d_u, d_v, (hipStream_t)stream);
}

OFLK_API int oflk_plan_pyramidal(oflk_plan *p, const float *d_prev, const float *d_curr, float *d_u,
                                 float *d_v, void *stream)
{
    return plan_pyramidal(p, d_prev, d_curr, false, d_u, d_v, (hipStream_t)stream);
}

OFLK_API int oflk_plan_pyramidal_u8(oflk_plan *p, const unsigned char *d_prev, const unsigned char *d_curr,
                                    float *d_u, float *d_v, void *stream)
{
    return plan_pyramidal(p, d_prev, d_curr, true, d_u, d_v, (hipStream_t)stream);
}

namespace {
int plan_pyramidal(oflk_plan *p, const void *d_prev_in, const void *d_curr_in, bool u8, float *d_u, float *d_v,
                   hipStream_t s)
{
    if (!p || !d_prev_in || !d_curr_in || !d_u || !d_v) return fail(OFLK_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    const int B = p->B, L = p->L, K = p->K;
    int rc;
    if (u8 && L > 1 &&
        !pyr_fused_fits(p->dims[2 * (L - 1)], p->dims[2 * (L - 1) + 1], p->dims[2 * (L - 2)], p->dims[2 * (L - 2) + 1], p->gauss)) {
        // the unfused pyramid kernels read float32: convert once and run the float path
        const size_t n = (size_t)B * p->H * p->W;
        size_t tot = 0;
        for (auto &q : p->u8_stage)
            if (!q && (rc = dmalloc(&q, n, &tot))) return rc;
        p->ws_bytes += tot;
        dim3 grid((unsigned)((n + 4095) / 4096));
        hipLaunchKernelGGL(k_u8_to_f32, grid, dim3(256), 0, s, static_cast<const unsigned char *>(d_prev_in), p->u8_stage[0], n);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_u8_to_f32, grid, dim3(256), 0, s, static_cast<const unsigned char *>(d_curr_in), p->u8_stage[1], n);
        HIP_TRY(hipGetLastError());
        return plan_pyramidal(p, p->u8_stage[0], p->u8_stage[1], false, d_u, d_v, s);
    }
    // typed float for the common case; with u8 the kernels of the finest level read them as uint8
    const float *d_prev = static_cast<const float *>(d_prev_in), *d_curr = static_cast<const float *>(d_curr_in);

    // every level's flow lives in two interleaved {u, v} ping-pong slots of the plan; only the finest
    // level's last launch writes the caller's planar planes (k_export_fixup de-interleaves the rest)
    if (!p->flow[0]) {   // a plan created with levels = 1, iters = 0 (single-scale use) asked for a pyramidal pass after all
        for (int l = 0; l < L; l++)
            if ((rc = dmalloc(&p->flow[l], (size_t)2 * 2 * B * p->npix(l), &p->ws_bytes))) return rc;
    }

    if (!tiled_window(p->hw)) {
        // 1x1 / 13x13 and larger windows have no fused iteration kernel: every pair runs the reference's own sequence of
        // steps with the standalone kernels (resolve_pair: pyramid, warp, generic single-scale LK, flow += d, upsample,
        // np.mean in NumPy's order, the exit test on the host) -- exact, and as slow as that sounds
        for (int b = 0; b < B; b++)
            if ((rc = resolve_pair(p, b, d_prev_in, d_curr_in, u8, d_u, d_v, s))) return rc;
        return OFLK_OK;
    }

    // per-call state (acc, iters_run, log) = 0 and flow = zeros at the coarsest level (:182-184):
    // carried by the first pyramid launch, or a launch of its own when there is no pyramid
    PyrExtra first;
    first.zero_words = reinterpret_cast<unsigned *>(p->state);
    first.n_zero_words = p->state_words();
    first.zero_u = reinterpret_cast<float *>(p->fl(0, 0));   // slot 0 of the coarsest level: 2 * B * n floats
    first.zero_v = first.zero_u + (size_t)B * p->npix(0);
    first.n_zero_flow = (size_t)B * p->npix(0);
    if (L == 1) {
        rc = launch_call_init(p, s, first);
        if (rc) return rc;
    }

    // ---- pyramids (lucas_kanade_pyramidal.py:173-174), fine -> coarse ---------
    for (int l = L - 2; l >= 0; l--) {
        int h = p->dims[2 * (l + 1)], w = p->dims[2 * (l + 1) + 1];
        int ho = p->dims[2 * l], wo = p->dims[2 * l + 1];
        if (l == L - 2) {
            // the finest level is the caller's frames (image.copy() at :40 is a no-op here):
            // prev and curr in one launch, images 0..B-1 from d_prev, B..2B-1 from d_curr
            first.in2 = d_curr;
            first.nsplit = B;
            first.u8 = u8;
            rc = launch_pyr_down(p, p->gauss, s, d_prev, p->pyr[l], p->tmpA, p->tmpB, 2 * B, h, w, ho, wo, &first);
            if (rc) return rc;
        } else {
            rc = launch_pyr_down(p, p->gauss, s, p->pyr[l + 1], p->pyr[l], p->tmpA, p->tmpB, 2 * B, h, w, ho, wo);
            if (rc) return rc;
        }
    }

    for (int l = 0; l < L; l++) {
        const int h = p->dims[2 * l], w = p->dims[2 * l + 1];
        const size_t n = (size_t)h * w;
        // tolerant mode: the two finest levels take the streaming kernel (order-free window sums, fused-lerp warp), and the flow
        // upsampling into such a level is fused into its first iteration
        const bool stream = p->arith == OFLK_ARITH_TOLERANT && p->hw == 2 && l >= L - 2 && h > 4 && w > 4;
        const bool fuse_up = stream && l > 0 && K >= 1 && p->dims[2 * (l - 1)] >= 2 && p->dims[2 * (l - 1) + 1] >= 2;
        if (l > 0 && !fuse_up) {
            // upsample_flow (:195-197) from whichever slot holds level l-1's result
            const int hc = p->dims[2 * (l - 1)], wc = p->dims[2 * (l - 1) + 1];
            ResampleArgs r{};
            // interleaved planes: slot s of level l-1 sits s * (B*n_c) float2 elements after slot 0
            r.interleaved = 1;
            r.in[0] = reinterpret_cast<const float *>(p->fl(l - 1, 0));
            r.in[1] = nullptr;
            r.acc = p->acc();
            r.acc_level = l - 1; r.L = L; r.K = p->Kc(); r.iters = K;
            r.acc_thr = conv_threshold((double)p->npix(l - 1));
            r.in_sel_stride = (size_t)B * p->npix(l - 1);
            r.out[0] = reinterpret_cast<float *>(p->fl(l, 0));
            r.out[1] = nullptr;
            r.scale[0] = (float)((double)w / (double)wc);  // scale_x (:123, :135)
            r.scale[1] = (float)((double)h / (double)hc);  // scale_y (:122, :136)
            r.H = hc; r.W = wc; r.Ho = h; r.Wo = w;
            r.ly = make_linspace(hc, h);
            r.lx = make_linspace(wc, w);
            r.nplanes = 2;
            r.apply_scale = 1;
            rc = launch_upsample(p, s, r, B);
            if (rc) return rc;
        }
        const float *lp = (l == L - 1) ? d_prev : p->pyr[l];
        const float *lc = (l == L - 1) ? d_curr : p->pyr[l] + (size_t)B * n;
        for (int k = 0; k < K; k++) {
            LkArgs a{};
            a.prev = lp; a.curr = lc;
            a.fl[0] = p->fl(l, 0); a.fl[1] = p->fl(l, 1);
            a.ou = d_u; a.ov = d_v;
            a.planar_out = (l == L - 1 && k == K - 1) ? 1 : 0;
            a.acc = p->acc();
            a.conv_thr = conv_threshold((double)n);
            a.level = l; a.iter = k; a.L = L; a.K = p->Kc();
            a.H = h; a.W = w;
            if (fuse_up && k == 0) {
                const int hc = p->dims[2 * (l - 1)], wc = p->dims[2 * (l - 1) + 1];
                a.up_src = p->fl(l - 1, 0);
                a.up_slot_stride = (size_t)B * p->npix(l - 1);
                a.up_level = l - 1;
                a.up_iters = K;
                a.up_thr = conv_threshold((double)p->npix(l - 1));
                a.Hc = hc; a.Wc = wc;
                a.up_ly = make_linspace(hc, h);
                a.up_lx = make_linspace(wc, w);
                a.up_sx = (float)((double)w / (double)wc);   // scale_x (:123, :135)
                a.up_sy = (float)((double)h / (double)hc);   // scale_y (:122, :136)
            }
            if (stream) rc = launch_lks<MODE_ITER>(p, s, l == L - 1 ? KC_LK_ITER_FINEST : KC_LK_ITER, a, B, u8 && l == L - 1, WARP_LERP64);
            else rc = launch_lk<MODE_ITER>(p, s, l == L - 1 ? KC_LK_ITER_FINEST : KC_LK_ITER, p->hw, a, B, u8 && l == L - 1);
            if (rc) return rc;
        }
    }
    // pairs whose finest level exited early hold their result in the internal slot
    {
        ExportArgs e{};
        e.src[0] = p->fl(L - 1, 0);
        e.src[1] = p->fl(L - 1, 1);
        e.dst_u = d_u;
        e.dst_v = d_v;
        e.acc = p->acc();
        e.want = 0;
        e.L = L; e.K = p->Kc(); e.iters = K;
        for (int l = 0; l < L; l++) {
            e.counts[l] = (double)p->npix(l);
            e.thr[l] = conv_threshold(e.counts[l]);
        }
        e.log = p->log();
        e.iters_run = p->iters_run();
        e.uncertain = p->uncertain();
        for (int l = 0; l < L; l++) {
            const double t = (double)e.thr[l];
            const double g = decision_guard(e.counts[l]);
            e.guard_lo[l] = (unsigned long long)std::floor(t * (1.0 - g));
            e.guard_hi[l] = (unsigned long long)std::ceil(t * (1.0 + g));
        }
        e.plane = (size_t)p->H * p->W;
        // few blocks per pair: the copy is the rare case, the common one is "nothing to do"
        dim3 grid((unsigned)std::min<size_t>((e.plane + 255) / 256, 128), B);
        Prof pr(p, s, KC_EXPORT);
        hipLaunchKernelGGL(k_export_fixup, grid, dim3(256), 0, s, e);
        HIP_TRY(hipGetLastError());
    }
    return OFLK_OK;
}
}  // namespace

OFLK_API int oflk_plan_read_log(oflk_plan *p, float *residual_log, int *iters_run, void *stream)
{
    if (!p) return fail(OFLK_ERR_INVALID, "NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    if (residual_log)
        HIP_TRY(hipMemcpyAsync(residual_log, p->log(),
                               (size_t)p->B * p->L * std::max(p->K, 1) * 2 * sizeof(float),
                               hipMemcpyDeviceToHost, s));
    if (iters_run)
        HIP_TRY(hipMemcpyAsync(iters_run, p->iters_run(), (size_t)p->B * p->L * sizeof(int),
                               hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return OFLK_OK;
}

#ifdef OFLK_STAMPS
// diagnostic build only (tools/stamps.py): [blocks][4 waves][8 tiles][16] s_memtime stamps of the last
// finest-level iteration launch; returns the number of blocks
OFLK_API long oflk_debug_stamps(oflk_plan *p, unsigned *out, long max_blocks)
{
    if (!p || !p->stamps) return 0;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    const long n = std::min<long>((long)p->stamps_blocks, max_blocks);
    if (out && hipMemcpy(out, p->stamps, (size_t)n * 512 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long)p->stamps_blocks;
}

// [blocks][8]: entry, loop start, loop end, exit on the 100 MHz chip-wide clock; HW_ID; XCC_ID; tiles; valid
OFLK_API long oflk_debug_block_times(oflk_plan *p, unsigned *out, long max_blocks)
{
    if (!p || !p->stamps) return 0;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    const long n = std::min<long>((long)p->stamps_blocks, max_blocks);
    if (out && hipMemcpy(out, p->stamps + (size_t)p->stamps_blocks * 512, (size_t)n * 8 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (long)p->stamps_blocks;
}
#endif

namespace {

// np.mean(np.abs(d)) of a device plane in NumPy's own summation order (k_np_abs_piece_sums + the serial
// addition of the pieces, here on the host in fp32; this translation unit is built with -ffp-contract=off)
int np_mean_abs(oflk_plan *p, const float *d, size_t n, hipStream_t s, float *mean)
{
    const size_t pieces = (n + kNpPiece - 1) / kNpPiece;
    hipLaunchKernelGGL(k_np_abs_piece_sums, dim3((unsigned)((pieces + 63) / 64)), dim3(64), 0, s, d, n, p->exact.pieces);
    HIP_TRY(hipGetLastError());
    std::vector<float> h(pieces);
    HIP_TRY(hipMemcpyAsync(h.data(), p->exact.pieces, pieces * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    volatile float acc = 0.0f;   // every add rounded to fp32, in order
    for (size_t i = 0; i < pieces; i++) acc = acc + h[i];
    *mean = (float)((double)acc / (double)n);
    return OFLK_OK;
}

// One pair, the reference's own sequence of steps (lucas_kanade_pyramidal.py:173-223) with the standalone
// kernels -- pyramid, warp, single-scale LK, flow += d, upsample, all value-identical to the fused path --
// and the exit decision taken on the HOST from NumPy-order means.  Results and log replace pair b's.
int resolve_pair(oflk_plan *p, int b, const void *d_prev_in, const void *d_curr_in, bool u8, float *d_u, float *d_v, hipStream_t s)
{
    const int L = p->L, K = p->K, H = p->H, W = p->W;
    const size_t N = (size_t)H * W;
    oflk_plan::Exact &x = p->exact;
    int rc;
    if (!x.ready) {
        size_t tot = 0;
        for (int l = 0; l < L; l++) {
            const size_t n = p->npix(l);
            if (l < L - 1 && (rc = dmalloc(&x.pyr[l], 2 * n, &tot))) return rc;
            if ((rc = dmalloc(&x.u[l], n, &tot)) || (rc = dmalloc(&x.v[l], n, &tot))) return rc;
        }
        if ((rc = dmalloc(&x.warped, N, &tot)) || (rc = dmalloc(&x.du, N, &tot)) || (rc = dmalloc(&x.dv, N, &tot)) ||
            (rc = dmalloc(&x.f32[0], N, &tot)) || (rc = dmalloc(&x.f32[1], N, &tot)) ||
            (rc = dmalloc(&x.pieces, (N + kNpPiece - 1) / kNpPiece, &tot)))
            return rc;
        p->ws_bytes += tot;
        x.ready = true;
    }
    // the pair's frames as float32 planes
    const float *fp, *fc;
    if (u8) {
        dim3 grid((unsigned)((N + 4095) / 4096));
        hipLaunchKernelGGL(k_u8_to_f32, grid, dim3(256), 0, s, static_cast<const unsigned char *>(d_prev_in) + (size_t)b * N, x.f32[0], N);
        hipLaunchKernelGGL(k_u8_to_f32, grid, dim3(256), 0, s, static_cast<const unsigned char *>(d_curr_in) + (size_t)b * N, x.f32[1], N);
        HIP_TRY(hipGetLastError());
        fp = x.f32[0];
        fc = x.f32[1];
    } else {
        fp = static_cast<const float *>(d_prev_in) + (size_t)b * N;
        fc = static_cast<const float *>(d_curr_in) + (size_t)b * N;
    }
    for (int l = L - 2; l >= 0; l--) {
        const int h = p->dims[2 * (l + 1)], w = p->dims[2 * (l + 1) + 1], ho = p->dims[2 * l], wo = p->dims[2 * l + 1];
        if (l == L - 2) {
            PyrExtra two;
            two.in2 = fc;
            two.nsplit = 1;
            rc = launch_pyr_down(nullptr, p->gauss, s, fp, x.pyr[l], p->tmpA, p->tmpB, 2, h, w, ho, wo, &two);
        } else {
            rc = launch_pyr_down(nullptr, p->gauss, s, x.pyr[l + 1], x.pyr[l], p->tmpA, p->tmpB, 2, h, w, ho, wo);
        }
        if (rc) return rc;
    }
    std::vector<float> log((size_t)L * p->Kc() * 2, 0.0f);
    std::vector<int> runs((size_t)L, 0);
    HIP_TRY(hipMemsetAsync(x.u[0], 0, p->npix(0) * sizeof(float), s));   // flow = zeros at the coarsest level (:182-184)
    HIP_TRY(hipMemsetAsync(x.v[0], 0, p->npix(0) * sizeof(float), s));
    for (int l = 0; l < L; l++) {
        const int h = p->dims[2 * l], w = p->dims[2 * l + 1];
        const size_t n = (size_t)h * w;
        if (l > 0) {
            const int hc = p->dims[2 * (l - 1)], wc = p->dims[2 * (l - 1) + 1];
            ResampleArgs r{};
            r.in[0] = x.u[l - 1]; r.in[1] = x.v[l - 1];
            r.out[0] = x.u[l]; r.out[1] = x.v[l];
            r.scale[0] = (float)((double)w / (double)wc);
            r.scale[1] = (float)((double)h / (double)hc);
            r.H = hc; r.W = wc; r.Ho = h; r.Wo = w;
            r.ly = make_linspace(hc, h);
            r.lx = make_linspace(wc, w);
            r.nplanes = 2;
            r.apply_scale = 1;
            if ((rc = launch_upsample(nullptr, s, r, 1))) return rc;
        }
        const float *lp = (l == L - 1) ? fp : x.pyr[l];
        const float *lc = (l == L - 1) ? fc : x.pyr[l] + n;
        for (int k = 0; k < K; k++) {
            hipLaunchKernelGGL(k_warp, grid2d(w, h, 1), dim3(256), 0, s, lc, (const float *)x.u[l], (const float *)x.v[l], x.warped, h, w);
            HIP_TRY(hipGetLastError());
            LkArgs a{};
            a.prev = lp; a.curr = x.warped;
            a.ou = x.du; a.ov = x.dv;
            a.H = h; a.W = w;
            if ((rc = launch_lk<MODE_SINGLE>(nullptr, s, KC_LK_SINGLE, p->hw, a, 1))) return rc;
            hipLaunchKernelGGL(k_flow_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x.u[l], x.v[l], (const float *)x.du,
                               (const float *)x.dv, n);
            HIP_TRY(hipGetLastError());
            float mu, mv;
            if ((rc = np_mean_abs(p, x.du, n, s, &mu)) || (rc = np_mean_abs(p, x.dv, n, s, &mv))) return rc;
            log[((size_t)l * p->Kc() + k) * 2] = mu;
            log[((size_t)l * p->Kc() + k) * 2 + 1] = mv;
            runs[(size_t)l] = k + 1;
            if (mu < 0.01f && mv < 0.01f) break;   // :221-223
        }
    }
    HIP_TRY(hipMemcpyAsync(d_u + (size_t)b * N, x.u[L - 1], N * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_v + (size_t)b * N, x.v[L - 1], N * sizeof(float), hipMemcpyDeviceToDevice, s));
    // the coarser levels' final flows where oflk_plan_read_level_flow looks for them: the interleaved slot the
    // (new) iteration count of the level selects (strided copies: u into the .x, v into the .y of every float2)
    for (int l = 0; l < L - 1; l++) {
        const size_t n = p->npix(l);
        float *dst = reinterpret_cast<float *>(p->fl(l, runs[(size_t)l] & 1) + (size_t)b * n);
        HIP_TRY(hipMemcpy2DAsync(dst, sizeof(float2), x.u[l], sizeof(float), sizeof(float), n, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpy2DAsync(dst + 1, sizeof(float2), x.v[l], sizeof(float), sizeof(float), n, hipMemcpyDeviceToDevice, s));
    }
    // the pair's log, iteration counts and (cleared) flags in the plan's state, where read_log looks
    HIP_TRY(hipMemcpyAsync(p->log() + (size_t)b * L * p->Kc() * 2, log.data(), log.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(p->iters_run() + (size_t)b * L, runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(p->uncertain() + (size_t)b * L, 0, (size_t)L * sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));   // log / runs are host vectors
    return OFLK_OK;
}

int resolve_uncertain(oflk_plan *p, const void *d_prev, const void *d_curr, bool u8, float *d_u, float *d_v, hipStream_t s,
                      int *resolved)
{
    if (!p || !d_prev || !d_curr || !d_u || !d_v) return fail(OFLK_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    std::vector<int> flags((size_t)p->B * p->L);
    HIP_TRY(hipMemcpyAsync(flags.data(), p->uncertain(), flags.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int n = 0;
    for (int b = 0; b < p->B; b++) {
        bool any = false;
        for (int l = 0; l < p->L; l++) any = any || flags[(size_t)b * p->L + l] != 0;
        if (!any) continue;
        int rc = resolve_pair(p, b, d_prev, d_curr, u8, d_u, d_v, s);
        if (rc) return rc;
        n++;
    }
    if (resolved) *resolved = n;
    return OFLK_OK;
}

}  // namespace

OFLK_API int oflk_plan_resolve_uncertain(oflk_plan *p, const float *d_prev, const float *d_curr, float *d_u, float *d_v,
                                         void *stream, int *resolved)
{
    return resolve_uncertain(p, d_prev, d_curr, false, d_u, d_v, (hipStream_t)stream, resolved);
}

OFLK_API int oflk_plan_resolve_uncertain_u8(oflk_plan *p, const unsigned char *d_prev, const unsigned char *d_curr, float *d_u,
                                            float *d_v, void *stream, int *resolved)
{
    return resolve_uncertain(p, d_prev, d_curr, true, d_u, d_v, (hipStream_t)stream, resolved);
}

OFLK_API int oflk_last_resolved(void) { return t_resolved; }

OFLK_API int oflk_plan_read_uncertain(oflk_plan *p, int *uncertain, void *stream)
{
    if (!p || !uncertain) return fail(OFLK_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(uncertain, p->uncertain(), (size_t)p->B * p->L * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return OFLK_OK;
}

OFLK_API int oflk_plan_read_level_flow(oflk_plan *p, int level, int pair, float *u, float *v, void *stream)
{
    if (!p || !u || !v) return fail(OFLK_ERR_INVALID, "NULL argument");
    if (level < 0 || level >= p->L - 1)
        return fail(OFLK_ERR_INVALID, "level must be in [0,%d): the finest level's flow is the call's result", p->L - 1);
    if (pair < 0 || pair >= p->B) return fail(OFLK_ERR_INVALID, "pair %d out of range [0,%d)", pair, p->B);
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    int executed = 0;
    HIP_TRY(hipMemcpyAsync(&executed, p->iters_run() + (size_t)pair * p->L + level, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int slot = executed & 1;   // the ping-pong slot the level's last executed iteration wrote
    const size_t n = p->npix(level);
    std::vector<float2> both(n);
    HIP_TRY(hipMemcpyAsync(both.data(), p->fl(level, slot) + (size_t)pair * n, n * sizeof(float2), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; i++) {   // de-interleave on the host (a debugging / plotting path)
        u[i] = both[i].x;
        v[i] = both[i].y;
    }
    return OFLK_OK;
}

OFLK_API int oflk_plan_set_arithmetic(oflk_plan *p, int mode)
{
    if (!p) return fail(OFLK_ERR_INVALID, "NULL plan");
    if (mode != OFLK_ARITH_EXACT && mode != OFLK_ARITH_CONTRACTED && mode != OFLK_ARITH_TOLERANT)
        return fail(OFLK_ERR_INVALID, "arithmetic mode must be OFLK_ARITH_EXACT (0), OFLK_ARITH_CONTRACTED (1) or OFLK_ARITH_TOLERANT (2), got %d", mode);
    p->arith = mode;
    return OFLK_OK;
}

OFLK_API int oflk_set_host_arithmetic(int mode)
{
    if (mode != OFLK_ARITH_EXACT && mode != OFLK_ARITH_CONTRACTED && mode != OFLK_ARITH_TOLERANT)
        return fail(OFLK_ERR_INVALID, "arithmetic mode must be OFLK_ARITH_EXACT (0), OFLK_ARITH_CONTRACTED (1) or OFLK_ARITH_TOLERANT (2), got %d", mode);
    g_host_arith.store(mode);
    return OFLK_OK;
}

OFLK_API int oflk_multi_rehearsal(int workers)
{
    if (workers < 0 || workers > 64) return fail(OFLK_ERR_INVALID, "workers must be 0 ... 64, got %d", workers);
    g_multi_workers.store(workers);
    return OFLK_OK;
}

OFLK_API int oflk_plan_set_kernels(oflk_plan *p, int choice)
{
    if (!p) return fail(OFLK_ERR_INVALID, "NULL plan");
    if (choice != OFLK_KERNELS_AUTO && choice != OFLK_KERNELS_TILE && choice != OFLK_KERNELS_STREAM)
        return fail(OFLK_ERR_INVALID, "kernel choice must be OFLK_KERNELS_AUTO (0), OFLK_KERNELS_TILE (1) or OFLK_KERNELS_STREAM (2), got %d", choice);
    p->kernels = choice;
    return OFLK_OK;
}

OFLK_API int oflk_plan_set_profiling(oflk_plan *p, int enabled)
{
    if (!p) return fail(OFLK_ERR_INVALID, "NULL plan");
    p->prof = enabled != 0;
    p->prof_only = enabled == 2 ? KC_LK_ITER_FINEST : -1;  // 2: only the dominant kernel
    for (auto &e : p->pending) p->pool.push_back({e.a, e.b});
    p->pending.clear();
    for (int i = 0; i < KC_COUNT; i++) {
        p->acc_ms[i] = 0;
        p->acc_n[i] = 0;
    }
    return OFLK_OK;
}

OFLK_API int oflk_plan_kernel_times(oflk_plan *p, const char **names, double *total_ms,
                                    long *launches, int max_entries)
{
    if (!p) return fail(OFLK_ERR_INVALID, "NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    for (auto &e : p->pending) {
        HIP_TRY(hipEventSynchronize(e.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        p->acc_ms[e.cls] += ms;
        p->acc_n[e.cls] += 1;
        p->pool.push_back({e.a, e.b});
    }
    p->pending.clear();
    int n = std::min<int>(KC_COUNT, max_entries);
    for (int i = 0; i < n; i++) {
        if (names) names[i] = kClassNames[i];
        if (total_ms) total_ms[i] = p->acc_ms[i];
        if (launches) launches[i] = p->acc_n[i];
    }
    return n;
}

// =============================================================================
// host-pointer entry points
// =============================================================================
namespace {

struct Arena {
    std::vector<void *> blocks;
    ~Arena() { release(); }
    void release()
    {
        for (void *b : blocks) (void)hipFree(b);
        blocks.clear();
    }
    template <typename T>
    int get(T **p, size_t n)
    {
        size_t tot = 0;
        int rc = dmalloc(p, n, &tot);
        if (!rc) blocks.push_back(*p);
        return rc;
    }
};

// Per-device state of the host entry points: a small cache of plans keyed by shape (the verifier
// alternates single-scale and pyramidal calls of one size; a caller may mix a few sizes) and the
// device buffers the frames and flows are staged in.  Everything in a context lives on ITS device,
// so switching devices (oflk_set_device, the multi-GPU entry points) never mixes allocations.
constexpr int kMaxDevices = 64;
constexpr size_t kPlanCache = 6;

struct HostCtx {
    std::mutex mu;   // one host call at a time per device; different devices run concurrently
    std::vector<oflk_plan *> plans;                          // most recently used first
    float *io[4] = {nullptr, nullptr, nullptr, nullptr};    // device prev, curr, u, v
    size_t io_elems = 0;
    unsigned char *u8[2] = {nullptr, nullptr};              // device uint8 frames
    size_t u8_elems = 0;
    // chunked host calls (run_batch_chunked): two slots of chunk-sized frames in / flow out, three streams
    void *ring_in[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [slot][prev, curr], PIXELS
    float *ring_out[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [slot][u, v]
    size_t ring_in_bytes = 0, ring_out_elems = 0;
    hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
};
HostCtx g_ctx[kMaxDevices];

// lock the context of `dev` and make the device current for the calling thread
int acquire(int dev, HostCtx **out, std::unique_lock<std::mutex> &lk)
{
    if (dev < 0 || dev >= kMaxDevices) return fail(OFLK_ERR_INVALID, "device %d out of range", dev);
    lk = std::unique_lock<std::mutex>(g_ctx[dev].mu);
    int rc = ensure_device(dev);
    if (rc) return rc;
    *out = &g_ctx[dev];
    return OFLK_OK;
}

int host_plan(HostCtx &c, int dev, int B, int H, int W, int L, int win, int K, oflk_plan **out)
{
    for (size_t i = 0; i < c.plans.size(); i++) {
        oflk_plan *q = c.plans[i];
        if (q->B == B && q->H == H && q->W == W && q->L == L && q->win == win && q->K == K) {
            c.plans.erase(c.plans.begin() + (long)i);
            c.plans.insert(c.plans.begin(), q);
            q->arith = g_host_arith.load();
            *out = q;
            return OFLK_OK;
        }
    }
    oflk_plan *q = nullptr;
    int rc = oflk_plan_create(&q, dev, B, H, W, L, win, K);
    if (rc == OFLK_ERR_NOMEM && !c.plans.empty()) {
        // make room: drop every cached plan and try once more
        for (oflk_plan *old : c.plans) plan_free(old);
        c.plans.clear();
        rc = oflk_plan_create(&q, dev, B, H, W, L, win, K);
    }
    if (rc) return rc;
    q->arith = g_host_arith.load();
    c.plans.insert(c.plans.begin(), q);
    while (c.plans.size() > kPlanCache) {
        plan_free(c.plans.back());
        c.plans.pop_back();
    }
    *out = q;
    return OFLK_OK;
}

int host_io(HostCtx &c, size_t need)
{
    if (need <= c.io_elems) return OFLK_OK;
    for (auto &q : c.io) {
        if (q) (void)hipFree(q);
        q = nullptr;
    }
    c.io_elems = 0;
    size_t tot = 0;
    for (auto &q : c.io) {
        int rc = dmalloc(&q, need, &tot);
        if (rc) return rc;
    }
    c.io_elems = need;
    return OFLK_OK;
}

int host_u8(HostCtx &c, size_t need)
{
    if (need <= c.u8_elems) return OFLK_OK;
    for (auto &q : c.u8) {
        if (q) (void)hipFree(q);
        q = nullptr;
    }
    c.u8_elems = 0;
    size_t tot = 0;
    for (auto &q : c.u8) {
        int rc = dmalloc(&q, need, &tot);
        if (rc) return rc;
    }
    c.u8_elems = need;
    return OFLK_OK;
}

int host_ring(HostCtx &c, size_t in_bytes, size_t out_elems)
{
    // each stream on its own: a call that failed half-way here must not leave a later one on the null (blocking) stream
    for (hipStream_t *st : {&c.s_in, &c.s_comp, &c.s_out}) {
        if (*st) continue;
        hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
        if (e != hipSuccess) {
            *st = nullptr;
            return fail(OFLK_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
        }
    }
    size_t tot = 0;
    if (in_bytes > c.ring_in_bytes) {
        for (auto &slot : c.ring_in)
            for (auto &q : slot) {
                if (q) (void)hipFree(q);
                q = nullptr;
            }
        c.ring_in_bytes = 0;
        for (auto &slot : c.ring_in)
            for (auto &q : slot) {
                unsigned char *b = nullptr;
                int rc = dmalloc(&b, in_bytes, &tot);
                if (rc) return rc;
                q = b;
            }
        c.ring_in_bytes = in_bytes;
    }
    if (out_elems > c.ring_out_elems) {
        for (auto &slot : c.ring_out)
            for (auto &q : slot) {
                if (q) (void)hipFree(q);
                q = nullptr;
            }
        c.ring_out_elems = 0;
        for (auto &slot : c.ring_out)
            for (auto &q : slot) {
                int rc = dmalloc(&q, out_elems, &tot);
                if (rc) return rc;
            }
        c.ring_out_elems = out_elems;
    }
    return OFLK_OK;
}

int check_hw(const void *a, const void *b, int H, int W)
{
    if (!a || !b) return fail(OFLK_ERR_INVALID, "NULL array argument");
    if (H < 1 || W < 1) return fail(OFLK_ERR_INVALID, "H and W must be >= 1 (got %d x %d)", H, W);
    if ((size_t)H * (size_t)W >= ((size_t)1 << 30) || H >= (1 << 24) || W >= (1 << 24))
        return fail(OFLK_ERR_UNSUPPORTED, "frames of 2^30 pixels or more are not supported");  // 32-bit byte offsets
    return OFLK_OK;
}

// A large host batch in chunks, so that the PCIe link works in both directions while the GPU computes:
//   main thread    H2D(k+1) on s_in   |  kernels(k) on s_comp  |  (exit decisions, log of chunk k)
//   second thread                         D2H(k-1) on s_out
// Two slots of chunk-sized device buffers; a chunk's inputs wait for the kernels two chunks back, its kernels for
// the D2H two chunks back.  The callers' arrays are ordinary pageable memory: a copy from / to them holds its host
// thread until it is done, which is why the two directions have a thread each.  Frame pairs are independent, so
// the results do not depend on the cut (tests/test_gpu_round3.py).  A synchronous float32 call moves 33 MB per
// 1080p pair over the link, half of it each way: overlapped, the floor is one direction's time.
template <class PIXELS>
int run_batch_chunked(HostCtx *c, int dev, const PIXELS *prev, const PIXELS *curr, int B, int C, int H, int W, int levels,
                      int window_size, int iters, float *u, float *v, float *residual_log, int *iters_run)
{
    constexpr bool U8 = sizeof(PIXELS) == 1;
    const bool single = levels == 0;
    const int Lp = single ? 1 : levels, Kp = single ? 0 : iters;
    const int Lc = std::max(levels, 1), Kc = std::max(iters, 1);
    const size_t plane = (size_t)H * W;
    const int nchunk = (B + C - 1) / C;
    int rc;
    if ((rc = host_ring(*c, (size_t)C * plane * sizeof(PIXELS), (size_t)C * plane))) return rc;
    oflk_plan *pc = nullptr, *pt = nullptr;
    if ((rc = host_plan(*c, dev, C, H, W, Lp, window_size, Kp, &pc))) return rc;
    const int tail = B - (nchunk - 1) * C;
    if (tail != C) {
        // Both plans must be alive at once.  A lookup that runs out of device memory frees EVERY cached plan (host_plan), the
        // other one of this pair included, so after the second lookup the first is looked up again and both are checked
        // against the cache; if the two do not fit the device together the batch cannot run chunked.
        if ((rc = host_plan(*c, dev, tail, H, W, Lp, window_size, Kp, &pt))) return rc;
        if ((rc = host_plan(*c, dev, C, H, W, Lp, window_size, Kp, &pc))) return rc;
        auto cached = [&](const oflk_plan *q) { return std::find(c->plans.begin(), c->plans.end(), q) != c->plans.end(); };
        if (!cached(pc) || !cached(pt))
            return fail(OFLK_ERR_NOMEM, "the chunk plan (%d pairs) and the tail plan (%d pairs) of %dx%d do not fit the device together", C, tail, W, H);
    }

    hipEvent_t ev_in[2] = {nullptr, nullptr};   // a slot's frames have arrived
    for (int i = 0; i < 2; i++) {
        if (hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming) != hipSuccess) {
            if (ev_in[0]) (void)hipEventDestroy(ev_in[0]);
            return fail(OFLK_ERR_HIP, "hipEventCreate");
        }
    }
    // hand-over to the D2H thread: chunks [0, computed) are ready to leave, chunks [0, copied) have left
    std::mutex mu;
    std::condition_variable cv;
    int computed = 0, copied = 0, out_rc = OFLK_OK;
    bool stop = false;
    std::string out_msg;
    std::thread out_thread;
    auto out_body = [&]() {
        if (ensure_device(dev)) { std::lock_guard<std::mutex> g(mu); out_rc = OFLK_ERR_HIP; out_msg = t_err; cv.notify_all(); return; }
        for (int k = 0; k < nchunk; k++) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return computed > k || stop; });
                if (computed <= k) return;
            }
            const int slot = k & 1, nb = k == nchunk - 1 ? tail : C;
            const size_t off = (size_t)k * C * plane, bytes = (size_t)nb * plane * sizeof(float);
            hipError_t e = hipMemcpyAsync(u + off, c->ring_out[slot][0], bytes, hipMemcpyDeviceToHost, c->s_out);
            if (e == hipSuccess) e = hipMemcpyAsync(v + off, c->ring_out[slot][1], bytes, hipMemcpyDeviceToHost, c->s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(c->s_out);
            std::lock_guard<std::mutex> g(mu);
            if (e != hipSuccess) {
                out_rc = OFLK_ERR_HIP;
                out_msg = std::string("D2H of a chunk: ") + hipGetErrorString(e);
                cv.notify_all();
                return;
            }
            copied = k + 1;
            cv.notify_all();
        }
    };
    try {
        out_thread = std::thread(out_body);
    } catch (...) {   // no exception leaves the C ABI
        for (int i = 0; i < 2; i++) (void)hipEventDestroy(ev_in[i]);
        return fail(OFLK_ERR_HIP, "could not start the D2H thread of a chunked batch");
    }
    auto finish = [&](int code) {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv.notify_all();
        out_thread.join();
        if (code != OFLK_OK || out_rc != OFLK_OK) {
            // an abandoned batch may still have copies and kernels in flight on the ring buffers: the next call reuses them
            (void)hipStreamSynchronize(c->s_in);
            (void)hipStreamSynchronize(c->s_comp);
            (void)hipStreamSynchronize(c->s_out);
        }
        for (int i = 0; i < 2; i++) (void)hipEventDestroy(ev_in[i]);
        if (code == OFLK_OK && out_rc != OFLK_OK) return fail(out_rc, "%s", out_msg.c_str());
        return code;
    };
    auto h2d = [&](int k) -> int {
        const int slot = k & 1, nb = k == nchunk - 1 ? tail : C;
        const size_t off = (size_t)k * C * plane, bytes = (size_t)nb * plane * sizeof(PIXELS);
        HIP_TRY(hipMemcpyAsync(c->ring_in[slot][0], prev + off, bytes, hipMemcpyHostToDevice, c->s_in));
        HIP_TRY(hipMemcpyAsync(c->ring_in[slot][1], curr + off, bytes, hipMemcpyHostToDevice, c->s_in));
        HIP_TRY(hipEventRecord(ev_in[slot], c->s_in));
        return OFLK_OK;
    };
    int resolved_total = 0;
    if ((rc = h2d(0))) return finish(rc);
    for (int k = 0; k < nchunk; k++) {
        const int slot = k & 1, nb = k == nchunk - 1 ? tail : C;
        oflk_plan *p = nb == C ? pc : pt;
        if (k >= 2) {   // the slot's flow buffers are free once chunk k-2 has left
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return copied >= k - 1 || out_rc != OFLK_OK; });
            if (out_rc != OFLK_OK) { g.unlock(); return finish(OFLK_OK); }
        }
        if ((rc = hipStreamWaitEvent(c->s_comp, ev_in[slot], 0)) != hipSuccess) return finish(fail(OFLK_ERR_HIP, "hipStreamWaitEvent"));
        const void *dp = c->ring_in[slot][0], *dc = c->ring_in[slot][1];
        float *du = c->ring_out[slot][0], *dv = c->ring_out[slot][1];
        rc = single ? plan_single_scale(p, dp, dc, U8, du, dv, c->s_comp) : plan_pyramidal(p, dp, dc, U8, du, dv, c->s_comp);
        if (rc) return finish(rc);
        // the next chunk's frames travel while this one is computed (the kernels that read its slot, chunk k-1's, are done:
        // the previous turn of this loop ended by waiting for them)
        if (k + 1 < nchunk && (rc = h2d(k + 1))) return finish(rc);
        if (!single && iters > 0) {
            int n = 0;
            if ((rc = resolve_uncertain(p, dp, dc, U8, du, dv, c->s_comp, &n))) return finish(rc);
            resolved_total += n;
        }
        if (single) {
            if (hipStreamSynchronize(c->s_comp) != hipSuccess) return finish(fail(OFLK_ERR_HIP, "hipStreamSynchronize"));
        } else {
            rc = oflk_plan_read_log(p, residual_log ? residual_log + (size_t)k * C * Lc * Kc * 2 : nullptr,
                                    iters_run ? iters_run + (size_t)k * C * Lc : nullptr, c->s_comp);
            if (rc) return finish(rc);
        }
        {
            std::lock_guard<std::mutex> g(mu);
            computed = k + 1;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return copied >= nchunk || out_rc != OFLK_OK; });
    }
    t_resolved += resolved_total;
    return finish(OFLK_OK);
}

// One batch on one device, host pointers in and out.  PIXELS: float or unsigned char frames (the
// uint8 kernels read the frames as they are: no float32 copy of them exists on the device).
// levels == 0 selects single-scale.
template <class PIXELS>
int run_batch_on(int dev, const PIXELS *prev, const PIXELS *curr, int B, int H, int W, int levels, int window_size,
                 int iters, float *u, float *v, float *residual_log, int *iters_run)
{
    constexpr bool U8 = sizeof(PIXELS) == 1;
    t_resolved = 0;   // of THIS call (oflk_last_resolved)
    int rc = check_hw(prev, curr, H, W);
    if (rc) return rc;
    if (!u || !v) return fail(OFLK_ERR_INVALID, "NULL output");
    if (B < 1) return fail(OFLK_ERR_INVALID, "B must be >= 1");
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(dev, &c, lk))) return rc;
    const bool single = levels == 0;
    {
        // chunks of ~32 MB of flow per plane (four 1080p pairs, one 4K pair); worth it from four chunks on
        const size_t pair_out = (size_t)H * W * sizeof(float);
        const int C = (int)std::max<size_t>(1, ((size_t)32 << 20) / std::max<size_t>(pair_out, 1));
        if (B >= 4 * C && (size_t)B * pair_out >= ((size_t)64 << 20))
            return run_batch_chunked<PIXELS>(c, dev, prev, curr, B, C, H, W, levels, window_size, iters, u, v, residual_log, iters_run);
    }
    oflk_plan *p = nullptr;
    if ((rc = host_plan(*c, dev, B, H, W, single ? 1 : levels, window_size, single ? 0 : iters, &p))) return rc;
    const size_t n = (size_t)B * H * W, obytes = n * sizeof(float);
    if ((rc = host_io(*c, n))) return rc;
    const void *dp, *dc;
    if (U8) {
        if ((rc = host_u8(*c, n))) return rc;
        HIP_TRY(hipMemcpyAsync(c->u8[0], prev, n, hipMemcpyHostToDevice, nullptr));
        HIP_TRY(hipMemcpyAsync(c->u8[1], curr, n, hipMemcpyHostToDevice, nullptr));
        dp = c->u8[0];
        dc = c->u8[1];
    } else {
        HIP_TRY(hipMemcpyAsync(c->io[0], prev, obytes, hipMemcpyHostToDevice, nullptr));
        HIP_TRY(hipMemcpyAsync(c->io[1], curr, obytes, hipMemcpyHostToDevice, nullptr));
        dp = c->io[0];
        dc = c->io[1];
    }
    rc = single ? plan_single_scale(p, dp, dc, U8, c->io[2], c->io[3], nullptr)
                : plan_pyramidal(p, dp, dc, U8, c->io[2], c->io[3], nullptr);
    if (rc) return rc;
    // exit decisions the device could not take with certainty are redone in NumPy's own summation order
    if (!single && iters > 0) {
        int n_res = 0;
        if ((rc = resolve_uncertain(p, dp, dc, U8, c->io[2], c->io[3], nullptr, &n_res))) return rc;
        t_resolved += n_res;
    }
    HIP_TRY(hipMemcpyAsync(u, c->io[2], obytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(v, c->io[3], obytes, hipMemcpyDeviceToHost, nullptr));
    if (single) {
        HIP_TRY(hipStreamSynchronize(nullptr));
        return OFLK_OK;
    }
    return oflk_plan_read_log(p, residual_log, iters_run, nullptr);
}

// contiguous share of `total` units for shard `i` of `n`; sizes differ by at most one
// (the rule of optical-flow-fpga_amd/python/oflk_dist.py shard_range)
void shard_range(int total, int i, int n, int *begin, int *end)
{
    const int base = total / n, extra = total % n;
    *begin = i * base + std::min(i, extra);
    *end = *begin + base + (i < extra ? 1 : 0);
}

// Frame pairs are independent units (lucas_kanade_pyramidal.py:141-228 touches only its two inputs), and how long one
// takes depends on its data (the early exit of :221-223 ends a level after one iteration or after all of them), so the
// devices do not get fixed shards: the batch is cut into chunks of consecutive pairs and every device's host thread pulls
// the next chunk from a shared counter until none is left -- a device whose pairs converge early simply takes more chunks.
// No data crosses between devices, and a pair's result does not depend on which device computed it or in which chunk.
template <class PIXELS>
int run_batch_multi(const PIXELS *prev, const PIXELS *curr, int B, int H, int W, int levels, int window_size, int iters,
                    int n_gpus, float *u, float *v, float *residual_log, int *iters_run)
{
    int rc = check_hw(prev, curr, H, W);
    if (rc) return rc;
    if (!u || !v) return fail(OFLK_ERR_INVALID, "NULL output");
    if (B < 1) return fail(OFLK_ERR_INVALID, "B must be >= 1");
    const int ndev = oflk_device_count();
    if (ndev < 1) return fail(OFLK_ERR_NO_DEVICE, "no usable HIP device; liboflk has no CPU path");
    if (n_gpus <= 0) n_gpus = ndev;
    if (n_gpus > ndev) return fail(OFLK_ERR_INVALID, "n_gpus = %d but %d device(s) visible", n_gpus, ndev);
    const int rehearsal = g_multi_workers.load();
    int workers_n = rehearsal > 0 ? rehearsal : n_gpus;
    workers_n = std::min(workers_n, B);   // never more workers than pairs
    if (workers_n == 1)
        return run_batch_on<PIXELS>(g_device.load(), prev, curr, B, H, W, levels, window_size, iters, u, v, residual_log,
                                    iters_run);
    const size_t plane = (size_t)H * W;
    const int Lc = std::max(levels, 1), Kc = std::max(iters, 1);
    // chunks: about four per worker, so that the last ones even out what the data made uneven; at least one pair
    const int chunk = std::max(1, (B + 4 * workers_n - 1) / (4 * workers_n));
    std::atomic<int> next{0};
    std::vector<int> codes((size_t)workers_n, OFLK_OK), redone((size_t)workers_n, 0);
    std::vector<std::string> msgs((size_t)workers_n);
    std::atomic<bool> failed{false};
    std::vector<std::thread> workers;
    for (int g = 0; g < workers_n; g++) {
        workers.emplace_back([&, g]() {
            const int dev = g % n_gpus;
            while (!failed.load()) {
                const int b0 = next.fetch_add(chunk);
                if (b0 >= B) break;
                const int b1 = std::min(b0 + chunk, B);
                const size_t off = (size_t)b0 * plane;
                const int c = run_batch_on<PIXELS>(dev, prev + off, curr + off, b1 - b0, H, W, levels, window_size, iters,
                                                   u + off, v + off,
                                                   residual_log ? residual_log + (size_t)b0 * Lc * Kc * 2 : nullptr,
                                                   iters_run ? iters_run + (size_t)b0 * Lc : nullptr);
                redone[(size_t)g] += t_resolved;                  // the worker's count of pairs redone
                if (c) {
                    codes[(size_t)g] = c;
                    msgs[(size_t)g] = t_err;                      // the worker's thread-local message
                    failed.store(true);                           // the others stop after their current chunk
                    break;
                }
            }
        });
    }
    for (auto &w : workers) w.join();
    t_resolved = 0;
    for (int g = 0; g < workers_n; g++) t_resolved += redone[(size_t)g];
    for (int g = 0; g < workers_n; g++)
        if (codes[(size_t)g]) return fail(codes[(size_t)g], "device %d: %s", g % n_gpus, msgs[(size_t)g].c_str());
    return OFLK_OK;
}

}  // namespace

OFLK_API int oflk_single_scale_batch(const float *prev, const float *curr, int B, int H, int W,
                                     int window_size, float *u, float *v)
{
    return run_batch_on<float>(g_device.load(), prev, curr, B, H, W, 0, window_size, 0, u, v, nullptr, nullptr);
}

OFLK_API int oflk_single_scale(const float *prev, const float *curr, int H, int W, int window_size,
                               float *u, float *v)
{
    return oflk_single_scale_batch(prev, curr, 1, H, W, window_size, u, v);
}

OFLK_API int oflk_pyramidal_batch(const float *prev, const float *curr, int B, int H, int W,
                                  int levels, int window_size, int iters, float *u, float *v,
                                  float *residual_log, int *iters_run)
{
    if (levels < 1) return fail(OFLK_ERR_INVALID, "levels must be in [1,%d] (got %d)", OFLK_MAX_LEVELS, levels);
    return run_batch_on<float>(g_device.load(), prev, curr, B, H, W, levels, window_size, iters, u, v, residual_log,
                               iters_run);
}

OFLK_API int oflk_pyramidal(const float *prev, const float *curr, int H, int W, int levels,
                            int window_size, int iters, float *u, float *v, float *residual_log,
                            int *iters_run)
{
    return oflk_pyramidal_batch(prev, curr, 1, H, W, levels, window_size, iters, u, v, residual_log,
                                iters_run);
}

// ---- uint8 ingestion: raw 8-bit frames as the reference stores them (frame_0x.bin,
// generate_test_suite.py:259-261).  The kernels read the bytes directly (4 pixels per dword in the
// single-scale staging, byte gathers in the warp): the uint8 -> float32 conversion the verifier
// performs on the host (optical_flow_verifier.py:61-65) happens in registers ------------------
OFLK_API int oflk_single_scale_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                                  int window_size, float *u, float *v)
{
    return run_batch_on<unsigned char>(g_device.load(), prev, curr, B, H, W, 0, window_size, 0, u, v, nullptr, nullptr);
}

OFLK_API int oflk_pyramidal_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                               int levels, int window_size, int iters, float *u, float *v,
                               float *residual_log, int *iters_run)
{
    if (levels < 1) return fail(OFLK_ERR_INVALID, "levels must be in [1,%d] (got %d)", OFLK_MAX_LEVELS, levels);
    return run_batch_on<unsigned char>(g_device.load(), prev, curr, B, H, W, levels, window_size, iters, u, v,
                                       residual_log, iters_run);
}

OFLK_API int oflk_single_scale_fp16(const float *prev, const float *curr, int B, int H, int W, int window_size,
                                    float pixel_max, float *u, float *v)
{
    int rc = check_hw(prev, curr, H, W);
    if (rc) return rc;
    if (!u || !v) return fail(OFLK_ERR_INVALID, "NULL output");
    if (B < 1) return fail(OFLK_ERR_INVALID, "B must be >= 1");
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    oflk_plan *p = nullptr;
    if ((rc = host_plan(*c, g_device.load(), B, H, W, 1, window_size, 0, &p))) return rc;
    const size_t n = (size_t)B * H * W, bytes = n * sizeof(float);
    if ((rc = host_io(*c, n))) return rc;
    HIP_TRY(hipMemcpyAsync(c->io[0], prev, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(c->io[1], curr, bytes, hipMemcpyHostToDevice, nullptr));
    if ((rc = oflk_plan_single_scale_fp16(p, c->io[0], c->io[1], c->io[2], c->io[3], pixel_max, nullptr))) return rc;
    HIP_TRY(hipMemcpyAsync(u, c->io[2], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(v, c->io[3], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}

// ---- RTL-bit-accurate integer mode ---------------------------------
namespace {
int check_rtl(const void *prev, const void *curr, int B, int H, int W, const void *u, const void *v)
{
    if (!prev || !curr || !u || !v) return fail(OFLK_ERR_INVALID, "NULL argument");
    if (B < 1) return fail(OFLK_ERR_INVALID, "B must be >= 1");
    if (H < 5 || W < 5) return fail(OFLK_ERR_INVALID, "the RTL's line buffers need H, W >= 5 (got %d x %d)", H, W);
    if (W > kRtlMaxW || H > kRtlMaxH)
        return fail(OFLK_ERR_UNSUPPORTED, "the RTL's coordinate ports are 10 / 9 bits wide: W <= %d, H <= %d (got %d x %d)", kRtlMaxW,
                    kRtlMaxH, W, H);
    return OFLK_OK;
}
}  // namespace

OFLK_API long oflk_rtl_stream_length(int H, int W)
{
    return H < 5 || W < 5 ? 0 : (long)(H - 4) * (long)(W - 4);
}

OFLK_API int oflk_rtl_flow_u8_device(const unsigned char *d_prev, const unsigned char *d_curr, int B, int H, int W, short *d_u,
                                     short *d_v, void *stream)
{
    int rc = check_rtl(d_prev, d_curr, B, H, W, d_u, d_v);
    if (rc) return rc;
    RtlArgs a{};
    a.prev = d_prev; a.curr = d_curr; a.u = d_u; a.v = d_v;
    a.H = H; a.W = W; a.B = B;
    const long M = oflk_rtl_stream_length(H, W);
    dim3 grid((unsigned)((M + kRtlChunk - 1) / kRtlChunk), (unsigned)B);
    const size_t lds = (size_t)(4 * W + kRtlChunk) * sizeof(short4);   // <= 48 KB at W = 1024
    hipLaunchKernelGGL(k_rtl_flow, grid, dim3(256), lds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

OFLK_API int oflk_rtl_flow_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W, short *u, short *v)
{
    int rc = check_rtl(prev, curr, B, H, W, u, v);
    if (rc) return rc;
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    const size_t n = (size_t)B * H * W, m = (size_t)B * (size_t)oflk_rtl_stream_length(H, W);
    if ((rc = host_u8(*c, n))) return rc;
    if ((rc = host_io(*c, (m + 1) / 2))) return rc;   // two int16 planes in the float32 scratch planes 2 and 3
    short *d_u = reinterpret_cast<short *>(c->io[2]), *d_v = reinterpret_cast<short *>(c->io[3]);
    HIP_TRY(hipMemcpyAsync(c->u8[0], prev, n, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(c->u8[1], curr, n, hipMemcpyHostToDevice, nullptr));
    if ((rc = oflk_rtl_flow_u8_device(c->u8[0], c->u8[1], B, H, W, d_u, d_v, nullptr))) return rc;
    HIP_TRY(hipMemcpyAsync(u, d_u, m * sizeof(short), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(v, d_v, m * sizeof(short), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}

// ---- one process, several GPUs: the batch sharded over devices 0 .. n_gpus-1 ---------------
OFLK_API int oflk_single_scale_batch_multi(const float *prev, const float *curr, int B, int H, int W,
                                           int window_size, int n_gpus, float *u, float *v)
{
    return run_batch_multi<float>(prev, curr, B, H, W, 0, window_size, 0, n_gpus, u, v, nullptr, nullptr);
}

OFLK_API int oflk_pyramidal_batch_multi(const float *prev, const float *curr, int B, int H, int W, int levels,
                                        int window_size, int iters, int n_gpus, float *u, float *v,
                                        float *residual_log, int *iters_run)
{
    if (levels < 1) return fail(OFLK_ERR_INVALID, "levels must be in [1,%d] (got %d)", OFLK_MAX_LEVELS, levels);
    return run_batch_multi<float>(prev, curr, B, H, W, levels, window_size, iters, n_gpus, u, v, residual_log, iters_run);
}

OFLK_API int oflk_pyramidal_u8_multi(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                                     int levels, int window_size, int iters, int n_gpus, float *u, float *v,
                                     float *residual_log, int *iters_run)
{
    if (levels < 1) return fail(OFLK_ERR_INVALID, "levels must be in [1,%d] (got %d)", OFLK_MAX_LEVELS, levels);
    return run_batch_multi<unsigned char>(prev, curr, B, H, W, levels, window_size, iters, n_gpus, u, v, residual_log,
                                          iters_run);
}

namespace {
// the cached plan of an earlier host call of this shape on the current device (not created here)
int cached_plan(HostCtx &c, int B, int H, int W, int L, int win, int K, oflk_plan **out)
{
    for (oflk_plan *q : c.plans)
        if (q->B == B && q->H == H && q->W == W && q->L == L && q->win == win && q->K == K) {
            *out = q;
            return OFLK_OK;
        }
    return fail(OFLK_ERR_INVALID, "no pyramidal call of this shape (B=%d, %dx%d, %d levels, window %d, %d iterations) "
                                  "has run on this device yet", B, W, H, L, win, K);
}
}  // namespace

OFLK_API int oflk_pyramidal_last_level_flow(int B, int H, int W, int levels, int window_size, int iters, int level,
                                            int pair, float *u, float *v)
{
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    int rc = acquire(g_device.load(), &c, lk);
    if (rc) return rc;
    oflk_plan *p = nullptr;
    if ((rc = cached_plan(*c, B, H, W, levels, window_size, iters, &p))) return rc;
    return oflk_plan_read_level_flow(p, level, pair, u, v, nullptr);
}

OFLK_API int oflk_pyramidal_last_uncertain(int B, int H, int W, int levels, int window_size, int iters, int *uncertain)
{
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    int rc = acquire(g_device.load(), &c, lk);
    if (rc) return rc;
    oflk_plan *p = nullptr;
    if ((rc = cached_plan(*c, B, H, W, levels, window_size, iters, &p))) return rc;
    return oflk_plan_read_uncertain(p, uncertain, nullptr);
}

OFLK_API void oflk_shard_range(int total, int shard, int n_shards, int *begin, int *end)
{
    int b = 0, e = 0;
    if (n_shards >= 1 && shard >= 0 && shard < n_shards && total >= 0) shard_range(total, shard, n_shards, &b, &e);
    if (begin) *begin = b;
    if (end) *end = e;
}

namespace {

// device flows -> metrics; `dev_true` holds u_true[B] then v_true[B]
int metrics_device(const float *d_u, const float *d_v, int B, int H, int W, const float *u_true, const float *v_true,
                   int y0, int y1, int x0, int x1, double *out, hipStream_t s)
{
    if (!d_u || !d_v || !u_true || !v_true || !out) return fail(OFLK_ERR_INVALID, "NULL argument");
    if (B < 1 || H < 1 || W < 1) return fail(OFLK_ERR_INVALID, "B, H and W must be >= 1");
    // NumPy slice semantics for mask[y0:y1, x0:x1]: negative bounds count from the end, then clip
    auto norm = [](int i, int n) { return std::min(std::max(i < 0 ? i + n : i, 0), n); };
    y0 = norm(y0, H); y1 = norm(y1, H); x0 = norm(x0, W); x1 = norm(x1, W);
    const size_t count = (size_t)std::max(y1 - y0, 0) * (size_t)std::max(x1 - x0, 0);
    Arena ar;
    float *d_true = nullptr;
    double *d_part = nullptr;
    int rc = ar.get(&d_true, (size_t)2 * B);
    if (!rc) rc = ar.get(&d_part, (size_t)B * kMetricBlocks * kMetricTerms);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_true, u_true, (size_t)B * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_true + B, v_true, (size_t)B * sizeof(float), hipMemcpyHostToDevice, s));
    MetricArgs a{};
    a.u = d_u; a.v = d_v;
    a.u_true = d_true; a.v_true = d_true + B;
    a.H = H; a.W = W;
    a.y0 = y0; a.y1 = y1; a.x0 = x0; a.x1 = x1;
    a.partial = d_part;
    hipLaunchKernelGGL(k_flow_metrics, dim3(kMetricBlocks, (unsigned)B), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    std::vector<double> part((size_t)B * kMetricBlocks * kMetricTerms);
    HIP_TRY(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int b = 0; b < B; b++) {
        double t[kMetricTerms] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < kMetricBlocks; k++) {
            const double *q = &part[((size_t)b * kMetricBlocks + k) * kMetricTerms];
            for (int i = 0; i < kMetricTerms - 1; i++) t[i] += q[i];
            t[kMetricTerms - 1] = std::max(t[kMetricTerms - 1], q[kMetricTerms - 1]);
        }
        double *o = out + (size_t)b * 5;
        const double n = (double)count;   // an empty mask gives nan, as np.mean of an empty array does
        o[0] = (double)(float)(t[0] / n);
        o[1] = (double)(float)(t[1] / n);
        o[2] = (double)std::sqrt((float)(t[2] / n));   // np.sqrt of the fp32 mean (:69)
        o[3] = (double)(float)(t[3] / n);
        // "nothing moves and nothing was predicted" (:143-146)
        const double mt = std::sqrt((double)u_true[b] * u_true[b] + (double)v_true[b] * v_true[b]);
        o[4] = (mt < 1e-6 && t[5] < (double)1e-6f) ? 0.0 : (double)(float)(t[4] / n);
    }
    return OFLK_OK;
}

}  // namespace

OFLK_API int oflk_plan_metrics(oflk_plan *p, const float *d_u, const float *d_v, const float *u_true,
                               const float *v_true, int y0, int y1, int x0, int x1, double *out, void *stream)
{
    if (!p) return fail(OFLK_ERR_INVALID, "NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    return metrics_device(d_u, d_v, p->B, p->H, p->W, u_true, v_true, y0, y1, x0, x1, out, (hipStream_t)stream);
}

OFLK_API int oflk_flow_metrics(const float *u, const float *v, int B, int H, int W, const float *u_true,
                               const float *v_true, int y0, int y1, int x0, int x1, double *out)
{
    int rc = check_hw(u, v, H, W);
    if (rc) return rc;
    if (B < 1) return fail(OFLK_ERR_INVALID, "B must be >= 1");
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    Arena ar;
    const size_t n = (size_t)B * H * W;
    float *d_u = nullptr, *d_v = nullptr;
    if ((rc = ar.get(&d_u, n)) || (rc = ar.get(&d_v, n))) return rc;
    HIP_TRY(hipMemcpyAsync(d_u, u, n * sizeof(float), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d_v, v, n * sizeof(float), hipMemcpyHostToDevice, nullptr));
    return metrics_device(d_u, d_v, B, H, W, u_true, v_true, y0, y1, x0, x1, out, nullptr);
}

OFLK_API int oflk_u8_to_f32(const unsigned char *d_in, float *d_out, size_t n, void *stream)
{
    if (!d_in || !d_out) return fail(OFLK_ERR_INVALID, "NULL argument");
    if (n == 0) return OFLK_OK;
    dim3 grid((unsigned)((n + 4095) / 4096));
    hipLaunchKernelGGL(k_u8_to_f32, grid, dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
    HIP_TRY(hipGetLastError());
    return OFLK_OK;
}

OFLK_API int oflk_compute_gradients(const float *prev, const float *curr, int H, int W, float *Ix,
                                    float *Iy, float *It)
{
    int rc = check_hw(prev, curr, H, W);
    if (rc) return rc;
    if (!Ix || !Iy || !It) return fail(OFLK_ERR_INVALID, "NULL output");
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    Arena ar;
    size_t n = (size_t)H * W, bytes = n * sizeof(float);
    float *d[5];
    for (auto &q : d)
        if ((rc = ar.get(&q, n))) return rc;
    HIP_TRY(hipMemcpyAsync(d[0], prev, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d[1], curr, bytes, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_gradients, grid2d(W, H, 1), dim3(256), 0, nullptr, (const float *)d[0],
                       (const float *)d[1], d[2], d[3], d[4], H, W);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Ix, d[2], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(Iy, d[3], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(It, d[4], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}

OFLK_API int oflk_from_gradients(const float *Ix, const float *Iy, const float *It, int H, int W,
                                 int window_size, float *u, float *v)
{
    int rc = check_hw(Ix, Iy, H, W);
    if (rc) return rc;
    if (!It || !u || !v) return fail(OFLK_ERR_INVALID, "NULL argument");
    int hw = 0;
    if ((rc = window_hw(window_size, &hw))) return rc;
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    Arena ar;
    size_t n = (size_t)H * W, bytes = n * sizeof(float);
    float *d[5];
    for (auto &q : d)
        if ((rc = ar.get(&q, n))) return rc;
    HIP_TRY(hipMemcpyAsync(d[0], Ix, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d[1], Iy, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d[2], It, bytes, hipMemcpyHostToDevice, nullptr));
    LkArgs a{};
    a.prev = d[0]; a.curr = d[1]; a.aux = d[2];
    a.ou = d[3]; a.ov = d[4];
    a.H = H; a.W = W;
    rc = launch_lk<MODE_GRADS>(nullptr, nullptr, KC_LK_SINGLE, hw, a, 1);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(u, d[3], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(v, d[4], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}

namespace {
int build_pyramid_host(const float *image, int H, int W, int levels, double scale_factor, const GaussW *given, float *const *out_levels);
}

OFLK_API int oflk_build_pyramid(const float *image, int H, int W, int levels, double scale_factor,
                                float *const *out_levels)
{
    return build_pyramid_host(image, H, W, levels, scale_factor, nullptr, out_levels);
}

OFLK_API int oflk_build_pyramid_w(const float *image, int H, int W, int levels, double scale_factor, const double *weights,
                                  int radius, float *const *out_levels)
{
    if (!weights) return fail(OFLK_ERR_INVALID, "NULL weights");
    if (radius < 0 || radius > kMaxRadius) return fail(OFLK_ERR_UNSUPPORTED, "gaussian radius %d outside [0, %d]", radius, kMaxRadius);
    GaussW g;
    g.radius = radius;
    for (int k = 0; k <= radius; k++) g.w[k] = weights[k];
    return build_pyramid_host(image, H, W, levels, scale_factor, &g, out_levels);
}

namespace {
int build_pyramid_host(const float *image, int H, int W, int levels, double scale_factor, const GaussW *given, float *const *out_levels)
{
    int rc = check_hw(image, out_levels, H, W);
    if (rc) return rc;
    int dims[2 * OFLK_MAX_LEVELS];
    if ((rc = level_dims(H, W, levels, scale_factor, dims))) return rc;
    for (int l = 0; l < levels; l++)
        if (!out_levels[l]) return fail(OFLK_ERR_INVALID, "out_levels[%d] is NULL", l);
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    GaussW gauss;
    if (given) gauss = *given;
    else if ((rc = make_gauss(1.0 / scale_factor, &gauss))) return rc;
    Arena ar;
    size_t N = (size_t)H * W;
    float *cur = nullptr, *nxt = nullptr, *tA = nullptr, *tB = nullptr;
    if ((rc = ar.get(&cur, N)) || (rc = ar.get(&nxt, N)) || (rc = ar.get(&tA, N)) || (rc = ar.get(&tB, N)))
        return rc;
    HIP_TRY(hipMemcpyAsync(cur, image, N * sizeof(float), hipMemcpyHostToDevice, nullptr));
    std::memcpy(out_levels[levels - 1], image, N * sizeof(float));  // image.copy(), :40
    for (int l = levels - 2; l >= 0; l--) {
        int h = dims[2 * (l + 1)], w = dims[2 * (l + 1) + 1];
        int ho = dims[2 * l], wo = dims[2 * l + 1];
        rc = launch_pyr_down(nullptr, gauss, nullptr, cur, nxt, tA, tB, 1, h, w, ho, wo);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out_levels[l], nxt, (size_t)ho * wo * sizeof(float), hipMemcpyDeviceToHost,
                               nullptr));
        std::swap(cur, nxt);
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}
}  // namespace

OFLK_API int oflk_warp(const float *image, const float *flow_u, const float *flow_v, int H, int W,
                       float *out)
{
    int rc = check_hw(image, flow_u, H, W);
    if (rc) return rc;
    if (!flow_v || !out) return fail(OFLK_ERR_INVALID, "NULL argument");
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    Arena ar;
    size_t n = (size_t)H * W, bytes = n * sizeof(float);
    float *d[4];
    for (auto &q : d)
        if ((rc = ar.get(&q, n))) return rc;
    HIP_TRY(hipMemcpyAsync(d[0], image, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d[1], flow_u, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d[2], flow_v, bytes, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_warp, grid2d(W, H, 1), dim3(256), 0, nullptr, (const float *)d[0],
                       (const float *)d[1], (const float *)d[2], d[3], H, W);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d[3], bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}

OFLK_API int oflk_upsample_flow(const float *flow_u, const float *flow_v, int Hc, int Wc, int Ht,
                                int Wt, float *u_out, float *v_out)
{
    int rc = check_hw(flow_u, flow_v, Hc, Wc);
    if (rc) return rc;
    if ((rc = check_hw(u_out, v_out, Ht, Wt))) return rc;
    HostCtx *c = nullptr;
    std::unique_lock<std::mutex> lk;
    if ((rc = acquire(g_device.load(), &c, lk))) return rc;
    Arena ar;
    size_t nc = (size_t)Hc * Wc, nt = (size_t)Ht * Wt;
    float *du = nullptr, *dv = nullptr, *ou = nullptr, *ov = nullptr;
    if ((rc = ar.get(&du, nc)) || (rc = ar.get(&dv, nc)) || (rc = ar.get(&ou, nt)) || (rc = ar.get(&ov, nt)))
        return rc;
    HIP_TRY(hipMemcpyAsync(du, flow_u, nc * sizeof(float), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(dv, flow_v, nc * sizeof(float), hipMemcpyHostToDevice, nullptr));
    ResampleArgs r{};
    r.in[0] = du; r.in[1] = dv;
    r.out[0] = ou; r.out[1] = ov;
    r.scale[0] = (float)((double)Wt / (double)Wc);
    r.scale[1] = (float)((double)Ht / (double)Hc);
    r.H = Hc; r.W = Wc; r.Ho = Ht; r.Wo = Wt;
    r.ly = make_linspace(Hc, Ht);
    r.lx = make_linspace(Wc, Wt);
    r.nplanes = 2;
    r.apply_scale = 1;
    if ((rc = launch_upsample(nullptr, nullptr, r, 1))) return rc;
    HIP_TRY(hipMemcpyAsync(u_out, ou, nt * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(v_out, ov, nt * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OFLK_OK;
}
