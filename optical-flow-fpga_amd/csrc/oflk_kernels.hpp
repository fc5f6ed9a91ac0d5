// oflk_kernels.hpp -- gfx950 device code for dense Lucas-Kanade optical flow.
//
// Numerics contract (DESIGN.md "Exactness"): every kernel reproduces the IEEE
// operation sequence of the reference's NumPy/SciPy calls, so results are equal
// to the reference's value for value:
//   - fp32 stages (Sobel, products, window sums, 2x2 solve, flow accumulate) use
//     individually rounded fp32 ops in NumPy's order; the translation unit is
//     compiled with -ffp-contract=off and correctly rounded fp32 division;
//   - SciPy stages (Gaussian, bilinear sampling) accumulate in fp64 and round to
//     fp32 exactly where SciPy stores fp32.
// Reference line numbers are relative to /root/reference/python/.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

#include "../../include/oflk.h"
#include <stdint.h>

#pragma clang fp contract(off)

#ifndef OFLK_PYR_PAIRS
#define OFLK_PYR_PAIRS 1   // k_pyr_down: interior float32 tiles staged as column pairs (8-byte loads)
#endif
#ifndef OFLK_LK16_PF
#define OFLK_LK16_PF 2   // rows of frame loads in flight per wave (k_lk16d)
#endif

namespace oflk {

constexpr int kMaxRadius = 64;

// ---------------------------------------------------------------------------
// scipy.ndimage.map_coordinates(order=1, mode="constant", cval=0) at one point
// (lucas_kanade_pyramidal.py:59, :95, :131-132).  fp64 coordinates, weights
// w0 = 1 - frac, w1 = 1 - w0, taps accumulated (y0,x0),(y0,x1),(y1,x0),(y1,x1),
// each as (p * wy) * wx; hard zero outside [0, N-1].
// ---------------------------------------------------------------------------
struct BilinearTaps {
    int i00, i01, i10, i11;  // element offsets into the plane
    double wy0, wy1, wx0, wx1;
    bool inside;
};

__device__ __forceinline__ BilinearTaps bilinear_taps(int H, int W, double y, double x)
{
    BilinearTaps t;
    t.inside = !(y < 0.0 || y > (double)(H - 1) || x < 0.0 || x > (double)(W - 1));
    double fy = floor(y), fx = floor(x);
    int y0 = (int)fy, x0 = (int)fx;
    // keep addresses legal for non-finite coordinates (never produced from finite
    // inputs; the value is discarded or multiplied into NaN anyway)
    y0 = min(max(y0, 0), H - 1);
    x0 = min(max(x0, 0), W - 1);
    double ry = y - fy, rx = x - fx;
    t.wy0 = 1.0 - ry;
    t.wx0 = 1.0 - rx;
    t.wy1 = 1.0 - t.wy0;
    t.wx1 = 1.0 - t.wx0;
    // a tap past the last index only occurs with weight exactly 0; SciPy reads
    // the mirrored element there
    int y1 = (y0 + 1 < H) ? y0 + 1 : (H > 1 ? H - 2 : 0);
    int x1 = (x0 + 1 < W) ? x0 + 1 : (W > 1 ? W - 2 : 0);
    t.i00 = y0 * W + x0;
    t.i01 = y0 * W + x1;
    t.i10 = y1 * W + x0;
    t.i11 = y1 * W + x1;
    return t;
}

__device__ __forceinline__ float bilinear_apply(const float *__restrict__ img, const BilinearTaps &t)
{
    if (!t.inside) return 0.0f;
    double acc = 0.0, c;
    c = (double)img[t.i00]; c = c * t.wy0; c = c * t.wx0; acc = acc + c;
    c = (double)img[t.i01]; c = c * t.wy0; c = c * t.wx1; acc = acc + c;
    c = (double)img[t.i10]; c = c * t.wy1; c = c * t.wx0; acc = acc + c;
    c = (double)img[t.i11]; c = c * t.wy1; c = c * t.wx1; acc = acc + c;
    return (float)acc;
}

// same accumulation with the four taps already loaded
__device__ __forceinline__ float bilinear_finish(const BilinearTaps &t, float p00, float p01, float p10,
                                                 float p11)
{
    if (!t.inside) return 0.0f;
    double acc = 0.0, c;
    c = (double)p00; c = c * t.wy0; c = c * t.wx0; acc = acc + c;
    c = (double)p01; c = c * t.wy0; c = c * t.wx1; acc = acc + c;
    c = (double)p10; c = c * t.wy1; c = c * t.wx0; acc = acc + c;
    c = (double)p11; c = c * t.wy1; c = c * t.wx1; acc = acc + c;
    return (float)acc;
}

__device__ __forceinline__ float bilinear_f64(const float *__restrict__ img, int H, int W, double y,
                                              double x)
{
    BilinearTaps t = bilinear_taps(H, W, y, x);
    return bilinear_apply(img, t);
}

// clamp(x, 0, hi) in one instruction (the compiler only forms v_med3 for constant bounds)
__device__ __forceinline__ int clamp0(int x, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
    return r;
}

// Lean form of the same sampling for the fused iteration kernel and k_warp.
//   * the range test is two fp64 compares per axis on the coordinate itself;
//   * the 2x2 taps are two 8-byte pairs at byte offsets off0 and off0 + rowstep from the
//     plane base (32-bit offsets: a plane is < 4 GiB, checked on the host);
//   * a sample exactly on the last row/column (fraction 0) is expressed from the
//     previous cell instead: floor is capped at N-2, which makes the fraction exactly 1
//     and the weights exactly (0, 1).  SciPy reads the same two elements there (the
//     mirrored neighbour N-2 with weight 0); a zero product only moves inside the fp64
//     sum, which cannot change a sum that starts at +0 (finite pixels).
// Same values as bilinear_taps + bilinear_finish.
struct LeanGeom {
    double Hm1, Wm1;     // last valid coordinate
    double Hm2, Wm2;     // cap of the floor: max(N-2, 0)
    int W;
    unsigned rowstep;    // bytes from tap row 0 to tap row 1: 4W, or 0 when H == 1
    bool single;         // W == 1: the pair load would leave the row; taps are loaded one by one
};

// a wave-uniform double moved to scalar registers (int -> fp64 conversion only exists on
// the vector ALU, and its result would otherwise occupy a VGPR pair for the whole kernel)
__device__ __forceinline__ double uniform_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ LeanGeom lean_geom(int H, int W)
{
    LeanGeom g;
    g.Hm1 = uniform_f64((double)(H - 1));
    g.Wm1 = uniform_f64((double)(W - 1));
    g.Hm2 = uniform_f64((double)max(H - 2, 0));
    g.Wm2 = uniform_f64((double)max(W - 2, 0));
    g.W = W;
    g.rowstep = H > 1 ? 4u * (unsigned)W : 0u;
    g.single = W == 1;
    return g;
}

struct LeanTaps {
    unsigned off0;       // byte offset of the first pair; the second is off0 + rowstep
    double wy0, wy1, wx0, wx1;
    bool inside;
};

// 8-byte load from a 4-byte aligned address (x0 is arbitrary); gfx950 global memory
// accepts the unaligned dwordx2
struct __attribute__((packed, aligned(4))) PairF { float a, b; };

// load at a uniform base plus a 32-bit byte offset (one SGPR pair + one VGPR: no 64-bit
// address arithmetic on the vector ALU)
template <class T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_off(void *base, unsigned byte_off, T v)
{
    *reinterpret_cast<T *>(static_cast<char *>(base) + byte_off) = v;
}

// a wave-uniform pointer pinned to a scalar register pair: a load at `scalar_ptr(row) + lane offset` then takes the
// scalar-base form (SGPR pair + 32-bit VGPR offset) instead of 64-bit address arithmetic on the vector ALU
template <class T>
__device__ __forceinline__ const T *scalar_ptr(const T *p)
{
    typedef const T __attribute__((address_space(1))) *gptr;   // keep the global address space through the asm
    gptr g = (gptr)p;
    asm volatile("" : "+s"(g));
    return (const T *)g;
}

// Frame pixels are float32 (what the reference's callers hand over) or uint8 (its on-disk frame
// format, generate_test_suite.py:259-261): PIX selects the element type a kernel reads; the uint8 ->
// float32 conversion (optical_flow_verifier.py:61-65) is exact, so both give the same values.
template <class PIX>
__device__ __forceinline__ float ld_pix(const void *base, unsigned elem)
{
    return (float)*reinterpret_cast<const PIX *>(static_cast<const char *>(base) + (size_t)elem * sizeof(PIX));
}
// two horizontally adjacent pixels at element offset `elem` (any alignment)
struct __attribute__((packed)) PairB { unsigned char a, b; };
template <class PIX>
__device__ __forceinline__ PairF ld_pix_pair(const void *base, unsigned elem)
{
    if constexpr (sizeof(PIX) == 1) {
        const PairB q = *reinterpret_cast<const PairB *>(static_cast<const char *>(base) + elem);
        return PairF{(float)q.a, (float)q.b};
    } else {
        return ld_off<PairF>(base, elem * 4u);   // (two aligned 4-byte loads instead were measured: +6 %)
    }
}

// the sample point (y, x) itself; see lean_taps for how it is formed
__device__ __forceinline__ LeanTaps lean_taps_at(const LeanGeom &g, double y, double x)
{
    LeanTaps t;
    // 0 <= y <= H-1 as ONE unsigned compare of the bit patterns: non-negative doubles order like
    // their bits, and a set sign bit (y < 0; y = -0.0 cannot come out of the sum above) or a NaN
    // reads as larger than any in-range value.  Two compares and one scalar AND, no branches.
    const int in_y = (unsigned long long)__double_as_longlong(y) <= (unsigned long long)__double_as_longlong(g.Hm1);
    const int in_x = (unsigned long long)__double_as_longlong(x) <= (unsigned long long)__double_as_longlong(g.Wm1);
    t.inside = (in_y & in_x) != 0;
    const double fy = fmin(floor(y), g.Hm2), fx = fmin(floor(x), g.Wm2);
    const double ry = y - fy, rx = x - fx;
    t.wy0 = 1.0 - ry;
    t.wx0 = 1.0 - rx;
    t.wy1 = 1.0 - t.wy0;
    t.wx1 = 1.0 - t.wx0;
    const int y0 = (int)fy, x0 = (int)fx;      // in range whenever `inside`
    const unsigned cell = (unsigned)__mul24(y0, g.W) + (unsigned)x0;   // H, W < 2^24 (host check)
    t.off0 = t.inside ? cell * 4u : 0u;
    return t;
}

// The same sample with the weights formed as late as possible: between the gathers' issue and their use a cell
// holds its two fractions (4 registers) instead of four weights (8), so that all cells of a tile can have their
// gathers in flight together.  Same operations in the same order as lean_taps_at + lean_finish.
struct LeanFrac {
    unsigned off0;
    double ry, rx;
    bool inside;
};

__device__ __forceinline__ LeanFrac lean_frac_at(const LeanGeom &g, double y, double x)
{
    LeanFrac t;
    const int in_y = (unsigned long long)__double_as_longlong(y) <= (unsigned long long)__double_as_longlong(g.Hm1);
    const int in_x = (unsigned long long)__double_as_longlong(x) <= (unsigned long long)__double_as_longlong(g.Wm1);
    t.inside = (in_y & in_x) != 0;
    const double fy = fmin(floor(y), g.Hm2), fx = fmin(floor(x), g.Wm2);
    t.ry = y - fy;
    t.rx = x - fx;
    const int y0 = (int)fy, x0 = (int)fx;      // in range whenever `inside`
    const unsigned cell = (unsigned)__mul24(y0, g.W) + (unsigned)x0;   // H, W < 2^24 (host check)
    t.off0 = t.inside ? cell * 4u : 0u;
    return t;
}

__device__ __forceinline__ LeanTaps lean_taps(const LeanGeom &g, int gy, int gx, float u, float v)
{
    const double y = (double)gy + (double)v;   // int64 + float32 -> float64, as the reference
    const double x = (double)gx + (double)u;
    return lean_taps_at(g, y, x);
}

// (double)n for 0 <= n < 2^31 with integer operations only: for a wave-uniform n they all run on the
// scalar ALU and the result lives in an SGPR pair (int -> fp64 conversion only exists on the vector ALU,
// which is the fused iteration kernel's scarce resource)
__device__ __forceinline__ double uint_to_f64_bits(int n)
{
    const int e = 31 - __builtin_clz((unsigned)n | 1u);   // n = 0: e = 0, patched below
    const unsigned long long bits = ((unsigned long long)(1023 + e) << 52) + ((unsigned long long)(unsigned)n << (52 - e)) -
                                    (1ull << 52);
    return __longlong_as_double((long long)(n ? bits : 0ull));
}

// NARROW = false: the caller guarantees W >= 2 (and skips the one-column form)
template <bool NARROW, class PIX = float, class TAPS = LeanTaps>
__device__ __forceinline__ void lean_load(const LeanGeom &g, const void *__restrict__ img, const TAPS &t,
                                          PairF &r0, PairF &r1)
{
    // off0 / rowstep are byte offsets of float32 cells: element offsets are a quarter of them
    const unsigned e0 = t.off0 >> 2, e1 = (t.off0 + g.rowstep) >> 2;
    if (NARROW && g.single) {  // uniform; one-column image: the x+1 tap is the mirrored (same) element, weight 0
        r0.a = r0.b = ld_pix<PIX>(img, e0);
        r1.a = r1.b = ld_pix<PIX>(img, e1);
    } else {
        r0 = ld_pix_pair<PIX>(img, e0);
        r1 = ld_pix_pair<PIX>(img, e1);
    }
}

__device__ __forceinline__ float lean_finish(const LeanTaps &t, PairF r0, PairF r1)
{
    // SciPy starts the sum at +0.0; adding the first term to it only matters for the sign of an
    // all-zero result (-0.0 vs +0.0, equal as values), so the sum starts at the first term
    double acc, c;
    c = (double)r0.a; c = c * t.wy0; acc = c * t.wx0;
    c = (double)r0.b; c = c * t.wy0; c = c * t.wx1; acc = acc + c;
    c = (double)r1.a; c = c * t.wy1; c = c * t.wx0; acc = acc + c;
    c = (double)r1.b; c = c * t.wy1; c = c * t.wx1; acc = acc + c;
    return t.inside ? (float)acc : 0.0f;
}

__device__ __forceinline__ float lean_finish(const LeanFrac &t, PairF r0, PairF r1)
{
    const double wy0 = 1.0 - t.ry, wx0 = 1.0 - t.rx;
    const double wy1 = 1.0 - wy0, wx1 = 1.0 - wx0;
    double acc, c;
    c = (double)r0.a; c = c * wy0; acc = c * wx0;
    c = (double)r0.b; c = c * wy0; c = c * wx1; acc = acc + c;
    c = (double)r1.a; c = c * wy1; c = c * wx0; acc = acc + c;
    c = (double)r1.b; c = c * wy1; c = c * wx1; acc = acc + c;
    return t.inside ? (float)acc : 0.0f;
}

// np.linspace(0, S-1, T)[i]; `step` = (double)(S-1)/(double)(T-1) from the host
struct Linspace {
    double step;
    double last;  // S - 1
    int T;
};

__device__ __forceinline__ double linspace_at(const Linspace &l, int i)
{
    if (l.T <= 1) return 0.0;
    if (i == l.T - 1) return l.last;
    return (double)i * l.step;  // step == 0 (S == 1) also yields 0, as NumPy does
}

// Two correctly rounded float32 quotients nu / d, nv / d that share one reciprocal: v_rcp_f32 refined once, then per
// quotient the product, two remainder corrections and the final fused correction -- the compiler's own IEEE expansion
// (rcp, fma x2, mul, fma x4) minus its range scaling (v_div_scale / v_div_fmas / v_div_fixup), with the reciprocal's three
// instructions paid once: 13 instructions for two quotients instead of 22.  Equal to nu / d, nv / d bit for bit PROVIDED
// nothing leaves the normal range: 2^-64 < |d| < 2^64 and each numerator is 0 or 2^-40 <= |n| < 2^60 (then |q| and every
// remainder n - d q stay normal, so each fma's exact argument is representable as Markstein's argument needs; a zero
// numerator gives a zero of either sign, equal as a value).  Callers establish the range (k_lks on verified 8-bit frames).
__device__ __forceinline__ void div2_shared_rcp(float nu, float nv, float d, float &qu, float &qv)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r = __builtin_fmaf(e0, r0, r0);
    float q = nu * r;
    float e = __builtin_fmaf(-d, q, nu);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, nu);
    qu = __builtin_fmaf(e, r, q);
    q = nv * r;
    e = __builtin_fmaf(-d, q, nv);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, nv);
    qv = __builtin_fmaf(e, r, q);
}

#ifndef OFLK_PROBE_FASTDIV
#define OFLK_PROBE_FASTDIV 0   // diagnostic builds only: the shared-reciprocal quotients everywhere, unguarded (timing probe)
#endif
#if OFLK_PROBE_FASTDIV && !defined(OFLK_DIAG)
#error "OFLK_PROBE_FASTDIV is a diagnostic build: add -DOFLK_DIAG"
#endif

// 2x2 Cramer solve, every op individually rounded (lucas_kanade_core.py:122-133).  SAFE_RANGE: the caller guarantees
// div2_shared_rcp's range conditions, and the two divisions share their reciprocal (same values).
template <bool SAFE_RANGE = false>
__device__ __forceinline__ void lk_solve(float Sxx, float Syy, float Sxy, float Sxt, float Syt,
                                         float &u, float &v)
{
    float b0 = -Sxt, b1 = -Syt;
    float m0 = Sxx * Syy;
    float m1 = Sxy * Sxy;
    float det = m0 - m1;
    u = 0.0f;
    v = 0.0f;
    if (fabsf(det) > 1e-4f) {  // compared as float32(1e-4), lucas_kanade_core.py:131
        float n0 = Syy * b0, n1 = Sxy * b1;
        float n2 = Sxx * b1, n3 = Sxy * b0;
        float nu = n0 - n1, nv = n2 - n3;
        if constexpr (SAFE_RANGE || OFLK_PROBE_FASTDIV) {
            div2_shared_rcp(nu, nv, det, u, v);
        } else {
            u = nu / det;
            v = nv / det;
        }
    }
}

// ---------------------------------------------------------------------------
// K1/K7: fused Lucas-Kanade tile kernel.
//   MODE_SINGLE : lucas_kanade_single_scale(prev, curr)            -> u, v
//   MODE_ITER   : one pyramid iteration (lucas_kanade_pyramidal.py:203-214):
//                 warp(curr, flow) -> LK(prev, warped) -> flow_out = flow_in + d,
//                 per-block sums of |du|, |dv|
//   MODE_GRADS  : lucas_kanade_from_gradients(Ix, Iy, It)           -> u, v
// One 256-thread block produces a 64 x 24 output tile (k_lkw below).  HBM traffic per
// output pixel: 8 B in + 8 B out (SINGLE), 16 B in + 8 B out (ITER; the warp's
// gathers of `curr` hit L1/L2).
// ---------------------------------------------------------------------------
enum { MODE_SINGLE = 0, MODE_ITER = 1, MODE_GRADS = 2 };


constexpr int kMaxSegs = 40;

struct LkArgs {
    const float *prev;  // [B][H][W]   (MODE_GRADS: Ix)
    const float *curr;  // [B][H][W]   (MODE_GRADS: Iy)
    const float *aux;   // MODE_GRADS: It
    // ITER: the flow of a level lives in two ping-pong slots of INTERLEAVED float2 {u, v} [B][H][W]: one
    // 8-byte load per staging cell and one 16-byte load / store per output pair instead of twice as many
    // half as wide (vector-memory instructions are the kernel's scarce resource).
    // An iteration reads slot `sel` and writes slot 1 - sel, or, when `planar_out` is set (the last
    // iteration of the finest level), the caller's planar u / v planes.  SINGLE / GRADS write ou / ov.
    float2 *fl[2];
    float *ou, *ov;     // planar outputs [B][H][W]
    int planar_out;
    // ITER: residual accumulators of the whole call, [B][L][K][kAccShards][kAccStride] (see lk_report /
    // lk_level_state); this launch is iteration `iter` of level `level`
    unsigned long long *acc;
    int level, iter, L, K;
    unsigned long long conv_thr;   // early-exit threshold of the level as a total (lk_level_state)
    int H, W;
    int B;              // frame pairs in the launch (k_lkw decodes pair/tile from a 1-D grid)
    // vertical chaining: tile rows [seg_row[s], seg_row[s+1]) form segment s, walked by one
    // block; nseg = 0 means one tile per block (plain XCD tile order)
    int nseg;
    unsigned short seg_row[kMaxSegs + 1];
    // streaming kernel k_lks (oflk_stream.hpp): rows per segment, segments per strip
    int Hs, segs;
    // SINGLE 5x5 on integer-valued frames: redo list shared by k_lks and k_lkw, 32-bit words
    //   [0] entries in the list   [1] ticket of the redo pass   [2 .. 2+T) one flag per 64 x 24 tile, T = B * tiles_y * tiles_x
    //   [2+T .. 2+2T) the list: tile indices (b * tiles_y + tile_y) * tiles_x + tile_x
    // k_lks appends the tiles in which its order-free window sums may differ from NumPy's (flag = dedup); k_lkw launched
    // with redo_pass = 1 walks the list -- block i takes entries i, i + gridDim.x, ... -- clears the flags, and its last
    // block to finish resets [0] and [1]: the buffer is all zero between calls.  nullptr: no list.
    unsigned *redo;
    int redo_pass;
    // streaming kernel, first iteration of a level in the tolerant mode (k_lks<.., UPS = true>): the level's initial flow is
    // the coarser level's final flow upsampled on the fly (upsample_flow, lucas_kanade_pyramidal.py:100-138, in the
    // fused-lerp form) instead of planes written by k_upsample and read back: 16 B/px of traffic and a launch less
    const float2 *up_src;        // the coarser level's two flow slots [2][B][Hc][Wc], interleaved {u, v}
    size_t up_slot_stride;       // elements between the slots
    int up_level, up_iters;      // the coarser level's index; iterations launched per level (lk_level_state)
    unsigned long long up_thr;   // its convergence threshold
    int Hc, Wc;
    Linspace up_ly, up_lx;       // np.linspace(0, Hc-1, H), np.linspace(0, Wc-1, W)
    float up_sx, up_sy;          // float32(W / Wc), float32(H / Hc)
#ifdef OFLK_STAMPS
    unsigned *stamps;   // diagnostic build only: [block][wave][16] cycle sums per code section
#endif
};

// ---------------------------------------------------------------------------
// K1/K7: the fused kernel k_lkw<HW, MODE, VEC>; the 5x5 window (window_size 4 or 5) is the
// hot case and has a specialised sum stage.
//
// Tile 64 x 24 per 256-thread block (OFLK_NY = 3), each thread 2 (x) x 3 (y) outputs; large
// launches let one block walk several vertically adjacent tiles (see k_lkw).
//
// Exact window sums with fewer adds.  NumPy sums the 25 products a[0..24]
// (row-major) as r[j] = (a[j] + a[j+8]) + a[j+16], j = 0..7, then
// ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) + a[24].  With a[5*row+col] =
// P[y-2+row][x-2+col] several r's of neighbouring windows are THE SAME three
// products added in THE SAME order, hence bit-identical:
//     r1(x,y) = r0(x+1,y)   r3(x,y) = r2(x+1,y)   r5(x,y) = r0(x,y+1)
//     r6(x,y) = r0(x+1,y+1) r7(x,y) = r2(x,y+1)
// so a 2x3 output patch needs 12 r0's, 12 r2's and 6 r4's (60 adds) instead of
// 48 r's (96 adds); tree and tail are per output.  No rounding is changed.
//
// LDS layout: products interleaved as float2 {Ix*Ix, Iy*Iy}, float2 {Ix*Iy, Ix*It}
// and float {Iy*It}, so the sums of two planes ride one v_pk_add_f32.  The
// frame-average / It staging tiles alias the product planes (gradients wait in
// registers across the barrier).
// ---------------------------------------------------------------------------
// build-time tuning knobs (defaults are the measured best on MI355X)
#ifndef OFLK_NY
#define OFLK_NY 3      // output rows per thread: tile = 64 x 8*NY
#endif
#ifndef OFLK_BATCH
#define OFLK_BATCH 7   // warp cells whose gathers are in flight together (ITER stage 1, 5x5 window)
#endif
// XCD-aware tile order (speed only, never correctness).  Workgroups of a 1-D grid are
// dealt round-robin over the 8 XCDs (ids i and i+8 share an XCD, each with its own L2).
// Giving XCD x the contiguous tile range [x*chunk, (x+1)*chunk) makes x- and
// y-adjacent tiles, which share halo cache lines, meet in one L2 at about the same time.
__device__ __forceinline__ int xcd_tile_index(int bid, int ntiles)
{
    constexpr int NXCD = 8;
    const int chunk = (ntiles + NXCD - 1) / NXCD;
    const int full = ntiles - (NXCD - 1) * chunk;   // tiles in the last XCD's range (may be < chunk)
    // ids below NXCD*full interleave all 8 ranges; the rest only the first 7
    if (bid < NXCD * full || full < 0) {
        if (full < 0) return bid;                   // tiny grids: plain order
        return (bid % NXCD) * chunk + bid / NXCD;
    }
    const int r = bid - NXCD * full;
    return (r % (NXCD - 1)) * chunk + full + r / (NXCD - 1);
}

// Sum over the 64 lanes of a wave in a fixed order, result in lane 63 (other lanes hold
// partial sums): row_shr 1/2/4/8 inside each 16-lane row, then row_bcast 15 and 31.
__device__ __forceinline__ float wave_sum_to_lane63(float x)
{
    auto dpp = [](float v, auto ctrl, auto row_mask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                    decltype(row_mask)::value, 0xf, true));
    };
    x += dpp(x, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
    x += dpp(x, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
    x += dpp(x, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
    x += dpp(x, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
    x += dpp(x, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 -> rows 1, 3
    x += dpp(x, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 -> rows 2, 3
    return x;
}

constexpr int k5NY = OFLK_NY;      // output rows per thread
constexpr int k5TX = 64;
constexpr int k5TY = 8 * k5NY;     // 8 thread rows x NY

// Optimisation fence on a value: the compiler must have it computed here and may not
// sink its computation past later fences (keeps the row-streaming order, and with it
// the register footprint, of the window-sum code).  No instruction is emitted.
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float2 &x)
{
    double d = __builtin_bit_cast(double, x);
    asm volatile("" : "+v"(d));
    x = __builtin_bit_cast(float2, d);
}

// Two planes summed side by side.  OFLK_PK_SUMS = 1 adds them with one v_pk_add_f32 (float2);
// 0 (default) with two v_add_f32: on gfx950 a packed fp32 add occupies the vector ALU 2.7x as long
// as a scalar one (tools/ubench/valu_cycles.hip), so two scalar adds are the cheaper pair.
#ifndef OFLK_PK_SUMS
#define OFLK_PK_SUMS 0
#endif
struct F2 {
    float x, y;
};
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return F2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ void pin(F2 &v)
{
    asm volatile("" : "+v"(v.x));
    asm volatile("" : "+v"(v.y));
}
#if OFLK_PK_SUMS
using Sum2 = float2;
__device__ __forceinline__ Sum2 sum2(float a, float b) { return make_float2(a, b); }
#else
using Sum2 = F2;
__device__ __forceinline__ Sum2 sum2(float a, float b) { return F2{a, b}; }
#endif

template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ F2 zero_of<F2>() { return F2{0.0f, 0.0f}; }
template <> __device__ __forceinline__ float zero_of<float>() { return 0.0f; }
template <> __device__ __forceinline__ float2 zero_of<float2>() { return make_float2(0.0f, 0.0f); }

// Window sums of a 2 (x) by NY (y) output patch, streaming the NY+4 product rows
// top to bottom: at most three rows and three output rows' partial r's are live.
// `load_row(i, row)` fills row[0..5] with products P[y0-2+i][x0-2 .. x0+3].
template <typename T, int NY, typename LoadRow>
__device__ __forceinline__ void patch5_sums(LoadRow load_row, T (&out)[NY][2])
{
    T R0[NY + 1][3], R2[NY + 1][3], R4[NY][2];
    T w[NY + 4][6];
    load_row(0, w[0]);
#pragma unroll
    for (int i = 0; i < NY + 4; i++) {
        // one row is prefetched ahead of the arithmetic; the fence keeps the compiler
        // from hoisting every LDS read to the top (which would need ~250 VGPRs)
        if (i + 1 < NY + 4) load_row(i + 1, w[i + 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (i >= 1 && i - 1 <= NY) {
            const int oy = i - 1;  // r0 family, first add: a[0] + a[8]
#pragma unroll
            for (int x = 0; x < 3; x++) R0[oy][x] = w[i - 1][x] + w[i][x + 3];
        }
        if (i >= 2 && i - 2 <= NY) {
            const int oy = i - 2;  // r2 and r4 families, first add: a[2] + a[10], a[4] + a[12]
#pragma unroll
            for (int x = 0; x < 3; x++)
                if (oy < NY || x < 2) R2[oy][x] = w[i - 2][x + 2] + w[i][x];
            if (oy < NY) {
#pragma unroll
                for (int x = 0; x < 2; x++) R4[oy][x] = w[i - 2][x + 4] + w[i][x + 2];
            }
        }
        if (i >= 3 && i - 3 <= NY) {
            const int oy = i - 3;  // second add: + a[16], + a[18]
#pragma unroll
            for (int x = 0; x < 3; x++) {
                R0[oy][x] = R0[oy][x] + w[i][x + 1];
                if (oy < NY || x < 2) R2[oy][x] = R2[oy][x] + w[i][x + 3];
            }
        }
        if (i >= 4) {
            const int oy = i - 4;  // + a[20]; the window of output row oy is complete
#pragma unroll
            for (int x = 0; x < 2; x++) {
                R4[oy][x] = R4[oy][x] + w[i][x];
                T lo = (R0[oy][x] + R0[oy][x + 1]) + (R2[oy][x] + R2[oy][x + 1]);
                T hi = (R4[oy][x] + R0[oy + 1][x]) + (R0[oy + 1][x + 1] + R2[oy + 1][x]);
                T res = lo + hi;
                res = res + w[i][x + 4];           // a[24]
                out[oy][x] = zero_of<T>() + res;   // np.sum starts from the identity 0
                pin(out[oy][x]);
            }
        }
    }
}

// Window sums of NX adjacent outputs of ONE output row for any window with
// (2HW+1)^2 <= 128, in NumPy's pairwise order (r[t % 8] += a[t] for t < N - N % 8, the
// fixed tree, then the tail).  `load_row(i, row)` fills row[0 .. NX+2HW-1] with products
// P[y-HW+i][x0-HW .. x0+NX-1+HW].  Used for the 3x3 and 7x7 windows (the 5x5 window has
// the cheaper shared-r form above).
template <typename T, int HW, int NX, typename LoadRow>
__device__ __forceinline__ void window_sums_row(LoadRow load_row, T (&out)[NX])
{
    constexpr int S = 2 * HW + 1, N = S * S, NB = N - (N % 8);
    static_assert(N >= 8 && N <= 128, "window must fit NumPy's unrolled pairwise block");
    T r[NX][8];
    T res[NX];
#pragma unroll
    for (int i = 0; i < S; i++) {
        T row[NX + 2 * HW];
        load_row(i, row);
#pragma unroll
        for (int c = 0; c < S; c++) {
            const int t = S * i + c;
#pragma unroll
            for (int x = 0; x < NX; x++) {
                if (t < 8) {
                    r[x][t] = row[x + c];
                } else if (t < NB) {
                    r[x][t % 8] = r[x][t % 8] + row[x + c];
                } else {
                    if (t == NB)
                        res[x] = ((r[x][0] + r[x][1]) + (r[x][2] + r[x][3])) + ((r[x][4] + r[x][5]) + (r[x][6] + r[x][7]));
                    res[x] = res[x] + row[x + c];
                }
            }
        }
    }
#pragma unroll
    for (int x = 0; x < NX; x++) {
        out[x] = zero_of<T>() + res[x];  // np.sum starts from the identity 0
        pin(out[x]);
    }
}

// Stage 1 works on groups of four horizontally adjacent cells: the staging tiles are
// 38 rows x 72 columns starting at (y0-3, x0-4), so with W % 4 == 0 every group is one
// aligned 16-byte load per plane (a group lies entirely inside or entirely outside the
// image).  Other widths take the VEC = false instantiation (cell-by-cell loads).
// staging tile geometry: columns start at x0 - SX (a multiple of 4 >= the halo HW+1)
template <int HW> struct LkGeom {
    static constexpr int SX = (HW + 1 <= 4) ? 4 : 8;   // staging column of image column x0
    static constexpr int GW = (k5TX + 2 * SX) / 4;     // 4-cell groups per staging row (18 or 20)
    static constexpr int AS = 4 * GW;                  // staging columns (72 or 80)
};

// ---------------------------------------------------------------------------
// K6: per-pair residual means, convergence and ping-pong bookkeeping
// (lucas_kanade_pyramidal.py:213-223) without a kernel of its own.
//
// Every block of an iteration launch adds its |du|, |dv| sums, as 64-bit fixed point, to
// the accumulators of (pair, level, iteration) with two fire-and-forget device-scope atomics
// (integer adds commute: the totals do not depend on the order in which blocks finish).  Global
// atomics execute at the memory side and queue per line: with every block of a pair adding into
// one 128-byte line a 1080p launch spent 9 us draining them, so each accumulator has eight
// shards, picked by block id, 512 bytes apart (1.3 us).  Nothing in the same launch reads them.  Later launches on the stream -- the next iteration, the
// flow upsample, the export at the end of the call -- see the final totals and each block
// re-derives from them, with identical arithmetic, what a finalize step would have stored:
// the means, how many iterations of the level were executed before the early exit
// (:221-223) and hence which ping-pong slot holds the current flow.
//
// The reference's np.mean is an fp32 pairwise sum; the two agree to ~1e-7 relative, which can
// flip the "< 0.01" test only when the mean sits within that distance of the threshold
// (DESIGN.md "Known deviations").
// ---------------------------------------------------------------------------
constexpr int kAccShards = 8;
constexpr int kAccStride = 64;   // u64 words between shards (512 B: every shard on a line, and likely a channel, of its own)
constexpr double kAccScale = 1048576.0;       // 2^20 steps per pixel of |d|
constexpr double kAccBlockMax = 268435456.0;  // 2^28: a block's sum is clamped here (also catches NaN)

__device__ __forceinline__ size_t acc_index(int b, int l, int k, int L, int K)
{
    return ((((size_t)b * L + l) * K + k) * kAccShards) * kAccStride;
}

// one thread per block, after the block's last tile
__device__ __forceinline__ void lk_report(const LkArgs &a, int b, double su, double sv)
{
    su = su < kAccBlockMax ? su : kAccBlockMax;
    sv = sv < kAccBlockMax ? sv : kAccBlockMax;
    unsigned long long *slot = a.acc + acc_index(b, a.level, a.iter, a.L, a.K) + kAccStride * (blockIdx.x & (kAccShards - 1));
    __hip_atomic_fetch_add(slot + 0, (unsigned long long)__double2ll_rn(su * kAccScale), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(slot + 1, (unsigned long long)__double2ll_rn(sv * kAccScale), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// totals of iteration k (a finished launch)
__device__ __forceinline__ void lk_totals(const unsigned long long *acc, int b, int l, int k, int L, int K,
                                          unsigned long long &tu, unsigned long long &tv)
{
    const unsigned long long *s = acc + acc_index(b, l, k, L, K);
    tu = 0;
    tv = 0;
#pragma unroll
    for (int i = 0; i < kAccShards; i++) {
        tu += s[kAccStride * i];
        tv += s[kAccStride * i + 1];
    }
}

// mean|du| (or |dv|) from a total: what np.mean(np.abs(d)) is compared and logged as
__host__ __device__ inline float lk_mean_of(unsigned long long total, double count)
{
    return (float)(((double)total / kAccScale) / count);
}

struct LevelState {
    int executed;   // iterations of the level run so far (the flow sits in slot executed & 1)
    bool done;      // the early exit has fired
};

// State of (pair b, level l) before iteration k; k = K gives the level's final state.
// `thr` is the smallest total whose mean is not below the exit threshold
// (lk_mean_of(thr, count) >= 0.01f, found on the host with the same arithmetic), so the
// test "mean < 0.01" (:221-223) is an integer compare here and costs every block a few
// scalar operations instead of two fp64 divisions per thread.
__device__ __forceinline__ LevelState lk_level_state(const unsigned long long *acc, int b, int l, int k, int L,
                                                     int K, unsigned long long thr)
{
    LevelState st{0, false};
    for (int j = 0; j < k && !st.done; j++) {
        unsigned long long tu, tv;
        lk_totals(acc, b, l, j, L, K, tu, tv);
        st.executed = j + 1;
        st.done = tu < thr && tv < thr;
    }
    return st;
}

// Start of a pyramidal call: zero the per-call state block and the coarsest level's flow
// planes (lucas_kanade_pyramidal.py:182-184) in one launch.
__global__ __launch_bounds__(256) void k_call_init(unsigned *__restrict__ state, size_t nwords,
                                                   float *__restrict__ u0, float *__restrict__ v0, size_t n)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    for (size_t i = t; i < nwords; i += nt) state[i] = 0u;
    for (size_t i = t; i < n; i += nt) {
        u0[i] = 0.0f;
        v0[i] = 0.0f;
    }
}

// Levels narrower or shorter than the window: no pixel has a full window, the reference's
// loop (lucas_kanade_core.py:101-108) runs over nothing and d = 0 everywhere.  Launched in
// place of k_lkw so that the tile kernel may assume H, W > 2*HW (in particular W >= 2).
// grid: (ceil(H*W / 256), B)
template <int MODE>
__global__ __launch_bounds__(256) void k_lk_degenerate(LkArgs a)
{
    const int b = blockIdx.y;
    const size_t plane = (size_t)a.H * (size_t)a.W;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    int sel = 0;
    if (MODE == MODE_ITER) {
        const LevelState st = lk_level_state(a.acc, b, a.level, a.iter, a.L, a.K, a.conv_thr);
        if (st.done) return;
        sel = st.executed & 1;
    }
    if (e < plane) {
        const size_t i = (size_t)b * plane + e;
        if (MODE == MODE_ITER) {
            const float2 f = a.fl[sel][i];
            const float2 g = make_float2(f.x + 0.0f, f.y + 0.0f);   // flow += d
            if (a.planar_out) {
                a.ou[i] = g.x;
                a.ov[i] = g.y;
            } else {
                a.fl[1 - sel][i] = g;
            }
        } else {
            a.ou[i] = 0.0f;
            a.ov[i] = 0.0f;
        }
    }
    // d = 0: nothing to add to the accumulators; the iteration reads as converged
}

// ---------------------------------------------------------------------------
// Windows outside 3x3 ... 11x11 (the reference takes any window_size, lucas_kanade_core.py:104-119): one thread per
// output pixel walks its (2hw+1)^2 window and forms the five sums in np.sum's pairwise order for ANY length
// (numpy/_core/src/umath/loops_utils.h.src: fewer than 8 values one after the other; up to 128 values in eight
// interleaved accumulators, the fixed tree, then the tail; beyond that two halves, the first a multiple of 8 long,
// each summed the same way -- a 13x13 window is 169 products: 80 + 89).  Exact and slow: the tiled kernel's window
// sums live in registers because their shape is known at compile time; this one is the catch-all.
// SINGLE evaluates the gradients of every window element from the frames as k_gradients does (same operations).
// ---------------------------------------------------------------------------
struct Five {
    float xx, yy, xy, xt, yt;
};
__device__ __forceinline__ Five operator+(Five a, Five b)
{
    return Five{a.xx + b.xx, a.yy + b.yy, a.xy + b.xy, a.xt + b.xt, a.yt + b.yt};
}

template <class ELEM>
__device__ __forceinline__ Five np_pairwise_leaf(ELEM &elem, int lo, int n, int side)   // n <= 128
{
    int r = lo / side, c = lo - r * side;
    auto next = [&]() {
        const Five e = elem(r, c);
        if (++c == side) {
            c = 0;
            r++;
        }
        return e;
    };
    if (n < 8) {
        Five res{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < n; i++) res = res + next();
        return res;
    }
    Five acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = next();
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = acc[j] + next();
    }
    Five res = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (; i < n; i++) res = res + next();
    return res;
}

// np.sum halves unevenly (the first half is cut down to a multiple of 8), so d levels of halving cover a little less than
// 128 << d values: 1 808 at d = 4, 3 600 at d = 5.  The host admits windows of up to 45 x 45 = 2 025 products and checks
// the depth of each (pairwise_depth).
constexpr int kGenericDepth = 5;
__host__ __device__ constexpr int pairwise_depth(int n)
{
    if (n <= 128) return 0;
    const int n2 = n / 2 - (n / 2) % 8;
    const int a = pairwise_depth(n2), b = pairwise_depth(n - n2);
    return 1 + (a > b ? a : b);
}
static_assert(pairwise_depth(45 * 45) <= kGenericDepth && pairwise_depth(169) == 1, "block recursion depth of np.sum");

// np.sum's recursion (two halves, the first a multiple of 8 long, down to blocks of at most 128) walked with an explicit
// stack, so that the block sum exists once in the code: a frame is (range, stage, the left half's sum).
template <class ELEM>
__device__ __forceinline__ Five np_pairwise(ELEM &elem, int n, int side)
{
    int f_lo[kGenericDepth + 1], f_n[kGenericDepth + 1], f_stage[kGenericDepth + 1];
    Five f_left[kGenericDepth + 1];
    int sp = 0;
    f_lo[0] = 0;
    f_n[0] = n;
    f_stage[0] = 0;
    Five ret{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    while (sp >= 0) {
        const int lo = f_lo[sp], len = f_n[sp];
        if (len <= 128 || sp == kGenericDepth) {
            ret = np_pairwise_leaf(elem, lo, len, side);
            sp--;
            continue;
        }
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (f_stage[sp] == 0) {          // descend into the left half
            f_stage[sp] = 1;
            f_lo[sp + 1] = lo;
            f_n[sp + 1] = n2;
            f_stage[sp + 1] = 0;
            sp++;
        } else if (f_stage[sp] == 1) {   // left half done: keep it, descend into the right half
            f_left[sp] = ret;
            f_stage[sp] = 2;
            f_lo[sp + 1] = lo + n2;
            f_n[sp + 1] = len - n2;
            f_stage[sp + 1] = 0;
            sp++;
        } else {                         // both done
            ret = f_left[sp] + ret;
            sp--;
        }
    }
    return ret;
}

// grid: (ceil(W / 64), ceil(H / 4), B)
template <int MODE, class PIX = float>
__global__ __launch_bounds__(256) void k_lk_generic(LkArgs a, int hw)
{
    static_assert(MODE == MODE_SINGLE || MODE == MODE_GRADS, "the iteration is run unfused for these windows");
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    if (x >= W || y >= H) return;
    const size_t base = (size_t)blockIdx.z * (size_t)H * (size_t)W;
    const size_t o = base + (size_t)y * W + x;
    float u = 0.0f, v = 0.0f;
    if (y >= hw && y < H - hw && x >= hw && x < W - hw) {   // borders stay 0 (lucas_kanade_core.py:101-108)
        const int side = 2 * hw + 1;
        auto elem = [&](int r, int c) -> Five {
            const int gy = y - hw + r, gx = x - hw + c;
            float ix, iy, it;
            if constexpr (MODE == MODE_GRADS) {
                const size_t i = base + (size_t)gy * W + gx;
                ix = a.prev[i];
                iy = a.curr[i];
                it = a.aux[i];
            } else {
                const PIX *__restrict__ prev = reinterpret_cast<const PIX *>(a.prev);
                const PIX *__restrict__ curr = reinterpret_cast<const PIX *>(a.curr);
                auto avg = [&](int yy, int xx) -> float {   // (prev + curr) / 2 with the "symm" ring, as k_gradients
                    yy = min(max(yy, 0), H - 1);
                    xx = min(max(xx, 0), W - 1);
                    const size_t i = base + (size_t)yy * W + xx;
                    const float s = (float)prev[i] + (float)curr[i];
                    return s * 0.5f;
                };
                const float a_mm = avg(gy - 1, gx - 1), a_m0 = avg(gy - 1, gx), a_mp = avg(gy - 1, gx + 1);
                const float a_0m = avg(gy, gx - 1), a_0p = avg(gy, gx + 1);
                const float a_pm = avg(gy + 1, gx - 1), a_p0 = avg(gy + 1, gx), a_pp = avg(gy + 1, gx + 1);
                ix = a_pp * -0.125f;
                ix = fmaf(a_pm, 0.125f, ix);
                ix = fmaf(a_0p, -0.25f, ix);
                ix = fmaf(a_0m, 0.25f, ix);
                ix = fmaf(a_mp, -0.125f, ix);
                ix = fmaf(a_mm, 0.125f, ix);
                iy = a_pp * -0.125f;
                iy = fmaf(a_p0, -0.25f, iy);
                iy = fmaf(a_pm, -0.125f, iy);
                iy = fmaf(a_mp, 0.125f, iy);
                iy = fmaf(a_m0, 0.25f, iy);
                iy = fmaf(a_mm, 0.125f, iy);
                const size_t i = base + (size_t)gy * W + gx;
                it = (float)prev[i] - (float)curr[i];
            }
            return Five{ix * ix, iy * iy, ix * iy, ix * it, iy * it};
        };
        const Five s = np_pairwise(elem, side * side, side);
        // np.sum starts from the identity 0
        lk_solve(0.0f + s.xx, 0.0f + s.yy, 0.0f + s.xy, 0.0f + s.xt, 0.0f + s.yt, u, v);
    }
    a.ou[o] = u;
    a.ov[o] = v;
}

// does k_lkw<HW, MODE> support walking several tiles per block (vertical chaining)?
template <int HW, int MODE>
constexpr bool kLkChain = HW <= 2 && MODE != MODE_GRADS;

// In-kernel timeline (diagnostic build -DOFLK_STAMPS only; tools/stamps.py): every wave keeps the
// s_memtime value of each stamp of each of its (up to 8) tiles in LDS and copies them out once.
// The values go to a buffer of their own; no output is computed from them.
#if defined(OFLK_STAMPS) && !defined(OFLK_DIAG)
#error "OFLK_STAMPS is a diagnostic build: add -DOFLK_DIAG"
#endif
#ifdef OFLK_STAMPS
#ifndef OFLK_STAMP_MASK
#define OFLK_STAMP_MASK 0xffff
#endif
#define OFLK_STAMP(i)                                                                              \
    if ((OFLK_STAMP_MASK >> (i)) & 1) do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        s_st[st_wave][st_tile & 7][i] = (unsigned)t_;                                              \
    } while (0)
#define OFLK_STAMP_OCC
#else
#define OFLK_STAMP(i) do { } while (0)
#define OFLK_STAMP_OCC
#endif


// Sensitivity probes (diagnostic builds only, -DOFLK_DIAG -DOFLK_PROBE=(site << 8 | kind); tools/abn.sh): extra
// instructions of one kind at one place of the tile loop, to read off what an instruction of that kind costs the
// launch.  site 1 = stage 1 after the coalesced loads are issued, 3 = stage 3 after the window sums; kind 1 = 96
// v_add_f32, 2 = 96 v_add_f64, 3 = 96 s_mov_b32, 4 = 24 dependent ds_read_b32.  Results stay correct.
#if defined(OFLK_DIAG) && defined(OFLK_PROBE)
#define OFLK_PROBE_SITE_KIND OFLK_PROBE
#else
#define OFLK_PROBE_SITE_KIND 0
#endif
template <int SITE>
__device__ __forceinline__ void probe(const float *lds)
{
    if constexpr ((OFLK_PROBE_SITE_KIND >> 8) == SITE) {
        constexpr int KIND = OFLK_PROBE_SITE_KIND & 255;
        if constexpr (KIND == 1) {
            float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
            for (int i = 0; i < 12; i++)
                asm volatile("v_add_f32 %0, %0, %0\nv_add_f32 %1, %1, %1\nv_add_f32 %2, %2, %2\nv_add_f32 %3, %3, %3\n"
                             "v_add_f32 %4, %4, %4\nv_add_f32 %5, %5, %5\nv_add_f32 %6, %6, %6\nv_add_f32 %7, %7, %7\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
        } else if constexpr (KIND == 2) {
            double r[4] = {1, 2, 3, 4};
#pragma unroll
            for (int i = 0; i < 24; i++)
                asm volatile("v_add_f64 %0, %0, %0\nv_add_f64 %1, %1, %1\nv_add_f64 %2, %2, %2\nv_add_f64 %3, %3, %3\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
        } else if constexpr (KIND == 3) {
            int r[4] = {1, 2, 3, 4};
#pragma unroll
            for (int i = 0; i < 24; i++)
                asm volatile("s_mov_b32 %0, %1\ns_mov_b32 %1, %2\ns_mov_b32 %2, %3\ns_mov_b32 %3, %0\n"
                             : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]));
        } else if constexpr (KIND == 4) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 24; i++) acc += *(const volatile float *)(lds + threadIdx.x + 64 * i);
            asm volatile("" ::"v"(acc));
        }
    }
}

// REDOL: the instantiation of a window without vertical chaining (7x7) that can walk the streaming kernel's redo list
// (the 5x5 kernel walks it with its chaining loop); a separate instantiation, so that the ordinary 7x7 kernel keeps its registers
template <int HW, int MODE, bool VEC, class PIX = float, bool REDOL = false>
__global__ __launch_bounds__(256) void k_lkw(LkArgs a)
{
    static_assert(MODE != MODE_GRADS || sizeof(PIX) == 4, "gradient planes are float32");
    static_assert(HW >= 1 && HW <= 5, "windows up to 11x11 (NumPy's single pairwise block)");
    constexpr int R = HW + 1;                              // halo of the frame-average tile
    constexpr int SX = LkGeom<HW>::SX, k5GW = LkGeom<HW>::GW;
    constexpr int AH = k5TY + 2 * R;                       // staging rows (y0-R ..)
    constexpr int AS = LkGeom<HW>::AS;                     // staging columns (x0-SX ..)
    constexpr int PH = k5TY + 2 * HW, PW = k5TX + 2 * HW;  // product tile at (y0-HW, x0-HW)
    constexpr int NG = (PH * PW + 255) / 256;              // gradient pixels per thread (GRADS: linear cell index)
    constexpr int RPW2 = (PH + 3) / 4;                     // stage 2: gradient rows per wave (lane = column)
    constexpr int HC2 = 2 * HW;                            // halo columns of the gradient tile
    constexpr int NHP2 = (HC2 * RPW2 + 63) / 64;           // halo cells per thread
    constexpr int NGA = MODE == MODE_GRADS ? NG : RPW2 + NHP2;   // gradient cells a thread holds
    constexpr int NGRP = AH * k5GW;                        // groups of 4 staging cells
    constexpr int NV = (NGRP + 255) / 256;                 // groups per thread
    constexpr int GC = SX - HW;                            // staging column of gradient column 0
    constexpr int NCARRY = 2 * R * AS;                     // staging cells shared with the tile below
    constexpr int NC = (NCARRY + 255) / 256;               // carried cells per thread

    // one LDS block: [PA float2 | PB float2 | PC float]; avg and It alias its start
    __shared__ __attribute__((aligned(16))) float s_mem[PH * PW * 5];
    __shared__ double s_red[2][4];
    float2 *s_pa = reinterpret_cast<float2 *>(s_mem);
    float2 *s_pb = reinterpret_cast<float2 *>(s_mem + PH * PW * 2);
    float *s_pc = s_mem + PH * PW * 4;
    float *s_avg = s_mem;              // AH*AS floats
    float *s_it = s_mem + AH * AS;     // another AH*AS floats
    static_assert(2 * AH * AS <= PH * PW * 5, "staging tiles must fit the product planes");

    // A block walks a segment of vertically adjacent tiles of one 64-column strip.  The 2R staging
    // rows (frame average, It) that a tile shares with the tile below it are carried over in
    // registers instead of being recomputed: the exact fp64 warp, the dominant cost, runs on
    // 24 new rows per tile instead of 30.
#ifdef OFLK_STAMPS
    // block lifetime on the chip-wide 100 MHz clock: entry, tile loop start / end, exit; plus where it ran
    unsigned bt[4];
    bt[0] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    const int H = a.H, W = a.W;
    const int tiles_x = (W + k5TX - 1) / k5TX, tiles_y = (H + k5TY - 1) / k5TY;
    // windows above 5x5 run one tile per block: their sum stage needs the registers the loop
    // and the carried rows would take (7x7: 153 -> 172 VGPRs, 3 -> 2 waves per SIMD)
    constexpr bool CHAIN = kLkChain<HW, MODE>;
    // the redo pass of k_lks (SINGLE, 5x5): the block walks its share of the list of flagged tiles, one tile per trip
    constexpr bool CAN_REDO = MODE == MODE_SINGLE && (HW == 2 || REDOL);
    constexpr bool LOOPS = CHAIN || CAN_REDO;   // the tile loop below runs more than once
    const bool redo = CAN_REDO && a.redo_pass != 0;   // uniform
    int b, tile_x, tile_y_first, ntile;
    if (redo) {
        const unsigned count = a.redo[0];
        ntile = blockIdx.x < count ? (int)((count - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
        b = 0;
        tile_x = 0;
        tile_y_first = 0;
    } else if (!CHAIN || a.nseg == 0) {
        const int tile = xcd_tile_index(blockIdx.x, tiles_x * tiles_y * a.B);
        b = tile / (tiles_x * tiles_y);
        const int t = tile - b * (tiles_x * tiles_y);
        tile_y_first = t / tiles_x;
        tile_x = t - tile_y_first * tiles_x;
        ntile = 1;
    } else {
        // XCD x (block ids x, x+8, ...) owns the column strips [x*S/8, (x+1)*S/8) and runs
        // their segments longest first (seg_row is built that way), so the blocks still in
        // flight when the grid drains are the short ones
        const int S = a.B * tiles_x;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int s_lo = (int)((long)xcd * S / 8), n = (int)((long)(xcd + 1) * S / 8) - s_lo;
        if (n == 0) return;
        const int seg = j / n;
        if (seg >= a.nseg) return;
        const int strip = s_lo + (j - seg * n);
        b = strip / tiles_x;
        tile_x = strip - b * tiles_x;
        tile_y_first = a.seg_row[seg];
        ntile = a.seg_row[seg + 1] - tile_y_first;
    }
    int sel = 0;
    if (MODE == MODE_ITER) {
        const LevelState st = lk_level_state(a.acc, b, a.level, a.iter, a.L, a.K, a.conv_thr);
        if (st.done) return;
        sel = st.executed & 1;
    }
    const size_t plane = (size_t)H * (size_t)W;
    // frame planes of pair b; PIX-sized elements (a.prev / a.curr are typed float for the common case)
    const PIX *__restrict__ prev = reinterpret_cast<const PIX *>(a.prev) + (size_t)b * plane;   // (re-pointed per trip by a redo pass)
    const PIX *__restrict__ curr = reinterpret_cast<const PIX *>(a.curr) + (size_t)b * plane;
    int x0 = tile_x * k5TX;
    float carry_a[NC], carry_i[NC];
    double blk_u = 0.0, blk_v = 0.0;   // thread 0: |d| sums of the block's tiles

    // ---- ITER stage 1 geometry: lane = image column, wave = staging row -------------------------------------
    // Wave w fills rows rs + w, rs + w + 4, ... of the 64 columns x0 .. x0+63 (lane l = column x0 + l): the row, its
    // image row gy, (double)gy and the row's base addresses are wave-uniform (scalar ALU), the column and (double)gx
    // are fixed per lane, the LDS address is a per-thread base plus a constant -- a cell costs the vector ALU its fp64
    // sampling arithmetic and little else.  The 2R halo columns of a wave's rows are one more cell for
    // 2R * (rows per wave) of its lanes, by the generic per-cell arithmetic.
    // (Issuing a continuing tile's coalesced loads from the tile before it -- between its window sums and its solve, so
    // that they travel while it divides and stores -- was built and measured: stage 1 of a tile got 1 700 wave-cycles
    // shorter and the launch not at all)
    constexpr int ST1_HC = 2 * R;                                     // halo cells per staging row
    // staging row of main cell k of wave wv (wave-uniform); the last row group may run past the tile: those waves
    // redo the last row (same values to the same LDS cells)
    auto st1_main_row = [](auto rs, int wv, int k) {
        constexpr int rstart = decltype(rs)::value;
        return (rstart + 4 * k + 3 < AH) ? rstart + wv + 4 * k : min(rstart + wv + 4 * k, AH - 1);
    };
    // halo cell h of a thread: staging row and cell column (0 .. R-1, 64+R .. 64+2R-1)
    auto st1_halo_pos = [](auto rs, int wv, int lane, int h, int &rr, int &c) {
        constexpr int rstart = decltype(rs)::value;
        constexpr int RPW = (AH - rstart + 3) / 4;
        const int idx = min(lane + 64 * h, ST1_HC * RPW - 1);
        const int j = idx / ST1_HC, hc = idx - j * ST1_HC;
        rr = min(rstart + wv + 4 * j, AH - 1);
        c = hc < R ? hc : hc + 64;
    };
    // all coalesced loads of a thread's stage-1 cells for the tile at image row y0t
    auto st1_loads = [&](auto rs, int tid, int y0t, auto &P, auto &F) {
        constexpr int rstart = decltype(rs)::value;
        constexpr int RPW = (AH - rstart + 3) / 4, NHP = (ST1_HC * RPW + 63) / 64;
        const float2 *__restrict__ fl_in = a.fl[sel] + (size_t)b * plane;
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int Hm1 = H - 1, Wm1 = W - 1;
        const int gxm = min(x0 + lane, Wm1);             // "symm" ring; farther cells are never used
        const unsigned gx_pix = (unsigned)gxm * (unsigned)sizeof(PIX), gx_fl = (unsigned)gxm * 8u;   // byte offsets in a row
#pragma unroll
        for (int k = 0; k < RPW + NHP; k++) {
            if (k < RPW) {
                // scalar row base + the lane's fixed column offset: no vector address arithmetic
                const unsigned rowe = (unsigned)(min(max(y0t - R + st1_main_row(rs, wv, k), 0), Hm1) * W);
                P[k] = (float)ld_off<PIX>(scalar_ptr(prev + rowe), gx_pix);
                F[k] = ld_off<float2>(scalar_ptr(fl_in + rowe), gx_fl);
            } else {
                int rr, c;
                st1_halo_pos(rs, wv, lane, k - RPW, rr, c);
                const int gy = clamp0(y0t - R + rr, Hm1), gx = clamp0(x0 - R + c, Wm1);
                const unsigned ie = (unsigned)__mul24(gy, W) + (unsigned)gx;
                P[k] = ld_pix<PIX>(prev, ie);
                F[k] = ld_off<float2>(fl_in, ie * 8u);
            }
        }
    };
#ifdef OFLK_STAMPS
    __shared__ unsigned s_st[4][8][16];
    const int st_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int st_tile = 0;
    for (int i = threadIdx.x; i < 4 * 8 * 16; i += 256) (&s_st[0][0][0])[i] = 0u;
    __syncthreads();
    bt[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif

    for (int it = 0; it < (LOOPS ? ntile : 1); it++) {
        // re-derived per tile behind an opaque move: otherwise every per-thread address of all
        // three stages is hoisted out of the loop and held in registers (occupancy 4 -> 2)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if constexpr (CAN_REDO) {
            if (redo) {   // uniform: this trip's tile comes from the list
                const unsigned T = (unsigned)(tiles_x * tiles_y * a.B);
                const unsigned tile = a.redo[2u + T + blockIdx.x + (unsigned)it * gridDim.x];
                b = (int)(tile / (unsigned)(tiles_x * tiles_y));
                const int t = (int)tile - b * (tiles_x * tiles_y);
                tile_y_first = t / tiles_x;
                tile_x = t - tile_y_first * tiles_x;
                x0 = tile_x * k5TX;
                prev = reinterpret_cast<const PIX *>(a.prev) + (size_t)b * plane;
                curr = reinterpret_cast<const PIX *>(a.curr) + (size_t)b * plane;
                if (threadIdx.x == 0) a.redo[2u + tile] = 0u;
            }
        }
        const int tile_y = redo ? tile_y_first : tile_y_first + it;
        const int y0 = tile_y * k5TY;
        const int rstart = (!CHAIN || it == 0 || redo) ? 0 : 2 * R;  // first staging row to compute
#ifdef OFLK_STAMPS
        st_tile = it;
#endif
        OFLK_STAMP(0);   // [0] top of the tile

        float gix[NGA], giy[NGA], git[NGA];
        if (MODE == MODE_GRADS) {
            const float *__restrict__ gtp = a.aux + (size_t)b * plane;
#pragma unroll
            for (int k = 0; k < NG; k++) {
                int e = tid + k * 256;
                int r = e / PW, c = e - r * PW;
                int gy = y0 - HW + r, gx = x0 - HW + c;
                bool in = e < PH * PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
                int i = in ? gy * W + gx : 0;
                gix[k] = in ? (float)prev[i] : 0.0f;
                giy[k] = in ? (float)curr[i] : 0.0f;
                git[k] = in ? gtp[i] : 0.0f;
            }
        } else {
            // ---- stage 1: second frame (warped if ITER), frame average, It -------
            if (CHAIN && it > 0 && !redo) {
                // rows 0 .. 2R-1 are the previous tile's rows TY .. AH-1
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    int qi = tid + j * 256;
                    if (qi < NCARRY) {   // (the mask-free form of the other guards costs this loop's kernel a register too many)
                        s_avg[qi] = carry_a[j];
                        s_it[qi] = carry_i[j];
                    }
                }
            }
            // stage-1 body, instantiated for a fresh tile (rows 0 ..) and a continuing one (rows 2R ..):
            // with the row range known at compile time the per-thread cell count is static
            auto stage1 = [&](auto rs) {
                constexpr int rstart = decltype(rs)::value;
                if (MODE == MODE_ITER) {
                    // (geometry: see st1_loads above)  The bilinear gathers of all the thread's cells are in flight
                    // together when the registers allow it (a continuing tile of the 5x5 window: 7 cells); adjacent
                    // lanes = adjacent cells, so a wave's gather touches 2-3 cache lines per instruction.
                    constexpr int NR = AH - rstart;                  // staging rows to fill
                    constexpr int RPW = (NR + 3) / 4;                // rows (= main cells) per wave / thread
                    constexpr int NHP = (ST1_HC * RPW + 63) / 64;    // halo cells per thread
                    constexpr int NCELL = RPW + NHP;
                    constexpr int BATCH = HW != 2 ? 4 : (NCELL <= OFLK_BATCH ? NCELL : (NCELL + 1) / 2);
                    constexpr int SC = SX - R;                       // staging column of cell column 0
                    const int lane = tid & 63;
                    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                    const int Hm1 = H - 1, Wm1 = W - 1;
                    const LeanGeom lg = lean_geom(H, W);
                    const int gxm = min(x0 + lane, Wm1);
                    const double gxd = (double)gxm;
                    float p[NCELL], q[NCELL];
                    float2 f[NCELL];
                    st1_loads(rs, tid, y0, p, f);
                    OFLK_STAMP(1);   // [1] carry -> LDS, addresses, issue of the coalesced loads (prev, u, v)
                    probe<1>(s_mem);
    #pragma unroll
                    for (int k0 = 0; k0 < NCELL; k0 += BATCH) {
                        LeanFrac tp[BATCH];
                        PairF pr0[BATCH], pr1[BATCH];
    #pragma unroll
                        for (int j = 0; j < BATCH; j++) {
                            const int k = k0 + j;
                            if (k < NCELL) {
                                double y, x;   // int64 + float32 -> float64, as the reference (lucas_kanade_pyramidal.py:88-95)
                                if (k < RPW) {
                                    const int gy = min(max(y0 - R + st1_main_row(rs, wv, k), 0), Hm1);
                                    y = uint_to_f64_bits(gy) + (double)f[k].y;
                                    x = gxd + (double)f[k].x;
                                } else {
                                    int rr, c;
                                    st1_halo_pos(rs, wv, lane, k - RPW, rr, c);
                                    y = (double)clamp0(y0 - R + rr, Hm1) + (double)f[k].y;
                                    x = (double)clamp0(x0 - R + c, Wm1) + (double)f[k].x;
                                }
                                tp[j] = lean_frac_at(lg, y, x);
                                // two 8-byte gathers per cell (the x pair of each tap row)
                                lean_load<false, PIX>(lg, curr, tp[j], pr0[j], pr1[j]);
                            }
                        }
                        OFLK_STAMP(k0 == 0 ? 2 : 4);   // [2]/[4] wait for u, v; taps; gathers issued
    #pragma unroll
                        for (int j = 0; j < BATCH; j++)
                            if (k0 + j < NCELL) {
                                q[k0 + j] = lean_finish(tp[j], pr0[j], pr1[j]);
                                pin(q[k0 + j]);  // finished here, not sunk to its use after the last batch
                            }
                        __builtin_amdgcn_sched_barrier(0);  // one batch of taps in flight at a time
                        OFLK_STAMP(k0 == 0 ? 3 : 5);   // [3]/[5] wait for the gathers; fp64 tap sums
                    }
    #pragma unroll
                    for (int k = 0; k < NCELL; k++) {
                        int o;
                        if (k < RPW) {
                            o = st1_main_row(rs, wv, k) * AS + SX + lane;
                        } else {
                            int rr, c;
                            st1_halo_pos(rs, wv, lane, k - RPW, rr, c);
                            o = rr * AS + c + SC;
                        }
                        const float sum = p[k] + q[k];
                        s_avg[o] = sum * 0.5f;
                        s_it[o] = p[k] - q[k];
                    }
                } else {
                    // SINGLE: groups of four cells over staging rows rstart .. AH-1
                    constexpr int ngroups = (AH - rstart) * k5GW;
                    // clamped image coordinates of group g's first cell; `whole` = the group is one
                    // aligned float4 inside the image
                    auto group_pos = [&](int g, int &gy, int &gx, bool &whole) {
                        g = min(g, ngroups - 1);
                        int r = rstart + g / k5GW, c4 = g % k5GW;
                        gy = min(max(y0 - R + r, 0), H - 1);  // "symm" ring; farther cells are never used
                        gx = x0 - SX + 4 * c4;
                        whole = VEC && gx >= 0 && gx + 3 < W;
                    };
                    auto load4 = [&](const PIX *__restrict__ src, int gy, int gx, bool whole) -> float4 {
                        if (whole) {
                            if constexpr (sizeof(PIX) == 1) {   // four pixels in one aligned dword
                                const unsigned q = *reinterpret_cast<const unsigned *>(src + (unsigned)(gy * W + gx));
                                return make_float4((float)(q & 255u), (float)((q >> 8) & 255u), (float)((q >> 16) & 255u),
                                                   (float)(q >> 24));
                            } else {
                                return *reinterpret_cast<const float4 *>(src + (unsigned)(gy * W + gx));
                            }
                        }
                        const PIX *row = src + (unsigned)(gy * W);
                        float4 r;
                        r.x = (float)row[min(max(gx, 0), W - 1)];
                        r.y = (float)row[min(max(gx + 1, 0), W - 1)];
                        r.z = (float)row[min(max(gx + 2, 0), W - 1)];
                        r.w = (float)row[min(max(gx + 3, 0), W - 1)];
                        return r;
                    };
                    // every coalesced load of the thread's groups goes out first (one HBM latency per tile)
                    float4 p4[NV], q4[NV];
    #pragma unroll
                    for (int k = 0; k < NV; k++) {
                        if (k * 256 < ngroups) {
                            int gy, gx;
                            bool whole;
                            group_pos(tid + k * 256, gy, gx, whole);
                            p4[k] = load4(prev, gy, gx, whole);
                            q4[k] = load4(curr, gy, gx, whole);
                        }
                    }
    #pragma unroll
                    for (int k = 0; k < NV; k++) {
                        const int g = tid + k * 256;
                        if (k * 256 < ngroups && g < ngroups) {
                            float4 pp = p4[k], qv = q4[k], av, dv;
                            // (prev + curr) / 2.0 and prev - curr, lucas_kanade_core.py:36, :43
                            av.x = (pp.x + qv.x) * 0.5f; av.y = (pp.y + qv.y) * 0.5f;
                            av.z = (pp.z + qv.z) * 0.5f; av.w = (pp.w + qv.w) * 0.5f;
                            dv.x = pp.x - qv.x; dv.y = pp.y - qv.y; dv.z = pp.z - qv.z; dv.w = pp.w - qv.w;
                            const int o = rstart * AS + 4 * g;  // = r*AS + 4*c4
                            *reinterpret_cast<float4 *>(&s_avg[o]) = av;
                            *reinterpret_cast<float4 *>(&s_it[o]) = dv;
                        }
                    }
                }
            };
            if (rstart == 0) stage1(std::integral_constant<int, 0>{});
            else stage1(std::integral_constant<int, 2 * R>{});
            OFLK_STAMP(6);    // [6] avg / It -> LDS
            __syncthreads();
            OFLK_STAMP(7);    // [7] barrier 1
            // ---- stage 2: Sobel/8 in convolve2d's tap order; gradients stay in registers
            // Lane = column, wave = RPW2 consecutive rows of the PH x PW gradient tile (the last wave's rows
            // overlap its neighbour's when PH is not a multiple of 4: same values twice).  A wave walks down its
            // rows with the three frame-average rows of the stencil in registers, so a cell reads three new
            // averages and its It, and every LDS address is a per-thread base plus a constant.  The 2 HW halo
            // columns are one more cell for 2 HW * RPW2 of a wave's lanes.
            // gradient cell (r, c) = image (y0-HW+r, x0-HW+c) = staging cell (r+1, c+GC)
            auto sobel = [](float a_mm, float a_m0, float a_mp, float a_0m, float a_0p, float a_pm, float a_p0, float a_pp,
                            float &ix, float &iy) {
                ix = a_pp * -0.125f;
                iy = a_pp * -0.125f;
                ix = fmaf(a_pm, 0.125f, ix);
                ix = fmaf(a_0p, -0.25f, ix);
                ix = fmaf(a_0m, 0.25f, ix);
                ix = fmaf(a_mp, -0.125f, ix);
                ix = fmaf(a_mm, 0.125f, ix);
                iy = fmaf(a_p0, -0.25f, iy);
                iy = fmaf(a_pm, -0.125f, iy);
                iy = fmaf(a_mp, 0.125f, iy);
                iy = fmaf(a_m0, 0.25f, iy);
                iy = fmaf(a_mm, 0.125f, iy);
            };
            {
                const int lane = tid & 63;
                const int rbase = min(__builtin_amdgcn_readfirstlane(tid >> 6) * RPW2, PH - RPW2);
                const float *ap = &s_avg[rbase * AS + SX + lane];   // staging row rbase = row "minus" of gradient row rbase
                const float *tp = &s_it[(rbase + 1) * AS + SX + lane];
                float m_m = ap[-1], m_0 = ap[0], m_p = ap[1];
                float z_m = ap[AS - 1], z_0 = ap[AS], z_p = ap[AS + 1];
#pragma unroll
                for (int k = 0; k < RPW2; k++) {
                    const float p_m = ap[(k + 2) * AS - 1], p_0 = ap[(k + 2) * AS], p_p = ap[(k + 2) * AS + 1];
                    sobel(m_m, m_0, m_p, z_m, z_p, p_m, p_0, p_p, gix[k], giy[k]);
                    git[k] = tp[k * AS];
                    m_m = z_m; m_0 = z_0; m_p = z_p;
                    z_m = p_m; z_0 = p_0; z_p = p_p;
                }
#pragma unroll
                for (int h = 0; h < NHP2; h++) {
                    const int idx = min(lane + 64 * h, HC2 * RPW2 - 1);
                    const int j = idx / HC2, hc = idx - j * HC2;
                    const int r = rbase + j, c = hc < HW ? hc : hc + 64;
                    const float *hp = &s_avg[(r + 1) * AS + (c + GC)];
                    sobel(hp[-AS - 1], hp[-AS], hp[-AS + 1], hp[-1], hp[1], hp[AS - 1], hp[AS], hp[AS + 1], gix[RPW2 + h],
                          giy[RPW2 + h]);
                    git[RPW2 + h] = s_it[(r + 1) * AS + (c + GC)];
                }
            }
            if (CHAIN && it + 1 < ntile && !redo) {
                // staging rows TY .. AH-1 are the next tile's rows 0 .. 2R-1
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    int qi = tid + j * 256;
                    if (qi < NCARRY) {
                        carry_a[j] = s_avg[k5TY * AS + qi];
                        carry_i[j] = s_it[k5TY * AS + qi];
                    }
                }
            }
            OFLK_STAMP(8);    // [8] stage 2: Sobel from LDS, carry rows
            __syncthreads();  // everyone is done reading avg / It: the planes may overwrite them
            OFLK_STAMP(9);    // [9] barrier 2
        }
        // ---- products into the interleaved planes --------------------------------
        if (MODE == MODE_GRADS) {
#pragma unroll
            for (int k = 0; k < NG; k++) {
                int e = tid + k * 256;
                if ((k + 1) * 256 <= PH * PW || e < PH * PW) {   // only the last k is partial
                    float ix = gix[k], iy = giy[k], itv = git[k];
                    s_pa[e] = make_float2(ix * ix, iy * iy);
                    s_pb[e] = make_float2(ix * iy, ix * itv);
                    s_pc[e] = iy * itv;
                }
            }
        } else {
            const int lane = tid & 63;
            const int rbase = min(__builtin_amdgcn_readfirstlane(tid >> 6) * RPW2, PH - RPW2);
#pragma unroll
            for (int k = 0; k < RPW2 + NHP2; k++) {
                int e;
                if (k < RPW2) {
                    e = (rbase + k) * PW + HW + lane;
                } else {
                    const int idx = min(lane + 64 * (k - RPW2), HC2 * RPW2 - 1);
                    const int j = idx / HC2, hc = idx - j * HC2;
                    e = (rbase + j) * PW + (hc < HW ? hc : hc + 64);
                }
                float ix = gix[k], iy = giy[k], itv = git[k];
                s_pa[e] = make_float2(ix * ix, iy * iy);
                s_pb[e] = make_float2(ix * iy, ix * itv);
                s_pc[e] = iy * itv;
            }
        }
        OFLK_STAMP(10);   // [10] products -> LDS
        __syncthreads();
        OFLK_STAMP(11);   // [11] barrier 3

        // ---- stage 3: window sums in NumPy order (pk over plane pairs), solve, write -----
        // thread = 2 (x) by NY (y) outputs; a half-wave spans one tile row, so the
        // 16-byte LDS reads of 32 adjacent lanes are contiguous (conflict-free)
        constexpr int NY = k5NY;
        constexpr int RW = 2 + 2 * HW;   // product columns a thread reads per row
        const int tx = tid & 31, ty = tid >> 5;
        Sum2 sA[NY][2], sB[NY][2];
        float sC[NY][2];
        // rows of the interleaved planes: RW float2 = RW/2 aligned 16-byte reads; RW floats = RW/2 8-byte reads
        auto load_f2 = [](const float2 *src, Sum2 (&row)[RW]) {
            const float4 *r4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
            for (int j = 0; j < RW / 2; j++) {
                float4 qv = r4[j];
                row[2 * j] = sum2(qv.x, qv.y);
                row[2 * j + 1] = sum2(qv.z, qv.w);
            }
        };
        auto load_f1 = [](const float *src, float (&row)[RW]) {
            const float2 *r2 = reinterpret_cast<const float2 *>(src);
#pragma unroll
            for (int j = 0; j < RW / 2; j++) {
                float2 qv = r2[j];
                row[2 * j] = qv.x;
                row[2 * j + 1] = qv.y;
            }
        };
        // flow += d (lucas_kanade_pyramidal.py:209-210) needs the current flow of the thread's outputs: requested
        // before the sums of the last (scalar) plane, where the register pressure of the float2 planes is gone --
        // the loads then have that plane's sums and the divisions to arrive (only where the registers are there:
        // 5x5 window, width a multiple of 4; otherwise just before their use)
        constexpr bool PRELOAD = MODE == MODE_ITER && HW == 2 && VEC;
        float4 pf[NY];   // {u0, v0, u1, v1} of the two pixels
        auto preload = [&]() {
            const float2 *__restrict__ fin0 = a.fl[sel] + (size_t)b * plane;
            const int gx_ = x0 + 2 * tx, gy_ = y0 + NY * ty;
            const unsigned oel0 = (unsigned)__mul24(gy_, W) + (unsigned)gx_;
#pragma unroll
            for (int oy = 0; oy < NY; oy++) {
                const bool in = gy_ + oy < H && gx_ < W;
                pf[oy] = in ? ld_off<float4>(fin0, (oel0 + (unsigned)(oy * W)) * 8u) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        };
        if constexpr (HW == 2) {
            const float2 *ba = &s_pa[(NY * ty) * PW + 2 * tx];
            patch5_sums<Sum2, NY>([&](int i, Sum2 (&row)[6]) { load_f2(ba + i * PW, row); }, sA);
            const float2 *bb = &s_pb[(NY * ty) * PW + 2 * tx];
            patch5_sums<Sum2, NY>([&](int i, Sum2 (&row)[6]) { load_f2(bb + i * PW, row); }, sB);
            if constexpr (PRELOAD) preload();
            const float *bc = &s_pc[(NY * ty) * PW + 2 * tx];
            patch5_sums<float, NY>([&](int i, float (&row)[6]) { load_f1(bc + i * PW, row); }, sC);
        } else {
#pragma unroll
            for (int oy = 0; oy < NY; oy++) {
                const float2 *ba = &s_pa[(NY * ty + oy) * PW + 2 * tx];
                window_sums_row<Sum2, HW, 2>([&](int i, Sum2 (&row)[RW]) { load_f2(ba + i * PW, row); }, sA[oy]);
                __builtin_amdgcn_sched_barrier(0);
                const float2 *bb = &s_pb[(NY * ty + oy) * PW + 2 * tx];
                window_sums_row<Sum2, HW, 2>([&](int i, Sum2 (&row)[RW]) { load_f2(bb + i * PW, row); }, sB[oy]);
                __builtin_amdgcn_sched_barrier(0);
                const float *bc = &s_pc[(NY * ty + oy) * PW + 2 * tx];
                window_sums_row<float, HW, 2>([&](int i, float (&row)[RW]) { load_f1(bc + i * PW, row); }, sC[oy]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        probe<3>(s_mem);
        OFLK_STAMP(12);   // [12] stage 3: window sums
        const int gxb = x0 + 2 * tx;
        float su = 0.0f, sv = 0.0f;
        // planar outputs (SINGLE / GRADS always; ITER when this launch delivers the call's result)
        float *__restrict__ ou = a.ou + (size_t)b * plane;
        float *__restrict__ ov = a.ov + (size_t)b * plane;
        // interleaved {u, v} slots (ITER)
        const float2 *__restrict__ fin = MODE == MODE_ITER ? a.fl[sel] + (size_t)b * plane : nullptr;
        float2 *__restrict__ fout = MODE == MODE_ITER ? a.fl[1 - sel] + (size_t)b * plane : nullptr;
        const bool planar = MODE != MODE_ITER || a.planar_out != 0;   // uniform
        // borders stay zero (lucas_kanade_core.py:101-108); the column tests are per thread,
        // the row test per output row
        const bool okx0 = (gxb >= HW) & (gxb < W - HW), okx1 = (gxb + 1 >= HW) & (gxb + 1 < W - HW);
        const int gyb = y0 + NY * ty;
        // element offset of the thread's first output inside the plane (a plane is < 4 GiB)
        unsigned oel = (unsigned)__mul24(gyb, W) + (unsigned)gxb;
        const bool pairs = VEC || (W & 1) == 0;   // gxb is even: two adjacent pixels are one aligned access
#pragma unroll
        for (int oy = 0; oy < NY; oy++, oel += (unsigned)W) {
            const int gy = gyb + oy;
            const bool oky = (gy >= HW) & (gy < H - HW);
            float du[2], dv[2];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                float u, v;
                lk_solve(sA[oy][o].x, sA[oy][o].y, sB[oy][o].x, sB[oy][o].y, sC[oy][o], u, v);
                const bool interior = oky & (o ? okx1 : okx0);
                du[o] = interior ? u : 0.0f;
                dv[o] = interior ? v : 0.0f;
                if (MODE == MODE_ITER) {  // pixels outside the image are not interior: they add 0
                    su += fabsf(du[o]);
                    sv += fabsf(dv[o]);
                }
            }
            if (gy < H && gxb < W) {
                if (pairs) {
                    float2 ru = make_float2(du[0], du[1]);
                    float2 rv = make_float2(dv[0], dv[1]);
                    if (MODE == MODE_ITER) {
                        if (!PRELOAD) pf[oy] = ld_off<float4>(fin, oel * 8u);
                        ru.x = pf[oy].x + ru.x; ru.y = pf[oy].z + ru.y;
                        rv.x = pf[oy].y + rv.x; rv.y = pf[oy].w + rv.y;
                    }
                    if (planar) {
                        st_off<float2>(ou, oel * 4u, ru);
                        st_off<float2>(ov, oel * 4u, rv);
                    } else {
                        st_off<float4>(fout, oel * 8u, make_float4(ru.x, rv.x, ru.y, rv.y));
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        if (gxb + o < W) {
                            float ru = du[o], rv = dv[o];
                            if (MODE == MODE_ITER) {
                                const float2 f = ld_off<float2>(fin, (oel + o) * 8u);
                                ru = f.x + ru;
                                rv = f.y + rv;
                            }
                            if (planar) {
                                st_off<float>(ou, (oel + o) * 4u, ru);
                                st_off<float>(ov, (oel + o) * 4u, rv);
                            } else {
                                st_off<float2>(fout, (oel + o) * 8u, make_float2(ru, rv));
                            }
                        }
                    }
                }
            }
        }

        if (MODE == MODE_ITER) {
            // |d| sums of the tile: six fp32 terms per thread, a fixed-order wave reduction
            // (DPP adds, total in lane 63), then fp64 across the four waves.  The reference's
            // np.mean is itself an fp32 pairwise sum; see lk_report / lk_level_state.
            su = wave_sum_to_lane63(su);
            sv = wave_sum_to_lane63(sv);
            if ((tid & 63) == 63) {
                s_red[0][tid >> 6] = (double)su;
                s_red[1][tid >> 6] = (double)sv;
            }
        }
        OFLK_STAMP(13);   // [13] solve, flow += d, stores, |d| reduction
        // stage 3 has read the planes (the next tile's staging overwrites them) and s_red is complete
        if (MODE == MODE_ITER || (LOOPS && it + 1 < ntile)) __syncthreads();
        OFLK_STAMP(14);   // [14] barrier 4
        if (MODE == MODE_ITER && tid == 0) {
            blk_u += (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
            blk_v += (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        }
    }
#ifdef OFLK_STAMPS
    bt[2] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    if (MODE == MODE_ITER && threadIdx.x == 0) lk_report(a, b, blk_u, blk_v);
    if constexpr (CAN_REDO) {
        // every block read the count when it started; the last one to finish leaves the list empty for the next call
        if (redo && threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(&a.redo[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                a.redo[0] = 0u;
                a.redo[1] = 0u;
            }
        }
    }
#ifdef OFLK_STAMPS
    __syncthreads();
    if (a.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the report's atomics have been issued and acknowledged
        bt[3] = (unsigned)__builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            unsigned *o = a.stamps + (size_t)gridDim.x * 512 + (size_t)blockIdx.x * 8;
            o[0] = bt[0]; o[1] = bt[1]; o[2] = bt[2]; o[3] = bt[3];
            o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID: wave, simd, pipe, cu, sh, se ...
            o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
            o[6] = (unsigned)ntile;
            o[7] = 1u;
        }
        if (threadIdx.x < 4) s_st[threadIdx.x][0][15] = (unsigned)ntile;
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * 8 * 16; i += 256) a.stamps[(size_t)blockIdx.x * 512 + i] = (&s_st[0][0][0])[i];
    }
#endif
}

// ---------------------------------------------------------------------------
// K2: Gaussian blur (scipy.ndimage.gaussian_filter -> correlate1d, symmetric
// branch): fp64 accumulation  tmp = x[c]*w0; for k = r..1: tmp += (x[c-k]+x[c+k])*w[k],
// fp32 store after each axis (lucas_kanade_pyramidal.py:46-47).
// ---------------------------------------------------------------------------
struct GaussW {
    double w[kMaxRadius + 1];
    int radius;
};

__device__ __forceinline__ int reflect_idx(int i, int n)
{
    // scipy.ndimage "reflect": (d c b a | a b c d | d c b a)
    if (i >= 0 && i < n) return i;
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// AXIS 0: along y (columns), AXIS 1: along x (rows)
// FMA: the opt-in contracted form (see k_pyr_down): the multiply and the add of a tap fused
template <int AXIS, bool FMA = false>
__global__ __launch_bounds__(256) void k_blur(const float *__restrict__ in, float *__restrict__ out,
                                              int H, int W, GaussW g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * (size_t)W;
    const float *__restrict__ src = in + (size_t)blockIdx.z * plane;
    const int len = AXIS == 0 ? H : W;
    const int c = AXIS == 0 ? y : x;
    auto at = [&](int i) -> double {
        int j = reflect_idx(i, len);
        return (double)(AXIS == 0 ? src[(size_t)j * W + x] : src[(size_t)y * W + j]);
    };
    double tmp = at(c) * g.w[0];
    for (int k = g.radius; k >= 1; k--) {
        double s = at(c - k) + at(c + k);
        if constexpr (FMA) {
            tmp = __builtin_fma(s, g.w[k], tmp);
        } else {
            double m = s * g.w[k];
            tmp = tmp + m;
        }
    }
    out[(size_t)blockIdx.z * plane + (size_t)y * W + x] = (float)tmp;
}

// bilinear sampling on the linspace(0,H-1,Ho) x linspace(0,W-1,Wo) grid
// (lucas_kanade_pyramidal.py:55-59); NPL planes resampled with shared taps, each
// optionally scaled in fp32 afterwards (upsample_flow, :126-136).
struct ResampleArgs {
    const float *in[2];   // [nimg][H][W] each
    float *out[2];        // [nimg][Ho][Wo]
    // optional (flow upsample): the source is the ping-pong slot that holds level `acc_level`'s
    // final flow, slot * in_sel_stride elements after in[]
    const unsigned long long *acc;
    int acc_level, L, K;   // K: accumulator layout (>= 1)
    int iters;             // iterations launched per level (0 when the plan has none)
    unsigned long long acc_thr;
    size_t in_sel_stride; // elements between ping-pong buffers (0 when unused)
    float scale[2];
    int H, W, Ho, Wo;
    Linspace ly, lx;
    int nplanes;          // 1 or 2
    int apply_scale;
    int vec_store;        // Wo % 4 == 0 and 16-byte aligned outputs: 16-byte stores (host decides)
    // flow upsample inside a pyramidal call: in[0] / out[0] are INTERLEAVED float2 {u, v} planes
    // (in_sel_stride then counts float2 elements); 0 = two planar planes each (the standalone entry)
    int interleaved;
};

// One thread produces 4 horizontally adjacent outputs of NP planes: the row taps and
// weights are computed once, all 16*NP tap loads are issued before any arithmetic, and
// the stores are 16 bytes per lane.  grid = (ceil(Wo/256), ceil(Ho/4), nimg), block = 64 x 4.
// FMA: the opt-in contracted form of the tap sum (as stage D of k_pyr_down<PIX, true>), pyramid steps only
template <int NP, bool FMA = false>
__global__ __launch_bounds__(256) void k_resample(ResampleArgs a)
{
    const int j0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j0 >= a.Wo || i >= a.Ho) return;
    const int img = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t ip = (size_t)H * W, op = (size_t)a.Ho * a.Wo;
    size_t selofs = 0;
    if (a.acc)
        selofs = (size_t)(lk_level_state(a.acc, img, a.acc_level, a.iters, a.L, a.K, a.acc_thr).executed & 1) *
                 a.in_sel_stride;
    // row part of map_coordinates(order=1, mode="constant")
    const double y = linspace_at(a.ly, i);
    const bool y_in = !(y < 0.0 || y > (double)(H - 1));
    const double fy = floor(y);
    const int y0 = min(max((int)fy, 0), H - 1);
    const double wy0 = 1.0 - (y - fy), wy1 = 1.0 - wy0;
    const int y1 = (y0 + 1 < H) ? y0 + 1 : (H > 1 ? H - 2 : 0);
    const unsigned row0 = (unsigned)(y0 * W), row1 = (unsigned)(y1 * W);
    double wx0[4], wx1[4];
    bool inside[4];
    float t[NP][4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = min(j0 + k, a.Wo - 1);
        const double x = linspace_at(a.lx, j);
        inside[k] = y_in && !(x < 0.0 || x > (double)(W - 1));
        const double fx = floor(x);
        const int x0 = min(max((int)fx, 0), W - 1);
        wx0[k] = 1.0 - (x - fx);
        wx1[k] = 1.0 - wx0[k];
        const int x1 = (x0 + 1 < W) ? x0 + 1 : (W > 1 ? W - 2 : 0);
        if (NP == 2 && a.interleaved) {
            const float2 *__restrict__ src = reinterpret_cast<const float2 *>(a.in[0]) + selofs + (size_t)img * ip;
            const float2 q0 = src[row0 + (unsigned)x0], q1 = src[row0 + (unsigned)x1], q2 = src[row1 + (unsigned)x0],
                         q3 = src[row1 + (unsigned)x1];
            t[0][k][0] = q0.x; t[0][k][1] = q1.x; t[0][k][2] = q2.x; t[0][k][3] = q3.x;
            t[NP - 1][k][0] = q0.y; t[NP - 1][k][1] = q1.y; t[NP - 1][k][2] = q2.y; t[NP - 1][k][3] = q3.y;
        } else {
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const float *__restrict__ src = a.in[p] + selofs + (size_t)img * ip;
                t[p][k][0] = src[row0 + (unsigned)x0];
                t[p][k][1] = src[row0 + (unsigned)x1];
                t[p][k][2] = src[row1 + (unsigned)x0];
                t[p][k][3] = src[row1 + (unsigned)x1];
            }
        }
    }
    float res[NP][4];
#pragma unroll
    for (int p = 0; p < NP; p++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double acc = 0.0, c;
            if constexpr (FMA) {
                c = (double)t[p][k][0]; c = c * wy0; acc = c * wx0[k];
                c = (double)t[p][k][1]; c = c * wy0; acc = __builtin_fma(c, wx1[k], acc);
                c = (double)t[p][k][2]; c = c * wy1; acc = __builtin_fma(c, wx0[k], acc);
                c = (double)t[p][k][3]; c = c * wy1; acc = __builtin_fma(c, wx1[k], acc);
            } else {
                c = (double)t[p][k][0]; c = c * wy0; c = c * wx0[k]; acc = acc + c;
                c = (double)t[p][k][1]; c = c * wy0; c = c * wx1[k]; acc = acc + c;
                c = (double)t[p][k][2]; c = c * wy1; c = c * wx0[k]; acc = acc + c;
                c = (double)t[p][k][3]; c = c * wy1; c = c * wx1[k]; acc = acc + c;
            }
            float r = inside[k] ? (float)acc : 0.0f;
            if (a.apply_scale) r = r * a.scale[p];
            res[p][k] = r;
        }
    }
    if (NP == 2 && a.interleaved) {
        float2 *__restrict__ dst = reinterpret_cast<float2 *>(a.out[0]) + (size_t)img * op + (size_t)i * a.Wo + j0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (j0 + k < a.Wo) dst[k] = make_float2(res[0][k], res[NP - 1][k]);
        return;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        float *__restrict__ dst = a.out[p] + (size_t)img * op + (size_t)i * a.Wo + j0;
        if (a.vec_store) {
            *reinterpret_cast<float4 *>(dst) = make_float4(res[p][0], res[p][1], res[p][2], res[p][3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (j0 + k < a.Wo) dst[k] = res[p][k];
        }
    }
}

// ---------------------------------------------------------------------------
// K2 fused: one pyramid step (gaussian_filter sigma = 2, radius 8, then the
// linspace bilinear downsample; lucas_kanade_pyramidal.py:46-59) in one kernel.
// A block produces a 32 x 16 tile of the coarse level:
//   A  input rows/cols (reflect-extended) needed by the tile       -> LDS
//   B  vertical 17-tap pass, fp64 accumulation, fp32 store (SciPy stores fp32
//      between axes)                                                -> LDS
//   C  horizontal 17-tap pass on that, fp64 accumulation, fp32      -> LDS
//   D  bilinear sampling of the blurred tile in fp64                -> HBM
// Same operation order as the unfused k_blur<0>, k_blur<1>, k_resample chain;
// the blurred image never goes to HBM (algorithmic traffic: read 4 B per fine
// pixel, write 4 B per coarse pixel).
// The host checks that a tile's source span fits the static LDS tile
// (pyr_fused_fits) and otherwise uses the unfused chain.
// ---------------------------------------------------------------------------
constexpr int kPTW = 32, kPTH = 16;          // coarse outputs per block
constexpr int kPBW = 2 * kPTW + 2;           // blurred cols a tile may need (66)
constexpr int kPBH = 2 * kPTH + 2;           // blurred rows (34)
constexpr int kPIW = kPBW + 16;              // input cols incl. radius-8 halo (82)
constexpr int kPIH = kPBH + 16;              // input rows (50)
constexpr int kPVS = kPIW + 1;               // row stride of the vertical-pass tile (odd)
constexpr int kPHS = kPBW + 1;               // row stride of the blurred tile

struct PyrArgs {
    const float *in;   // [nimg][H][W]; images nsplit .. come from in2 instead (two caller buffers, one launch)
    const float *in2;
    int nsplit;
    float *out;        // [nimg][Ho][Wo]
    // optional, at the start of a pyramidal call: words to zero (per-call state) and the coarsest
    // level's flow planes (lucas_kanade_pyramidal.py:182-184), so that no launch of its own is needed
    unsigned *zero_words;
    size_t n_zero_words;
    float *zero_u, *zero_v;
    size_t n_zero_flow;
    int H, W, Ho, Wo;
    Linspace ly, lx;
    double w[9];       // gaussian weights, w[k] at distance k
};

// FMA = false: SciPy's operation sequence, every fp64 operation rounded on its own (the default; results equal to
// the reference's).  FMA = true (opt-in, oflk_plan_set_arithmetic): the same sums with the multiply and the add of a
// tap fused -- 17 instead of 25 fp64 operations per blurred value, on a kernel the fp64 pipe binds.  Every
// intermediate then differs from SciPy's by at most a few 1e-16 relative BEFORE it is rounded to float32 exactly where
// SciPy rounds (after each axis, after the sampling), so a float32 value differs from the reference's only where the
// fp64 value lies within that distance of a float32 rounding boundary: about one value in 10^7, by one float32 ulp.
template <class PIX, bool FMA = false>
__global__ __launch_bounds__(256) void k_pyr_down(PyrArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_in[kPIH * kPIW];   // stage A (8-byte column-pair writes); reused for the blurred tile (stage C output)
    __shared__ float s_v[kPBH * kPVS];
    float *s_h = s_in;
    static_assert(kPBH * kPHS <= kPIH * kPIW, "blurred tile must fit the input tile's storage");

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const size_t ip = (size_t)H * (size_t)W, op = (size_t)a.Ho * (size_t)a.Wo;
    const int img = blockIdx.z;
    const PIX *__restrict__ src = img < a.nsplit ? reinterpret_cast<const PIX *>(a.in) + (size_t)img * ip
                                                 : reinterpret_cast<const PIX *>(a.in2) + (size_t)(img - a.nsplit) * ip;
    if (a.zero_words) {
        // grid-stride over all blocks of the launch
        const size_t nblk = (size_t)gridDim.x * gridDim.y * gridDim.z;
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (size_t i = blk * 256 + threadIdx.x; i < a.n_zero_words; i += nblk * 256) a.zero_words[i] = 0u;
        for (size_t i = blk * 256 + threadIdx.x; i < a.n_zero_flow; i += nblk * 256) {
            a.zero_u[i] = 0.0f;
            a.zero_v[i] = 0.0f;
        }
    }
    const int j0 = blockIdx.x * kPTW, i0 = blockIdx.y * kPTH;
    // first blurred row / column any tap of this tile touches (a sample that lands
    // exactly on the last row reads the mirrored row H-2 with weight 0)
    int ylo = (int)floor(linspace_at(a.ly, i0));
    int xlo = (int)floor(linspace_at(a.lx, j0));
    if (i0 + kPTH >= a.Ho) ylo = min(ylo, max(H - 2, 0));
    if (j0 + kPTW >= a.Wo) xlo = min(xlo, max(W - 2, 0));

    // ---- A: input tile, reflect-extended; all of a thread's loads go out before the
    // first LDS write (one memory latency per tile instead of one per element).
    // Element k of a thread is e = tid + 256 k; its (row, col) advance by (QA, RA) with a
    // carry at kPIW columns, so the only division is the one that places element 0.
    // Three block-uniform cases: the tile's source span lies inside the image (plain
    // addresses: every tile but the frame's border ring), it leaves the image by less than
    // one image size (one mirror step, no division), anything else (tiny images). ------
    {
        constexpr int NA = (kPIH * kPIW + 255) / 256;  // 17 elements per thread
        constexpr int QA = 256 / kPIW, RA = 256 % kPIW;
        constexpr int NLAST = kPIH * kPIW - (NA - 1) * 256;   // threads that own an element NA-1
        const int ybase = ylo - 8, xbase = xlo - 8;
        const int r0 = tid / kPIW, c0 = tid - r0 * kPIW;
        float vals[NA];
        bool staged = false;
        if constexpr (sizeof(PIX) == 4 && OFLK_PYR_PAIRS) {
            if (ybase >= 0 && ybase + kPIH <= H && xbase >= 0 && xbase + kPIW <= W) {
                // interior tile, float32 frames: column PAIRS, one 8-byte load and one 8-byte LDS write each (the tile is
                // 82 = 2 x 41 columns wide; 2050 pairs = 8 per thread + 2)
                constexpr int PWD = kPIW / 2, NPR = kPIH * PWD, NB = (NPR + 255) / 256, QB = 256 / PWD, RB = 256 % PWD;
                static_assert(kPIW % 2 == 0, "pairs");
                float2 pv[NB];
                int r = tid / PWD, c = tid - r * PWD;
                unsigned off = (unsigned)__mul24(ybase + r, W) + (unsigned)(xbase + 2 * c);
                const unsigned step = (unsigned)__mul24(QB, W) + 2 * RB, wrap = (unsigned)(W - kPIW);
                int cc = c;
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    if ((k + 1) * 256 <= NPR || tid + k * 256 < NPR) pv[k] = ld_off<float2>(src, off * 4u);
                    off += step; cc += RB;
                    if (cc >= PWD) { cc -= PWD; off += wrap; }
                }
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if ((k + 1) * 256 <= NPR || tid + k * 256 < NPR) reinterpret_cast<float2 *>(s_in)[tid + k * 256] = pv[k];   // pair e -> cells 2e, 2e+1
                staged = true;
            }
        }
        if (staged) {
        } else
        if (ybase >= 0 && ybase + kPIH <= H && xbase >= 0 && xbase + kPIW <= W) {
            unsigned off = (unsigned)__mul24(ybase + r0, W) + (unsigned)(xbase + c0);   // element offsets
            const unsigned step = (unsigned)__mul24(QA, W) + RA, wrap = (unsigned)(W - kPIW);
            int c = c0;
#pragma unroll
            for (int k = 0; k < NA; k++) {
                if (k < NA - 1 || tid < NLAST) vals[k] = ld_pix<PIX>(src, off);
                off += step; c += RA;
                if (c >= kPIW) { c -= kPIW; off += wrap; }
            }
        } else {
            // one mirror step covers indices in [-n, 2n): scipy "reflect" (d c b a | a b c d | d c b a)
            const bool one_step = ybase >= -H && ybase + kPIH <= 2 * H && xbase >= -W && xbase + kPIW <= 2 * W;
            int r = r0, c = c0;
#pragma unroll
            for (int k = 0; k < NA; k++) {
                int gy = ybase + r, gx = xbase + c;
                if (one_step) {
                    gy = gy < 0 ? -1 - gy : (gy >= H ? 2 * H - 1 - gy : gy);
                    gx = gx < 0 ? -1 - gx : (gx >= W ? 2 * W - 1 - gx : gx);
                } else {
                    gy = reflect_idx(gy, H);
                    gx = reflect_idx(gx, W);
                }
                if (k < NA - 1 || tid < NLAST) vals[k] = ld_pix<PIX>(src, (unsigned)__mul24(gy, W) + (unsigned)gx);
                c += RA; r += QA;
                if (c >= kPIW) { c -= kPIW; r += 1; }
            }
        }
        if (!staged) {
#pragma unroll
            for (int k = 0; k < NA; k++)
                if (k < NA - 1 || tid < NLAST) s_in[tid + k * 256] = vals[k];
        }
    }
    __syncthreads();

    // ---- B: vertical pass; thread = one column, 12 consecutive rows ----------
    {
        constexpr int RS = 12;                       // rows per thread (3 segments cover 34)
        const int col = tid % kPIW, seg = tid / kPIW;
        if (seg < 3) {
            // the window's rows at a per-thread base plus constants; only the last segment's last rows lie past the
            // tile (they feed outputs that are not stored) and are clamped
            constexpr int KSAFE = kPIH - 2 * RS;         // rows k < KSAFE exist for every segment
            const float *wb = &s_in[seg * RS * kPIW + col];
            float win[RS + 16];
#pragma unroll
            for (int k = 0; k < RS + 16; k++) win[k] = k < KSAFE ? wb[k * kPIW] : wb[min(k, kPIH - 1 - seg * RS) * kPIW];
#pragma unroll
            for (int o = 0; o < RS; o++) {
                using AccB = double;
                AccB t = (AccB)win[o + 8] * (AccB)a.w[0];
#pragma unroll
                for (int k = 8; k >= 1; k--) {
                    AccB sgm = (AccB)win[o + 8 - k] + (AccB)win[o + 8 + k];
                    if constexpr (FMA) {
                        t = __builtin_fma(sgm, (AccB)a.w[k], t);
                    } else {
                        AccB m = sgm * (AccB)a.w[k];
                        t = t + m;
                    }
                }
                int r = seg * RS + o;
                if (r < kPBH) s_v[r * kPVS + col] = (float)t;
            }
        }
    }
    __syncthreads();

    // ---- C: horizontal pass; thread = one row, 10 consecutive columns (34 rows x 7 segments =
    // 238 busy lanes; the last segment's surplus columns are computed but not stored) --------
    {
        constexpr int CS = 10;                       // cols per thread (7 segments cover 66)
        constexpr int NSEG = (kPBW + CS - 1) / CS;
        static_assert(kPBH * NSEG <= 256, "one thread per (row, segment)");
        const int row = tid % kPBH, seg = tid / kPBH;
        if (seg < NSEG) {
            constexpr int KSAFE = kPIW - (NSEG - 1) * CS;   // columns k < KSAFE exist for every segment
            const float *wb = &s_v[row * kPVS + seg * CS];
            float win[CS + 16];
#pragma unroll
            for (int k = 0; k < CS + 16; k++) win[k] = k < KSAFE ? wb[k] : wb[min(k, kPIW - 1 - seg * CS)];
#pragma unroll
            for (int o = 0; o < CS; o++) {
                using AccC = double;
                AccC t = (AccC)win[o + 8] * (AccC)a.w[0];
#pragma unroll
                for (int k = 8; k >= 1; k--) {
                    AccC sgm = (AccC)win[o + 8 - k] + (AccC)win[o + 8 + k];
                    if constexpr (FMA) {
                        t = __builtin_fma(sgm, (AccC)a.w[k], t);
                    } else {
                        AccC m = sgm * (AccC)a.w[k];
                        t = t + m;
                    }
                }
                if (seg * CS + o < kPBW) s_h[row * kPHS + seg * CS + o] = (float)t;
            }
        }
    }
    __syncthreads();

    // ---- D: bilinear sampling from the blurred tile ----------------------------
    float *__restrict__ dst = a.out + (size_t)blockIdx.z * op;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        int o = tid + k * 256;
        int i = i0 + o / kPTW, j = j0 + o % kPTW;
        if (i >= a.Ho || j >= a.Wo) continue;
        double y = linspace_at(a.ly, i), x = linspace_at(a.lx, j);
        float r = 0.0f;
        {   // np.linspace(0, S-1, T) never leaves [0, S-1] (its last point IS S-1): map_coordinates' outside test cannot fire
            double fy = floor(y), fx = floor(x);
            int y0 = (int)fy, x0 = (int)fx;
            double wy0 = 1.0 - (y - fy), wx0 = 1.0 - (x - fx);
            double wy1 = 1.0 - wy0, wx1 = 1.0 - wx0;
            int y1 = (y0 + 1 < H) ? y0 + 1 : (H > 1 ? H - 2 : 0);
            int x1 = (x0 + 1 < W) ? x0 + 1 : (W > 1 ? W - 2 : 0);
            const float *r0 = s_h + (y0 - ylo) * kPHS - xlo;
            const float *r1 = s_h + (y1 - ylo) * kPHS - xlo;
            double acc = 0.0, c;
            if constexpr (FMA) {
                c = (double)r0[x0]; c = c * wy0; acc = c * wx0;
                c = (double)r0[x1]; c = c * wy0; acc = __builtin_fma(c, wx1, acc);
                c = (double)r1[x0]; c = c * wy1; acc = __builtin_fma(c, wx0, acc);
                c = (double)r1[x1]; c = c * wy1; acc = __builtin_fma(c, wx1, acc);
            } else {
                c = (double)r0[x0]; c = c * wy0; c = c * wx0; acc = acc + c;
                c = (double)r0[x1]; c = c * wy0; c = c * wx1; acc = acc + c;
                c = (double)r1[x0]; c = c * wy1; c = c * wx0; acc = acc + c;
                c = (double)r1[x1]; c = c * wy1; c = c * wx1; acc = acc + c;
            }
            r = (float)acc;
        }
        dst[(size_t)i * a.Wo + j] = r;
    }
}

// ---------------------------------------------------------------------------
// K4: upsample_flow (lucas_kanade_pyramidal.py:100-138) with the coarse tile staged in
// LDS.  A block produces 256 x 16 fine outputs of both planes; the coarse cells it
// samples (at most kUSW columns x kUSH rows per plane) are fetched with coalesced
// loads once, and the 16 taps per plane of each thread come from LDS (gathering them
// from global memory made the kernel L1-throughput bound).  Same arithmetic as
// k_resample<2> with scaling; the host checks that every block's source span fits
// (upsample_fits) and falls back to k_resample<2> otherwise.
// ---------------------------------------------------------------------------
constexpr int kUTW = 256, kUTH = 16;  // fine outputs per block: a thread owns 4 (x) by 4 (y) of them
constexpr int kUSW = 136, kUSH = 10;  // coarse columns / rows staged per plane

__global__ __launch_bounds__(256) void k_upsample(ResampleArgs a)
{
    __shared__ float s_src[2][kUSH][kUSW];
    const int tid = threadIdx.x;
    const int jb = blockIdx.x * kUTW, ib = blockIdx.y * kUTH;
    const int img = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t ip = (size_t)H * W, op = (size_t)a.Ho * a.Wo;
    size_t selofs = 0;
    if (a.acc)
        selofs = (size_t)(lk_level_state(a.acc, img, a.acc_level, a.iters, a.L, a.K, a.acc_thr).executed & 1) *
                 a.in_sel_stride;
    // first coarse row / column any tap of this block touches (a sample that lands
    // exactly on the last index reads the mirrored index N-2 with weight 0)
    int ylo = (int)floor(linspace_at(a.ly, ib));
    int xlo = (int)floor(linspace_at(a.lx, jb));
    if (ib + kUTH >= a.Ho) ylo = min(ylo, max(H - 2, 0));
    if (jb + kUTW >= a.Wo) xlo = min(xlo, max(W - 2, 0));
    if (a.interleaved) {
        // interleaved {u, v} source: one 8-byte load per cell, split into the two LDS planes
        constexpr int NL = (kUSH * kUSW + 255) / 256;      // 6 staged cells per thread
        constexpr int QS = 256 / kUSW, RS = 256 % kUSW;    // (row, column) advance per 256 cells
        const float2 *__restrict__ src = reinterpret_cast<const float2 *>(a.in[0]) + selofs + (size_t)img * ip;
        float2 vals[NL];
        int r = tid / kUSW, c = tid - r * kUSW;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int gy = min(ylo + min(r, kUSH - 1), H - 1), gx = min(xlo + c, W - 1);
            vals[k] = ld_off<float2>(src, ((unsigned)__mul24(gy, W) + (unsigned)gx) * 8u);
            c += RS; r += QS;
            if (c >= kUSW) { c -= kUSW; r += 1; }
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int e = tid + k * 256;
            if ((k + 1) * 256 <= kUSH * kUSW || e < kUSH * kUSW) {
                (&s_src[0][0][0])[e] = vals[k].x;
                (&s_src[1][0][0])[e] = vals[k].y;
            }
        }
    } else {
        // coalesced staging: all loads of a thread are issued before the first LDS write
        constexpr int NL = (2 * kUSH * kUSW + 255) / 256;  // 11 staged cells per thread
        constexpr int QS = 256 / kUSW, RS = 256 % kUSW;    // (row, column) advance per 256 cells
        float vals[NL];
        int r = tid / kUSW, c = tid - r * kUSW;            // row runs over both planes: 2 * kUSH rows
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int rr = min(r, 2 * kUSH - 1);           // only the last k runs past the tile
            const int p = rr >= kUSH ? 1 : 0, row = rr - p * kUSH;
            const int gy = min(ylo + row, H - 1), gx = min(xlo + c, W - 1);
            vals[k] = ld_off<float>(a.in[p] + selofs + (size_t)img * ip, ((unsigned)__mul24(gy, W) + (unsigned)gx) * 4u);
            c += RS; r += QS;
            if (c >= kUSW) { c -= kUSW; r += 1; }
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int e = tid + k * 256;
            if ((k + 1) * 256 <= 2 * kUSH * kUSW || e < 2 * kUSH * kUSW) (&s_src[0][0][0])[e] = vals[k];
        }
    }
    __syncthreads();

    const int j0 = jb + (tid & 63) * 4;
    const int i0 = ib + (tid >> 6) * 4;
    if (j0 >= a.Wo || i0 >= a.Ho) return;
    // Same sampling as lean_taps (see there): one unsigned compare per axis for the range test,
    // and a sample exactly on the last index is expressed from the cell before it (floor capped at
    // N-2, fraction exactly 1), so the two taps of an axis are always adjacent cells of the staged
    // tile.  The x side is computed once per thread and reused by its four rows.
    const LeanGeom lg = lean_geom(H, W);
    const int rstep = H > 1 ? kUSW : 0;           // LDS words from tap row 0 to tap row 1
    const int cstep = W > 1 ? 1 : 0;
    double wx0[4], wx1[4];
    int xo[4];
    bool x_in[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = min(j0 + k, a.Wo - 1);
        const double x = linspace_at(a.lx, j);
        x_in[k] = (unsigned long long)__double_as_longlong(x) <= (unsigned long long)__double_as_longlong(lg.Wm1);
        const double fx = fmin(floor(x), lg.Wm2);
        wx0[k] = 1.0 - (x - fx);
        wx1[k] = 1.0 - wx0[k];
        xo[k] = x_in[k] ? (int)fx - xlo : 0;
    }
    const float *__restrict__ tile = &s_src[0][0][0];
    const bool vec = a.vec_store != 0;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int i = i0 + o;
        if (i >= a.Ho) break;
        const double y = linspace_at(a.ly, i);
        const bool y_in = (unsigned long long)__double_as_longlong(y) <= (unsigned long long)__double_as_longlong(lg.Hm1);
        const double fy = fmin(floor(y), lg.Hm2);
        const double wy0 = 1.0 - (y - fy), wy1 = 1.0 - wy0;
        const int row0 = y_in ? ((int)fy - ylo) * kUSW : 0;
        float res[2][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool inside = y_in & x_in[k];
            const int o00 = inside ? row0 + xo[k] : 0;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const float *t = tile + p * (kUSH * kUSW) + o00;
                double acc, c;
                c = (double)t[0]; c = c * wy0; acc = c * wx0[k];
                c = (double)t[cstep]; c = c * wy0; c = c * wx1[k]; acc = acc + c;
                c = (double)t[rstep]; c = c * wy1; c = c * wx0[k]; acc = acc + c;
                c = (double)t[rstep + cstep]; c = c * wy1; c = c * wx1[k]; acc = acc + c;
                const float r = inside ? (float)acc : 0.0f;
                res[p][k] = r * a.scale[p];   // fp32 multiply by float32(scale), :135-136
            }
        }
        if (a.interleaved) {
            float2 *__restrict__ dst = reinterpret_cast<float2 *>(a.out[0]) + (size_t)img * op + (size_t)i * a.Wo + j0;
            if (vec) {   // (i * Wo + j0) * 8 bytes is a multiple of 32
                reinterpret_cast<float4 *>(dst)[0] = make_float4(res[0][0], res[1][0], res[0][1], res[1][1]);
                reinterpret_cast<float4 *>(dst)[1] = make_float4(res[0][2], res[1][2], res[0][3], res[1][3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (j0 + k < a.Wo) dst[k] = make_float2(res[0][k], res[1][k]);
            }
            continue;
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float *__restrict__ dst = a.out[p] + (size_t)img * op + (size_t)i * a.Wo + j0;
            if (vec) {
                *reinterpret_cast<float4 *>(dst) = make_float4(res[p][0], res[p][1], res[p][2], res[p][3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (j0 + k < a.Wo) dst[k] = res[p][k];
            }
        }
    }
}

// K3 standalone: warp_image (lucas_kanade_pyramidal.py:66-97)
__global__ __launch_bounds__(256) void k_warp(const float *__restrict__ img,
                                              const float *__restrict__ fu,
                                              const float *__restrict__ fv, float *__restrict__ out,
                                              int H, int W)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * (size_t)W;
    const size_t base = (size_t)blockIdx.z * plane;
    size_t i = (size_t)y * W + x;
    // same tap code as the fused iteration kernel (k_lk5), so the unit tests of
    // warp_image exercise it
    const LeanGeom lg = lean_geom(H, W);
    LeanTaps t = lean_taps(lg, y, x, fu[base + i], fv[base + i]);
    PairF r0, r1;
    lean_load<true>(lg, img + base, t, r0, r1);
    out[base + i] = lean_finish(t, r0, r1);
}

// ---------------------------------------------------------------------------
// BASELINE config 5: single-scale LK with fp16 gradients and fp16 accumulators (opt-in; NOT the
// reference's arithmetic -- the reference is fp32 throughout, lucas_kanade_core.py:110-133 -- so this
// mode is judged by its EPE against the exact result, tests/test_gpu_fp16.py, never by equality).
//
// What fp16 buys: two planes per instruction and, because exactness is given up anyway, SEPARABLE
// window sums -- which lets the kernel stream: NO LDS and NO barrier.  A wave owns a strip of image
// columns (k_lk16d: two columns per lane, 128 per wave) and walks down Hs rows of it:
//   per row   one coalesced load of each frame (requested PF rows ahead)
//             avg (fp32), It; Sobel/8 of the row above from three avg rows held in registers, the
//             x-neighbours through DPP wave shifts
//             products {IxIx, IyIy}, {IxIy, IxIt}, {IyIt, 0} (packed fp16) into a ring of 2HW+1 rows held
//             in registers (the loop is unrolled by the ring length: static slot indices)
//             vertical sums: three packed adds per row and register (prefix sums of the current block of
//             2HW+1 rows + suffix sums of the previous one, no subtraction; blocks aligned to absolute rows,
//             so a pixel's sums do not depend on where segments are cut); horizontal sums by wave shifts
//             fp32 solve and one coalesced store of u and v, HW + 1 rows behind the loads
// A segment of Hs rows costs 2R extra rows of loads (R = HW + 1).  77 VGPRs at 7x7: six waves per SIMD, which
// is what hides the memory latency here -- the exact tile kernel sits at four.
//
// Range: sum over (2HW+1)^2 taps of Ix^2 must stay below fp16's 65504.  Frames are scaled by powers
// of two on the way in (exact): gradients carry s_g, It carries s_t = s_g / 2, with
//   s_g = 2^-k,  k = smallest integer with  taps * (pixel_max/2 * s_g)^2 <= 60000
// (|Ix| <= pixel_max/2 for the Sobel/8 kernel, |It| <= pixel_max), so every window sum is bounded by
// 60000.  The solve runs in fp32 on the five fp16 sums: det' = s_g^4 det is tested against 1e-4 s_g^4,
// and u = 2 u' (the factor s_g / s_t).  Frames beyond [0, pixel_max] overflow to inf/nan by design.
// (Two earlier forms of the same arithmetic -- an LDS-tiled kernel and a one-column-per-lane streaming kernel,
// both slower -- are in the repository's history up to round 2.)
// ---------------------------------------------------------------------------
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

struct Lk16sArgs {
    const float *prev, *curr;   // [B][H][W]
    float *u, *v;
    int H, W, B;
    int Hs, segs;               // rows per segment, segments per strip
    float s_g, s_t;             // input scales (powers of two), see above
    float det_thr;              // 1e-4 * s_g^4
};

template <int CTRL>
__device__ __forceinline__ h2 wave_shift(h2 v)   // 0x138: lane i takes lane i-1's value; 0x130: lane i+1's
{
    return __builtin_bit_cast(h2, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float wave_shift(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <class F, int... J>
__device__ __forceinline__ void static_for(std::integer_sequence<int, J...>, F &&f)
{
    (f(std::integral_constant<int, J>{}), ...);
}

// TWO columns per lane: a wave covers 128 columns and produces 128 - 4 ceil(R/2) of
// them (5x5 and 7x7: 120, halo 6 % instead of 9 - 12.5 %), the five products are packed by COLUMN pair {lo, hi}, and the
// horizontal sums need ceil(HW/2) wave shifts per side and register.  Odd HW (K = (HW-1)/2): lanes l-K .. l+K
// contribute both columns, lane l-K-1 its high and lane l+K+1 its low column.  Even HW (K = HW/2): lanes l-K+1 .. l+K-1
// contribute both columns to both outputs; lane l-K gives both columns to the low output and its high column to the high
// output, lane l+K its low column to the low output and both to the high one.
// VEC (W even, planes 8-byte aligned: the host checks): every lane moves its column pair with one 8-byte access; pairs
// that lie outside the frame (halo lanes of the first and last strip) take the edge pair and repeat its edge column.
template <int HW, bool VEC>
__global__ __launch_bounds__(256) void k_lk16d(Lk16sArgs a)
{
    constexpr bool ODD = HW % 2 == 1;
    constexpr int R = HW + 1, S = 2 * HW + 1, HL = (R + 1) / 2, K = ODD ? (HW - 1) / 2 : HW / 2, OUTW = 2 * (64 - 2 * HL), PF = OFLK_LK16_PF, WPB = 4;
    constexpr int SHR = 0x138, SHL = 0x130;   // DPP wave_shr:1 / wave_shl:1
    const int lane = threadIdx.x & 63;
    const int H = a.H, W = a.W;
    const int strips = (W + OUTW - 1) / OUTW;
    const int nwave = strips * a.segs * a.B;
    const int nblk = (nwave + WPB - 1) / WPB;
    const int task = __builtin_amdgcn_readfirstlane(xcd_tile_index(blockIdx.x, nblk) * WPB + (int)(threadIdx.x >> 6));
    if (task >= nwave) return;
    const int b = task / (strips * a.segs);
    const int t = task - b * (strips * a.segs);
    const int seg = t / strips, strip = t - seg * strips;
    const int xw = strip * OUTW - 2 * HL;          // first column of the wave (uniform)
    const int x = xw + 2 * lane;                   // the lane's low column
    const unsigned cb0 = 4u * (unsigned)min(max(x, 0), W - 1), cb1 = 4u * (unsigned)min(max(x + 1, 0), W - 1);
    const unsigned cbp = 4u * (unsigned)min(max(x, 0), max(W - 2, 0));   // VEC: the pair's (even) byte offset, clamped into the row
    const int ys = seg * a.Hs, ye = min(ys + a.Hs, H);
    const size_t plane = (size_t)H * (size_t)W;
    const float *__restrict__ prev = a.prev + (size_t)b * plane;
    const float *__restrict__ curr = a.curr + (size_t)b * plane;
    float *__restrict__ ou = a.u + (size_t)b * plane;
    float *__restrict__ ov = a.v + (size_t)b * plane;
    const float ha = 0.5f * a.s_g, st = a.s_t;
    const bool lane_out = lane >= HL && lane < 64 - HL;
    const bool in0 = x >= HW && x < W - HW, in1 = x + 1 >= HW && x + 1 < W - HW;   // columns with a full window

    auto row_off = [&](int r) { return (size_t)min(max(r, 0), H - 1) * (size_t)W; };   // "symm" ring
    auto load2 = [&](const float *base, size_t o) {
        if constexpr (VEC) {
            const float2 w = ld_off<float2>(base + o, cbp);
            return x < 0 ? make_float2(w.x, w.x) : (x >= W ? make_float2(w.y, w.y) : w);   // "symm": the edge column repeats
        } else {
            return make_float2(ld_off<float>(base + o, cb0), ld_off<float>(base + o, cb1));
        }
    };
    const int r0 = ys - R + 2;
    float2 a0, a1, it1;
    {
        const size_t o0 = row_off(r0 - 2), o1 = row_off(r0 - 1);
        const float2 p0 = load2(prev, o0), q0 = load2(curr, o0), p1 = load2(prev, o1), q1 = load2(curr, o1);
        a0 = make_float2((p0.x + q0.x) * ha, (p0.y + q0.y) * ha);
        a1 = make_float2((p1.x + q1.x) * ha, (p1.y + q1.y) * ha);
        it1 = make_float2((p1.x - q1.x) * st, (p1.y - q1.y) * st);
    }
    float2 pb[PF], qb[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const size_t o = row_off(r0 + k);
        pb[k] = load2(prev, o);
        qb[k] = load2(curr, o);
    }
    const h2 zero2 = h2{(_Float16)0.0f, (_Float16)0.0f};
    h2 ring[S][5], fw[5] = {zero2, zero2, zero2, zero2, zero2};   // vertical sums: blocks aligned to absolute rows
#pragma unroll
    for (int j = 0; j < S; j++)
#pragma unroll
        for (int pl = 0; pl < 5; pl++) ring[j][pl] = zero2;

    const int n_it = ye - ys + 2 * HW;
    const int j0 = ((r0 - 1) % S + S) % S;
    for (int i0 = -j0; i0 < n_it; i0 += S) {
        static_for(std::make_integer_sequence<int, S>{}, [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int i = i0 + j;
            if (i < 0 || i >= n_it) return;   // uniform
            const int r = r0 + i;
            const float2 p = pb[0], q = qb[0];
#pragma unroll
            for (int k = 0; k + 1 < PF; k++) {
                pb[k] = pb[k + 1];
                qb[k] = qb[k + 1];
            }
            {
                const size_t o = row_off(r + PF);
                pb[PF - 1] = load2(prev, o);
                qb[PF - 1] = load2(curr, o);
            }
            const float2 a2 = make_float2((p.x + q.x) * ha, (p.y + q.y) * ha), itn = make_float2((p.x - q.x) * st, (p.y - q.y) * st);
            // Sobel/8 of row r - 1: column 2l has its neighbours in lane l-1 (high) and lane l (high); column 2l+1 in lane l
            // (low) and lane l+1 (low)
            const float smx = (a0.x + a2.x) + 2.0f * a1.x, smy = (a0.y + a2.y) + 2.0f * a1.y;
            const float dfx = a0.x - a2.x, dfy = a0.y - a2.y;
            const float ix0 = (wave_shift<SHR>(smy) - smy) * 0.125f, ix1 = (smx - wave_shift<SHL>(smx)) * 0.125f;
            const float iy0 = ((wave_shift<SHR>(dfy) + dfy) + 2.0f * dfx) * 0.125f, iy1 = ((dfx + wave_shift<SHL>(dfx)) + 2.0f * dfy) * 0.125f;
            const h2 hx = h2{(_Float16)ix0, (_Float16)ix1}, hy = h2{(_Float16)iy0, (_Float16)iy1}, ht = h2{(_Float16)it1.x, (_Float16)it1.y};
            h2 c[5];
            c[0] = hx * hx;   // IxIx of the two columns
            c[1] = hy * hy;
            c[2] = hx * hy;
            c[3] = hx * ht;
            c[4] = hy * ht;
            a0 = a1; a1 = a2; it1 = itn;
            h2 vs[5];
#pragma unroll
            for (int pl = 0; pl < 5; pl++) {
                fw[pl] = j == 0 ? c[pl] : fw[pl] + c[pl];
                vs[pl] = j == S - 1 ? fw[pl] : ring[(j + 1) % S][pl] + fw[pl];
                ring[j][pl] = c[pl];
            }
            if constexpr (j == S - 1) {
#pragma unroll
                for (int k = S - 2; k >= 1; k--)
#pragma unroll
                    for (int pl = 0; pl < 5; pl++) ring[k][pl] = ring[k][pl] + ring[k + 1][pl];
            }
            const int o = r - R;
            if (o >= ys) {
                float sum[5][2];
#pragma unroll
                for (int pl = 0; pl < 5; pl++) {
                    auto swap = [](h2 q) { const unsigned w = __builtin_bit_cast(unsigned, q); return __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(w, w, 16)); };   // {hi, lo}
                    auto hi_lo = [](h2 p0, h2 p1) {   // {p0.hi, p1.lo}
                        return __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(__builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p0), 16));
                    };
                    h2 l = vs[pl], rr = vs[pl], m = vs[pl], out;
                    if constexpr (ODD) {
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            l = wave_shift<SHR>(l);
                            rr = wave_shift<SHL>(rr);
                            m = m + (l + rr);
                        }
                        l = wave_shift<SHR>(l);     // lane l-K-1: its high column
                        rr = wave_shift<SHL>(rr);   // lane l+K+1: its low column
                        out = (m + swap(m)) + hi_lo(l, rr);
                    } else {
#pragma unroll
                        for (int k = 0; k + 1 < K; k++) {
                            l = wave_shift<SHR>(l);
                            rr = wave_shift<SHL>(rr);
                            m = m + (l + rr);
                        }
                        l = wave_shift<SHR>(l);     // lane l-K
                        rr = wave_shift<SHL>(rr);   // lane l+K
                        const h2 e1 = hi_lo(l, rr);                                              // {l.hi, rr.lo}: in both outputs
                        const h2 e2 = __builtin_bit_cast(h2, (__builtin_bit_cast(unsigned, l) & 0xFFFFu) | (__builtin_bit_cast(unsigned, rr) & 0xFFFF0000u));   // {l.lo, rr.hi}
                        out = ((m + swap(m)) + (e1 + swap(e1))) + e2;
                    }
                    sum[pl][0] = (float)out.x;
                    sum[pl][1] = (float)out.y;
                }
                float uu[2], vv[2];
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    const float Sxx = sum[0][cc], Syy = sum[1][cc], Sxy = sum[2][cc], Sxt = sum[3][cc], Syt = sum[4][cc];
                    const float det = Sxx * Syy - Sxy * Sxy;
                    const float inv = __builtin_amdgcn_rcpf(det) * 2.0f;   // s_g / s_t = 2
                    const bool solve = fabsf(det) > a.det_thr && (cc ? in1 : in0) && o >= HW && o < H - HW;   // borders stay 0 (:101-108)
                    uu[cc] = solve ? (Sxy * Syt - Syy * Sxt) * inv : 0.0f;
                    vv[cc] = solve ? (Sxy * Sxt - Sxx * Syt) * inv : 0.0f;
                }
                if (lane_out) {
                    const size_t orow = (size_t)o * (size_t)W;
                    if constexpr (VEC) {
                        if (x < W) {   // x >= 0 for output lanes; W even: the pair is inside
                            st_off<float2>(ou + orow, 4u * (unsigned)x, make_float2(uu[0], uu[1]));
                            st_off<float2>(ov + orow, 4u * (unsigned)x, make_float2(vv[0], vv[1]));
                        }
                    } else {
                        if (x >= 0 && x < W) {
                            st_off<float>(ou + orow, 4u * (unsigned)x, uu[0]);
                            st_off<float>(ov + orow, 4u * (unsigned)x, vv[0]);
                        }
                        if (x + 1 >= 0 && x + 1 < W) {
                            st_off<float>(ou + orow, 4u * (unsigned)(x + 1), uu[1]);
                            st_off<float>(ov + orow, 4u * (unsigned)(x + 1), vv[1]);
                        }
                    }
                }
            }
        });
    }
}

// ---------------------------------------------------------------------------
// RTL-bit-accurate integer mode: what the reference's single-scale RTL computes
// for a frame pair, per element k of its gradient stream -- rtl/common/line_buffer_5x5.sv:75-151 (window
// geometry), rtl/unopt/gradient_compute.sv:89-139, window_accumulator.sv:100-189, flow_solver.sv:82-149 --
// in S8.7 fixed point.  The geometry is the RTL's, quirks included (see oracle/rtl_model.py, which states it
// and is held equal to a cycle-by-cycle execution of the modules): the gradient of stream position (r, c),
// r, c >= 4, is taken on the 3x3 neighbourhood of pixels (r-2 .. r, c-4 .. c-2) and is zero for c = W-1; the
// valid gradients form a stream of (H-4)(W-4) elements that the accumulator wraps at W, and element k's 5x5
// window is  rows 0..3 x columns 0..3: k - (3-i) W - (4-j)  (zero when k mod W = W-1),
//            rows 0..3, column 4:      k - (4-i) W,          row 4: k - (4-j).
// A block takes 2048 consecutive elements of one pair: the 4 W + 2048 gradients they touch are computed
// into LDS ({gx, gy, gt} as one 8-byte cell), then every thread sums the windows of two groups of four adjacent
// elements (52 cells read for 4 x 25) and solves.  Integer work, a few
// bytes per pixel: nothing here is shaped for the matrix cores.
// PARITY UNPINNED: no output of the RTL as committed exists.
// ---------------------------------------------------------------------------
constexpr int kRtlChunk = 2048, kRtlMaxW = 1024, kRtlMaxH = 512;   // flow_x / flow_y are 10 / 9 bits wide (flow_solver.sv:34-37)

struct RtlArgs {
    const unsigned char *prev, *curr;   // [B][H][W]
    short *u, *v;                       // [B][(H-4)(W-4)]  S8.7
    int H, W, B;
};

// (num <<< 7) / det as Verilog's signed `/` computes it (truncation towards zero, flow_solver.sv:121-122), in fp64:
// |num * 128| < 2^39 and 1000 < |det| < 2^31 are exact doubles, `rdet` ~ 1 / det to fp64 accuracy (shared by the two
// quotients of an element), so trunc(a * rdet) is within one of the quotient; the remainder a - q det is an integer
// below 2^33 in magnitude, which the fma returns exactly, and it settles the last unit.  |q| < 2^29 fits an int.
__device__ __forceinline__ double rtl_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ int rtl_trunc_div(int num, double det, double rdet)
{
    const double a = (double)num * 128.0;
    double q = __builtin_trunc(a * rdet);
    const double r = __builtin_fma(-q, det, a);          // exact
    // the quotient truncates towards zero: the remainder must carry the sign of `a` and be smaller than |det|
    const double ad = __builtin_fabs(det), s = (a < 0.0) != (det < 0.0) ? -1.0 : 1.0;
    const double ra = a < 0.0 ? -r : r;                  // remainder measured in the direction of a
    if (ra < 0.0) q -= s;                                // overshot: one step back towards zero
    else if (ra >= ad) q += s;                           // undershot
    return (int)q;
}

__global__ __launch_bounds__(256) void k_rtl_flow(RtlArgs a)
{
    // gradients of stream elements g_lo .. g_lo + L - 1 as {gx, gy, gt, 0}: one 8-byte LDS read per window element
    extern __shared__ __attribute__((aligned(16))) short4 s_g[];
    const int tid = threadIdx.x;
    const int W = a.W, H = a.H, Wv = W - 4;
    const int M = (H - 4) * Wv;
    const int b = blockIdx.y;
    const int k_lo = blockIdx.x * kRtlChunk, g_lo = k_lo - 4 * W, L = 4 * W + kRtlChunk;
    const unsigned char *__restrict__ prev = a.prev + (size_t)b * H * W;
    const unsigned char *__restrict__ curr = a.curr + (size_t)b * H * W;

    // ---- gradients (gradient_compute.sv:108-139): a thread takes four adjacent stream elements; when they lie in one
    // image row (all but the groups that straddle a row end) their 3 x 6 pixels come as one 8-byte load per row and
    // frame instead of 18 byte loads per element ----
    auto gradient = [&](auto px) {   // px(frame, i, j): pixel of the element's 3x3 neighbourhood, frame 0 = curr, 1 = prev
        int avg[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int sc = (signed char)px(0, i, j), sp = (signed char)px(1, i, j);   // `logic signed [7:0]` ports
                avg[i][j] = ((sc + sp) & 0x1FF) >> 1;                                     // 9-bit sum, logical shift
            }
        const int xl = -avg[0][0] - (avg[1][0] << 1) - avg[2][0], xr = avg[0][2] + (avg[1][2] << 1) + avg[2][2];
        const int yt = -avg[0][0] - (avg[0][1] << 1) - avg[0][2], yb = avg[2][0] + (avg[2][1] << 1) + avg[2][2];
        const int gt = (int)px(1, 1, 1) - (int)px(0, 1, 1);   // zero-extended pixels (:139)
        return make_short4((short)((xl + xr) >> 3), (short)((yt + yb) >> 3), (short)gt, 0);   // >>> of a signed sum
    };
    const int m_a = g_lo > 0 ? g_lo : 0, m_b = (g_lo + L < M ? g_lo + L : M) - 1;   // stream elements of this block that exist
    for (int e = tid; e < L; e += 256)   // cells before the stream's start / past its end read as zero
        if (g_lo + e < 0 || g_lo + e >= M) s_g[e] = make_short4(0, 0, 0, 0);
    if (W >= 8) {
        // groups of four elements of one image row (the row's last group may be shorter): every group takes the 8-byte path
        const int gpr = (Wv + 3) >> 2, row_a = m_a / Wv, nrow = m_b / Wv - row_a + 1;
#pragma unroll 1
        for (int t = tid; t < nrow * gpr; t += 256) {
            const int rw = row_a + t / gpr, cg = t - (t / gpr) * gpr;
            const int r = rw + 4, c = 4 * cg + 4;            // stream position of the group's first window
            const int start = min(c - 4, W - 8), sh = 8 * (c - 4 - start);   // the row's last group reads from W-8 and shifts
            unsigned long long w[2][3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                __builtin_memcpy(&w[0][i], curr + (r - 2 + i) * W + start, 8);   // pixels (r-2 .. r, c-4 ..): the stream runs one
                __builtin_memcpy(&w[1][i], prev + (r - 2 + i) * W + start, 8);   // pixel behind the frame
                w[0][i] >>= sh;
                w[1][i] >>= sh;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cc = c + q, m = rw * Wv + 4 * cg + q;
                if (cc < W && m >= m_a && m <= m_b) {
                    const short4 g = gradient([&](int f, int i, int j) { return (unsigned char)(w[f][i] >> (8 * (q + j))); });
                    // the window of a row's last position has only its newest column (line_buffer_5x5.sv:101-131)
                    s_g[m - g_lo] = cc == W - 1 ? make_short4(0, 0, 0, 0) : g;
                }
            }
        }
    } else {
#pragma unroll 1
        for (int e = tid; e < L; e += 256) {
            const int m = g_lo + e;
            if (m < 0 || m >= M) continue;
            const int rr = m / Wv + 4, cc = m % Wv + 4;
            s_g[e] = cc == W - 1 ? make_short4(0, 0, 0, 0)
                                 : gradient([&](int f, int i, int j) { return (f ? prev : curr)[(rr - 2 + i) * W + (cc - 4 + j)]; });
        }
    }
    __syncthreads();

    // ---- window sums and solve: a thread takes groups of four ADJACENT elements, whose windows share most cells ----
#pragma unroll 1
    for (int q = 0; q < kRtlChunk / 1024; q++) {
        const int kb = k_lo + 1024 * q + 4 * tid;   // first element of the group
        if (kb >= M) break;
        const int base = kb - g_lo;                 // LDS index of element kb (>= 4 W)
        // five sums per element: first the 4 x 4 block of rows 0..3 / columns 0..3, which is zero as a whole when
        // the element is a row's last position, then column 4 and row 4 on top of it
        int sm[4][5];
#pragma unroll
        for (int o = 0; o < 4; o++)
#pragma unroll
            for (int p = 0; p < 5; p++) sm[o][p] = 0;
        auto add = [](int (&acc)[5], short4 g) {
            const int x = g.x, y = g.y, t = g.z;
            acc[0] += x * x;   // 25 products of 12-bit values: 32 bits never overflow (window_accumulator.sv:128-166)
            acc[1] += y * y;
            acc[2] += x * y;
            acc[3] += x * t;
            acc[4] += y * t;
        };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // index kb + o - (3-i) W - (4-j): seven cells serve the four elements
            short4 row[7];
#pragma unroll
            for (int c = 0; c < 7; c++) row[c] = s_g[base - (3 - i) * W - 4 + c];
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int j = 0; j < 4; j++) add(sm[o], row[o + j]);
            __builtin_amdgcn_sched_barrier(0);   // one window row's cells in registers at a time
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const bool edge = (kb + o) % W == W - 1;
#pragma unroll
            for (int p = 0; p < 5; p++) sm[o][p] = edge ? 0 : sm[o][p];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)   // column 4: index kb + o - (4-i) W
#pragma unroll
            for (int o = 0; o < 4; o++) add(sm[o], s_g[base + o - (4 - i) * W]);
        __builtin_amdgcn_sched_barrier(0);
        {
            short4 row[8];   // row 4: kb + o - (4-j)
#pragma unroll
            for (int c = 0; c < 8; c++) row[c] = s_g[base - 4 + c];
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int j = 0; j < 5; j++) add(sm[o], row[o + j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        int sxx[4], syy[4], sxy[4], sxt[4], syt[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            sxx[o] = sm[o][0]; syy[o] = sm[o][1]; sxy[o] = sm[o][2]; sxt[o] = sm[o][3]; syt[o] = sm[o][4];
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int k = kb + o;
            if (k >= M) break;
            // flow_solver.sv:82-149: products keep their low 32 bits
            auto lo = [](int p, int r) { return (int)((unsigned)p * (unsigned)r); };
            const int det = (int)((unsigned)lo(sxx[o], syy[o]) - (unsigned)lo(sxy[o], sxy[o]));
            const int nu = (int)((unsigned)lo(syy[o], sxt[o]) - (unsigned)lo(sxy[o], syt[o]));
            const int nv = (int)((unsigned)lo(sxx[o], syt[o]) - (unsigned)lo(sxy[o], sxt[o]));
            int fu = 0, fv = 0;
            if (det > 1000 || det < -1000) {
                const double dd = (double)det, rd = rtl_recip(dd);
                fu = (short)rtl_trunc_div(nu, dd, rd);   // 39-bit quotient truncated towards zero, low 16 bits kept
                fv = (short)rtl_trunc_div(nv, dd, rd);
                fu = fu > 1024 ? 1024 : (fu < -1024 ? -1024 : fu);
                fv = fv > 1024 ? 1024 : (fv < -1024 ? -1024 : fv);
            }
            a.u[(size_t)b * M + k] = (short)fu;
            a.v[(size_t)b * M + k] = (short)fv;
            __builtin_amdgcn_sched_barrier(0);   // one element's 64-bit arithmetic at a time
        }
    }
}

// a1 standalone: compute_gradients (lucas_kanade_core.py:15-45)
__global__ __launch_bounds__(256) void k_gradients(const float *__restrict__ prev,
                                                   const float *__restrict__ curr,
                                                   float *__restrict__ Ix, float *__restrict__ Iy,
                                                   float *__restrict__ It, int H, int W)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t base = (size_t)blockIdx.z * (size_t)H * (size_t)W;
    auto avg = [&](int yy, int xx) -> float {
        yy = min(max(yy, 0), H - 1);
        xx = min(max(xx, 0), W - 1);
        size_t i = base + (size_t)yy * W + xx;
        float s = prev[i] + curr[i];
        return s * 0.5f;
    };
    float a_mm = avg(y - 1, x - 1), a_m0 = avg(y - 1, x), a_mp = avg(y - 1, x + 1);
    float a_0m = avg(y, x - 1), a_0p = avg(y, x + 1);
    float a_pm = avg(y + 1, x - 1), a_p0 = avg(y + 1, x), a_pp = avg(y + 1, x + 1);
    float ix = a_pp * -0.125f;
    ix = fmaf(a_pm, 0.125f, ix);
    ix = fmaf(a_0p, -0.25f, ix);
    ix = fmaf(a_0m, 0.25f, ix);
    ix = fmaf(a_mp, -0.125f, ix);
    ix = fmaf(a_mm, 0.125f, ix);
    float iy = a_pp * -0.125f;
    iy = fmaf(a_p0, -0.25f, iy);
    iy = fmaf(a_pm, -0.125f, iy);
    iy = fmaf(a_mp, 0.125f, iy);
    iy = fmaf(a_m0, 0.25f, iy);
    iy = fmaf(a_mm, 0.125f, iy);
    size_t i = base + (size_t)y * W + x;
    Ix[i] = ix;
    Iy[i] = iy;
    It[i] = prev[i] - curr[i];
}

// uint8 frames -> float32 (the conversion the verifier does on the host,
// optical_flow_verifier.py:61-65; exact for every uint8 value).  16 pixels per thread.
__global__ __launch_bounds__(256) void k_u8_to_f32(const unsigned char *__restrict__ in, float *__restrict__ out,
                                                   size_t n)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i + 16 <= n && ((reinterpret_cast<uintptr_t>(in + i) & 15) == 0)) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float4 f = make_float4((float)(w[k] & 255u), (float)((w[k] >> 8) & 255u), (float)((w[k] >> 16) & 255u),
                                   (float)(w[k] >> 24));
            *reinterpret_cast<float4 *>(out + i + 4 * k) = f;
        }
    } else {
        for (int k = 0; k < 16; k++)
            if (i + k < n) out[i + k] = (float)in[i + k];
    }
}

// de-interleave the finest-level flow of pairs whose result was not written to the caller's planar
// buffers by the level's last launch (early exit); no-op blocks otherwise
struct ExportArgs {
    const float2 *src[2];        // the finest level's two interleaved {u, v} slots [B][H][W]
    float *dst_u, *dst_v;        // caller's planar buffers
    const unsigned long long *acc;
    int want;                    // unused (kept for layout)
    int L, K;                    // K: accumulator / log layout (>= 1)
    int iters;                   // iterations launched per level
    double counts[OFLK_MAX_LEVELS];             // H*W per level
    unsigned long long thr[OFLK_MAX_LEVELS];    // early-exit thresholds as totals
    float *log;                  // [B][L][K][2]
    int *iters_run;              // [B][L]
    int *uncertain;              // [B][L]: bit k set = the exit decision after iteration k was taken within kDecisionGuard of the threshold
    unsigned long long guard_lo[OFLK_MAX_LEVELS], guard_hi[OFLK_MAX_LEVELS];   // totals bounding that band
    size_t plane;
};

// The early-exit test compares np.mean(np.abs(d)) -- an fp32 pairwise sum in 8192-element pieces added
// up one after the other (error bound ~ pieces x 2^-24 relative) -- with float32(0.01); the device
// compares an exactly accumulated fixed-point total instead.  The two decisions can only differ when a
// mean lies within the summation error of the threshold: every decision taken within +-kDecisionGuard
// (relative) of it is reported, so a caller can tell "provably the reference's decision" from "too
// close to call" (never seen outside constructed inputs; tests/test_gpu_round2.py builds them).
// The band follows the level's size: NumPy adds ceil(n / 8192) pieces one after the other (each itself a
// pairwise sum, a few 2^-24), so its worst-case error grows with the pixel count and passes 5e-5 at ~6.9 Mpx
// (a 4K finest level: 6.2e-5, 8K: 2.4e-4).
constexpr double kDecisionGuard = 5e-5;   // floor of the band
inline double decision_guard(double npix)
{
    const double pieces = std::ceil(npix / 8192.0);
    return std::max(kDecisionGuard, (pieces + 32.0) * 5.9604644775390625e-08);   // 2^-24
}

// End of a pyramidal call.  (1) Pairs whose finest level exited early hold their result in
// the internal ping-pong slot: copy it to the caller's buffers.  (2) Block 0 of each pair
// materialises the residual log and the per-level iteration counts from the accumulators.
__global__ __launch_bounds__(256) void k_export_fixup(ExportArgs a)
{
    const int b = blockIdx.y;
    for (int e = threadIdx.x; blockIdx.x == 0 && e < a.L * a.K; e += 256) {
        // one thread per (level, iteration) log entry
        const int l = e / a.K, k = e - l * a.K;
        const LevelState st = lk_level_state(a.acc, b, l, a.iters, a.L, a.K, a.thr[l]);
        if (k == 0) a.iters_run[b * a.L + l] = st.executed;
        float mu = 0.0f, mv = 0.0f;
        if (k < st.executed) {
            unsigned long long tu, tv;
            lk_totals(a.acc, b, l, k, a.L, a.K, tu, tv);
            mu = lk_mean_of(tu, a.counts[l]);
            mv = lk_mean_of(tv, a.counts[l]);
            // decision "both below": it could have gone the other way iff one mean is inside the band
            // while the other is not clearly above it
            const bool near_u = tu >= a.guard_lo[l] && tu <= a.guard_hi[l], near_v = tv >= a.guard_lo[l] && tv <= a.guard_hi[l];
            if ((near_u && tv <= a.guard_hi[l]) || (near_v && tu <= a.guard_hi[l])) atomicOr(&a.uncertain[b * a.L + l], 1 << k);
        }
        const size_t li = (((size_t)b * a.L + l) * a.K + k) * 2;
        a.log[li] = mu;
        a.log[li + 1] = mv;
    }
    // the finest level's last launch (iteration iters-1) writes the caller's planar planes itself; a pair
    // that left the level earlier (or a plan without iterations) still holds its result in an interleaved slot
    __shared__ int s_exec;
    if (threadIdx.x == 0) s_exec = lk_level_state(a.acc, b, a.L - 1, a.iters, a.L, a.K, a.thr[a.L - 1]).executed;
    __syncthreads();
    if (a.iters >= 1 && s_exec == a.iters) return;
    const float2 *__restrict__ src = a.src[s_exec & 1] + (size_t)b * a.plane;
    const size_t base = (size_t)b * a.plane;
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.plane; i += step) {
        const float2 f = src[i];
        a.dst_u[base + i] = f.x;
        a.dst_v[base + i] = f.y;
    }
}

// ---------------------------------------------------------------------------
// Exact residual means for the rare exit decision taken too close to the threshold (kDecisionGuard):
// np.mean(np.abs(d)) as NumPy evaluates it (lucas_kanade_pyramidal.py:213-214) -- an fp32 pairwise sum
// (eight strided accumulators per block of <= 128 elements, blocks combined by halving) inside
// 8192-element pieces of the flat array, the pieces added up one after the other, then
// float32(float64(sum) / n).  One thread per piece (this is a slow path: host-driven, one pair at a
// time, oflk_plan_resolve_uncertain); the pieces are added on the host in order.
// ---------------------------------------------------------------------------
constexpr int kNpPiece = 8192;   // np.getbufsize()

__device__ __noinline__ float np_pairwise_abs_sum(const float *a, size_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (size_t i = 0; i < n; i++) res = res + fabsf(a[i]);
        return res;
    }
    if (n <= 128) {
        float r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = fabsf(a[j]);
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + fabsf(a[i + j]);
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + fabsf(a[i]);
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_abs_sum(a, n2) + np_pairwise_abs_sum(a + n2, n - n2);
}

__global__ __launch_bounds__(64) void k_np_abs_piece_sums(const float *__restrict__ d, size_t n, float *__restrict__ out)
{
    const size_t c = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t begin = c * kNpPiece;
    if (begin >= n) return;
    const size_t m = n - begin < (size_t)kNpPiece ? n - begin : (size_t)kNpPiece;
    out[c] = np_pairwise_abs_sum(d + begin, m);
}

// flow += d (lucas_kanade_pyramidal.py:209-210), planar, for the same slow path
__global__ __launch_bounds__(256) void k_flow_add(float *__restrict__ fu, float *__restrict__ fv, const float *__restrict__ du,
                                                  const float *__restrict__ dv, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        fu[i] = fu[i] + du[i];
        fv[i] = fv[i] + dv[i];
    }
}

// ---------------------------------------------------------------------------
// F1: masked flow metrics on the device (python/flow_metrics.py:14-201) for the rectangular
// test regions the verifier uses (mask[y0:y1, x0:x1] = True, optical_flow_verifier.py:96-138),
// so that a batch run need not copy flow fields to the host.
//
// Per pixel the reference's own fp32 operations (error components, squares, square roots,
// the (u, v, 1) dot product and norms); arccos and the sums in fp64.  The reference takes
// fp32 pairwise means and NumPy's fp32 arccos, so the two agree to ~1e-6 relative, not bit
// for bit (tolerance stated in tests/test_gpu_metrics.py).
// grid (kMetricBlocks, B); out: [B][kMetricBlocks][kMetricTerms] partial sums, added up in a fixed
// order on the host.
// ---------------------------------------------------------------------------
constexpr int kMetricBlocks = 64;
constexpr int kMetricTerms = 6;   // sum|eu|, sum|ev|, sum(eu^2+ev^2), sum sqrt(..), sum angle [deg], max |pred|

struct MetricArgs {
    const float *u, *v;      // [B][H][W]
    const float *u_true;     // [B] device
    const float *v_true;     // [B]
    int H, W;
    int y0, y1, x0, x1;      // mask rectangle, already clipped to the frame
    double *partial;         // [B][kMetricBlocks][kMetricTerms]
};

__global__ __launch_bounds__(256) void k_flow_metrics(MetricArgs a)
{
    const int b = blockIdx.y;
    const size_t plane = (size_t)a.H * (size_t)a.W;
    const float *__restrict__ u = a.u + (size_t)b * plane;
    const float *__restrict__ v = a.v + (size_t)b * plane;
    const float ut = a.u_true[b], vt = a.v_true[b];
    const int rw = a.x1 - a.x0, rh = a.y1 - a.y0;
    const size_t n = (size_t)max(rw, 0) * (size_t)max(rh, 0);
    // |(ut, vt, 1)|: fp32 ops in the reference's order (flow_metrics.py:150)
    const float norm_t = sqrtf(ut * ut + vt * vt + 1.0f);
    double s[kMetricTerms] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)kMetricBlocks * 256) {
        const int y = a.y0 + (int)(e / (size_t)rw), x = a.x0 + (int)(e % (size_t)rw);
        const float up = u[(size_t)y * a.W + x], vp = v[(size_t)y * a.W + x];
        const float eu = up - ut, ev = vp - vt;           // :31-32
        const float sq = eu * eu + ev * ev;               // :66, :99
        const float mag2 = up * up + vp * vp;
        const float norm_p = sqrtf(mag2 + 1.0f);          // :149
        float c = (up * ut + vp * vt + 1.0f) / (norm_p * norm_t);   // :153-155
        c = fminf(fmaxf(c, -1.0f), 1.0f);                 // :158
        s[0] += (double)fabsf(eu);
        s[1] += (double)fabsf(ev);
        s[2] += (double)sq;
        s[3] += (double)sqrtf(sq);
        s[4] += acos((double)c) * 57.29577951308232;      // :161-162
        s[5] = fmax(s[5], (double)sqrtf(mag2));           // :143 (for the "nothing moves" case)
    }
    __shared__ double red[kMetricTerms][256];
#pragma unroll
    for (int t = 0; t < kMetricTerms; t++) red[t][threadIdx.x] = s[t];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int t = 0; t < kMetricTerms - 1; t++) red[t][threadIdx.x] += red[t][threadIdx.x + off];
            red[kMetricTerms - 1][threadIdx.x] =
                fmax(red[kMetricTerms - 1][threadIdx.x], red[kMetricTerms - 1][threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x < kMetricTerms)
        a.partial[((size_t)b * kMetricBlocks + blockIdx.x) * kMetricTerms + threadIdx.x] = red[threadIdx.x][0];
}

}  // namespace oflk
