// oflk_stream.hpp -- k_lks: the streaming form of the fused Lucas-Kanade kernel (5x5 window; 7x7 single-scale), gfx950.
//
// k_lkw (oflk_kernels.hpp) stages tiles in LDS because NumPy's summation order of a 5x5 window
// (lucas_kanade_core.py:115-119) has no separable form.  Where the order of the 25 additions is free, the window
// sums separate -- five rows added vertically, five columns horizontally -- and the kernel can STREAM: no LDS tile, no
// barrier.  A wave owns a strip of 128 image columns, two per lane, and walks down a segment of rows:
//   per row   coalesced loads of prev and the flow (three rows ahead) and a coalesced read of `curr` that brings the lines
//             the warp will gather from into the L2; the bilinear gathers of `curr` for the row after this one are issued
//             before this row's arithmetic and used after it
//             warp (ITER), frame average, It; Sobel/8 of the row above in convolve2d's own tap order, from three
//             average rows held in registers, the x-neighbours through DPP wave shifts (the gradients are the
//             reference's, bit for bit)
//             five products; vertical 5-sums  ((p[o-2] + p[o-1]) + (p[o] + p[o+1])) + p[o+2]  as three adds per row
//             and plane from a ring of pair sums; horizontal 5-sums by four wave shifts per plane:
//                 even column c:  ((V[c-2] + V[c-1]) + (V[c] + V[c+1])) + V[c+2]
//                 odd column c:   (V[c-2] + (V[c-1] + V[c])) + (V[c+1] + V[c+2])
//             2x2 solve in the reference's operation sequence (IEEE division), flow += d (the flow of the output row comes
//             back from a per-wave LDS ring of the last eight flow rows: same lane writes and reads), |d| sums
// A wave produces 120 of its 128 columns (window halo 2 + Sobel halo 1, rounded to lane pairs); a segment of Hs
// rows costs 6 extra rows.  Three things decide its speed, all found in the ISA: wide loads must stay
// whole (selects on a loaded value sit where it is USED), loads are issued oldest-needed first because results return in
// issue order, and the unrolled row body has no early exit, so that the compiler's wait counts leave the younger loads in
// flight.
//
// Uses:
//  * OFLK_ARITH_TOLERANT (opt-in): the iterations of the two finest pyramid levels; with UPS the first iteration of a level
//    also upsamples the coarser level's flow on the fly (upsample_flow in the fused-lerp form).  The sums are NOT in NumPy's
//    order, so flows are close to, not equal to, the reference's: oracle/oflk_tolerant_model.c states this arithmetic
//    on the CPU, tests hold this kernel to it bit for bit and the model to the reference-made dense flows within
//    the 1e-4 bar (tools/experiments/fast_mode_ablation.py: what each cell of the pass costs).
//  * MODE_SINGLE on integer-valued frames (EXACT; 5x5 and, HW = 3, 7x7): gradients are multiples of 1/16 there, products of
//    2^-8 (2^-4 with It), so while a window's sums stay below 2^16 every partial sum is exact in any order and the separable
//    sums ARE NumPy's (proof at kLksExactBound).  A wave checks Sxx, Syy < 2^16 on every window it solves -- and, for
//    float32 frames, that every pixel it loads is an integer in [0, 255] -- and lists the 64 x 24 tiles where that fails;
//    the tile kernel k_lkw then redoes exactly those tiles in NumPy's order (LkArgs::redo).
#pragma once

namespace oflk {

enum { WARP_SCIPY = 0, WARP_LERP64 = 1 };

// fused form of the bilinear sample: three lerps, each one fma.  The fp64 result differs from SciPy's 15-operation
// sum by a few 1e-16 relative before it is rounded to float32 where SciPy rounds.
__device__ __forceinline__ float lerp64_finish(const LeanFrac &t, PairF r0, PairF r1)
{
    const double a = (double)r0.a, b = (double)r0.b, c = (double)r1.a, d = (double)r1.b;
    const double top = __builtin_fma(t.rx, b - a, a), bot = __builtin_fma(t.rx, d - c, c);
    const double r = __builtin_fma(t.ry, bot - top, top);
    return t.inside ? (float)r : 0.0f;
}

template <int CTRL>
__device__ __forceinline__ float wshift(float v)   // 0x138: lane i takes lane i-1's value; 0x130: lane i+1's
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

constexpr int kLksOutW = 120;   // output columns per wave
#ifndef OFLK_LKS_PFROWS
#define OFLK_LKS_PFROWS 0     // the prefetch of `curr` runs this many rows ahead of the coalesced loads
#endif
#ifndef OFLK_LKS_LD_SINGLE
#define OFLK_LKS_LD_SINGLE 3
#endif
#ifndef OFLK_LKS_LD_ITER
#define OFLK_LKS_LD_ITER 3
#endif
#ifndef OFLK_LKS_WAVES
#define OFLK_LKS_WAVES 2      // waves per SIMD the kernel's register count allows (185 - 193 VGPRs; the host sizes segments with it)
#endif

// MODE_SINGLE on integer-valued frames p, q in [0, 255] (lucas_kanade_core.py:36-43, :110-119): avg = (p + q)/2 is a
// multiple of 1/2, Ix and Iy (Sobel weights 1/8, 1/4) multiples of 1/16 with |Ix| <= 127.5, It = p - q an integer with
// |It| <= 255 -- all exact.  The products are exact too (11-bit x 11-bit, 11-bit x 9-bit significands): Ix*Ix, Iy*Iy, Ix*Iy
// multiples of 2^-8, Ix*It, Iy*It multiples of 2^-4.  A float32 sum of multiples of g is exact while it stays below 2^24 g.
// With B = Sxx, Syy < 2^16:
//   every partial sum of the non-negative terms of Sxx (Syy) is <= Sxx (Syy) < 2^16 = 2^24 * 2^-8                  -> exact
//   every partial sum of Sxy is bounded by sum |Ix Iy| <= (Sxx + Syy) / 2 < 2^16                                    -> exact
//   every partial sum of Sxt by sum |Ix It| <= sqrt(Sxx * sum It^2) <= sqrt(2^16 * 25 * 255^2) = 326 400 < 2^20 = 2^24 * 2^-4 -> exact
//                               (7x7: sqrt(2^16 * 49 * 255^2) = 456 960 < 2^20)
// so the five sums are the exact real numbers in ANY order of additions, hence equal to np.sum's; the solve that follows
// is the reference's operation sequence on equal inputs.  A window that fails the bound (or a pixel that is not such an
// integer) flags its tile for the NumPy-order kernel.
constexpr float kLksExactBound = 65536.0f;

// timing-only ablations (diagnostic builds, wrong results): 1 no gathers (warp from the coalesced prefetch), 2 no flow
// re-read, 4 no stores, 8 no solve, 16 no horizontal sums, 32 no lerp arithmetic
#if defined(OFLK_LKS_ABL) && !defined(OFLK_DIAG)
#error "OFLK_LKS_ABL is a diagnostic build: add -DOFLK_DIAG"
#endif
#ifndef OFLK_LKS_ABL
#define OFLK_LKS_ABL 0
#endif
#ifdef OFLK_LKS_FORCE_WAVES
#define OFLK_LKS_ATTR __attribute__((amdgpu_waves_per_eu(OFLK_LKS_FORCE_WAVES)))
#else
#define OFLK_LKS_ATTR
#endif
struct __attribute__((packed, aligned(8))) Flow2 { float u0, v0, u1, v1; };   // two adjacent {u, v} cells at an 8-byte aligned address

// HW = 2 (5x5) everywhere; HW = 3 (7x7) for MODE_SINGLE only: there the order of the 49 additions does not matter on 8-bit
// frames (same bound: sum |Ix It| <= sqrt(2^16 * 49 * 255^2) = 456 960 < 2^20), so the association of the sums below is free.
template <int MODE, bool VEC, int WARPV, class PIX = float, bool UPS = false, int HW = 2>
__global__ __launch_bounds__(256) OFLK_LKS_ATTR void k_lks(LkArgs a)
{
    static_assert(MODE == MODE_SINGLE || MODE == MODE_ITER, "gradient planes take the tile kernel");
    static_assert(!UPS || MODE == MODE_ITER, "the fused flow upsampling belongs to an iteration");
    static_assert(HW == 2 || (HW == 3 && MODE == MODE_SINGLE), "7x7 streams only where the summation order is provably irrelevant");
    constexpr int R = HW + 1, HL = 2, OUTW = kLksOutW, WPB = 4;   // halo R = 3 or 4 columns: two lanes either side
    constexpr int QR = 2 * HW - 1 == 3 ? 3 : 6;                  // period of the ring of vertical pair sums
    constexpr int SHR = 0x138, SHL = 0x130;   // DPP wave_shr:1 / wave_shl:1
    const int lane = threadIdx.x & 63;
    const int H = a.H, W = a.W;
    const int strips = (W + OUTW - 1) / OUTW;
    const int per_pair = strips * a.segs;
    const int nwave = per_pair * a.B;
    const int nblk = (nwave + WPB - 1) / WPB;
    const int task = __builtin_amdgcn_readfirstlane(xcd_tile_index(blockIdx.x, nblk) * WPB + (int)(threadIdx.x >> 6));
    if (task >= nwave) return;
    const int b = task / per_pair;
    const int tt = task - b * per_pair;
    const int seg = tt / strips, strip = tt - seg * strips;
    int sel = 0;
    if (MODE == MODE_ITER) {
        const LevelState st = lk_level_state(a.acc, b, a.level, a.iter, a.L, a.K, a.conv_thr);
        if (st.done) return;
        sel = st.executed & 1;
    }
    const int xw = strip * OUTW - 2 * HL;          // first column of the wave (uniform, even)
    const int x = xw + 2 * lane;                   // the lane's low column (even)
    const int Wm1 = W - 1, Hm1 = H - 1;
    const int c0 = min(max(x, 0), Wm1), c1 = min(max(x + 1, 0), Wm1);   // "symm" ring: the edge column repeats
    const int cp = min(max(x, 0), max(W - 2, 0));                       // VEC: first column of the lane's pair (even)
    const bool dup_lo = x < 0, dup_hi = x >= W;
    const int ys = seg * a.Hs, ye = min(ys + a.Hs, H);
    const size_t plane = (size_t)H * (size_t)W;
    const PIX *__restrict__ prev = reinterpret_cast<const PIX *>(a.prev) + (size_t)b * plane;
    const PIX *__restrict__ curr = reinterpret_cast<const PIX *>(a.curr) + (size_t)b * plane;
    const float2 *__restrict__ fin = MODE == MODE_ITER ? a.fl[sel] + (size_t)b * plane : nullptr;
    float2 *__restrict__ fout = MODE == MODE_ITER ? a.fl[1 - sel] + (size_t)b * plane : nullptr;
    float *__restrict__ ou = a.ou + (size_t)b * plane;
    float *__restrict__ ov = a.ov + (size_t)b * plane;
    const bool planar = MODE != MODE_ITER || a.planar_out != 0;   // uniform
    const bool lane_out = lane >= HL && lane < 64 - HL;
    const bool in0 = x >= HW && x < W - HW, in1 = x + 1 >= HW && x + 1 < W - HW;   // columns with a full window

    auto row_e = [&](int r) { return (unsigned)(min(max(r, 0), Hm1) * W); };   // element offset of the (clamped) row
    // The lanes of a border strip that lie outside the frame repeat its edge column ("symm"): selects on the loaded value
    // WHERE IT IS USED (fix_*), behind an optimisation fence -- written as a conditional next to the load, the compiler
    // splits the wide load in two and branches around the second half, with a full wait after each.  `edge` is
    // wave-uniform: interior strips skip the selects.
    const bool edge = xw < 0 || xw + 128 > W;
    auto opaque = [](float &v) { asm volatile("" : "+v"(v)); };
    auto fix_pix2 = [&](float2 &r) {
        if constexpr (VEC) {
            opaque(r.x);
            opaque(r.y);
            if (edge) {
                r.x = dup_hi ? r.y : r.x;
                r.y = dup_lo ? r.x : r.y;
            }
        }
    };
    auto fix_flow2 = [&](float4 &w) {
        if constexpr (VEC) {
            opaque(w.x);
            opaque(w.y);
            opaque(w.z);
            opaque(w.w);
            if (edge) {
                w.x = dup_hi ? w.z : w.x;
                w.y = dup_hi ? w.w : w.y;
                w.z = dup_lo ? w.x : w.z;
                w.w = dup_lo ? w.y : w.w;
            }
        }
    };
    // two pixels of a frame row (VEC: raw, see fix_pix2)
    auto load_pix2 = [&](const PIX *base, unsigned rowe) -> float2 {
        if constexpr (VEC) {
            const PairF w = ld_pix_pair<PIX>(scalar_ptr(base + rowe), (unsigned)cp);
            return make_float2(w.a, w.b);
        } else {
            const PIX *rp = scalar_ptr(base + rowe);
            return make_float2(ld_pix<PIX>(rp, (unsigned)c0), ld_pix<PIX>(rp, (unsigned)c1));
        }
    };
    // {u0, v0, u1, v1} of the two pixels (VEC: raw, see fix_flow2)
    auto load_flow2 = [&](const float2 *base, unsigned rowe) -> float4 {
        if constexpr (VEC) {
            return ld_off<float4>(scalar_ptr(base + rowe), (unsigned)cp * 8u);
        } else {
            const float2 *rp = scalar_ptr(base + rowe);
            const float2 f0 = ld_off<float2>(rp, (unsigned)c0 * 8u), f1 = ld_off<float2>(rp, (unsigned)c1 * 8u);
            return make_float4(f0.x, f0.y, f1.x, f1.y);
        }
    };

    const LeanGeom lg = lean_geom(H, W);
    const double gxd0 = (double)c0, gxd1 = (double)c1;
    // ITER: flow += d (lucas_kanade_pyramidal.py:209-210) needs the flow of an output row again, four rows after its warp
    // coordinates were formed from it.  Re-reading it from memory cost 8 B/px of fabric traffic (measured: the launch
    // fetched 1.58x its algorithmic reads); a wave keeps its last eight flow rows in LDS instead -- 16 bytes per lane and
    // row, written when the row's coordinates are formed, read back by the same lane: no barrier, no bank conflict.
    // UPS: the x side of the upsampling is fixed per lane -- coarse cell and fraction of the lane's two columns (the cells
    // are the same or adjacent: a fine pixel is less than a coarse one wide) -- the y side is uniform per row
    const float2 *__restrict__ csrc = nullptr;
    int xb = 0, xd = 0;            // coarse column of the low fine column; 0 / 1: the high column's cell is the next one
    double urx0 = 0.0, urx1 = 0.0;
    if constexpr (UPS) {
        const int up_sel = lk_level_state(a.acc, b, a.up_level, a.up_iters, a.L, a.K, a.up_thr).executed & 1;
        csrc = a.up_src + (size_t)up_sel * a.up_slot_stride + (size_t)b * ((size_t)a.Hc * (size_t)a.Wc);
        const double Wc2 = (double)max(a.Wc - 2, 0);
        const double x0f = linspace_at(a.up_lx, c0), x1f = linspace_at(a.up_lx, c1);
        const double f0 = fmin(floor(x0f), Wc2), f1 = fmin(floor(x1f), Wc2);
        urx0 = x0f - f0;
        urx1 = x1f - f1;
        xb = (int)f0;
        xd = (int)f1 - xb;
    }
    constexpr int FRING = 8;
    __shared__ float4 s_flow[MODE == MODE_ITER ? WPB * FRING * 64 : 1];
    float4 *const ring = s_flow + (MODE == MODE_ITER ? (threadIdx.x >> 6) * (FRING * 64) + lane : 0);

    // ---- pipeline state ------------------------------------------------------------------------------------
    // coalesced loads run LD rows ahead of the arithmetic (a multiple of 3, the period of the other rings: the row loop is
    // unrolled LD times so that every ring slot is a compile-time index).  Bytes in flight are what a streaming kernel lives
    // on: a wave holds LD rows of 16 - 28 bytes per lane
    constexpr int LD = MODE == MODE_SINGLE ? OFLK_LKS_LD_SINGLE : OFLK_LKS_LD_ITER;
    static_assert(LD % 3 == 0, "the load ring's period must be a multiple of the other rings' period");
    constexpr int U = LD % QR == 0 ? LD : LD * QR / 3;   // unroll: lcm of the ring periods (LD and QR are multiples of 3)
    float2 Pr[LD];                        // prev rows
    float2 Qr[LD];                        // SINGLE: curr rows
    float4 Fr[LD];                        // ITER: flow rows
    float Cr[LD];                         // ITER: prefetch of `curr` (see issue_loads)
    Flow2 Ua[UPS ? LD : 1][2];            // UPS: coarse cells xb, xb+1 of the two tap rows
    float2 Ub[UPS ? LD : 1][2];           // UPS: coarse cell xb+2
    LeanFrac gt[2];                       // ITER: the gathers in flight (row r + 1)
    PairF g0[2], g1[2];
    // frame-average rows {lo, hi, left neighbour of lo, right neighbour of hi}, ring of three
    float Alo[3], Ahi[3], AL[3], AR[3];
    float2 it1 = make_float2(0.0f, 0.0f);                        // It of the row above
    float P1[5][2], Qv[QR][5][2];                                // previous product row; ring of vertical pair sums
#pragma unroll
    for (int j = 0; j < 3; j++) Alo[j] = Ahi[j] = AL[j] = AR[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < QR; j++) {
#pragma unroll
        for (int pl = 0; pl < 5; pl++) Qv[j][pl][0] = Qv[j][pl][1] = 0.0f;
    }
#pragma unroll
    for (int pl = 0; pl < 5; pl++) P1[pl][0] = P1[pl][1] = 0.0f;
    float su = 0.0f, sv = 0.0f;           // |d| sums of the current three rows
    double dsu = 0.0, dsv = 0.0;          // ... of the segment
    unsigned inexact = 0u;                // SINGLE: a window of this lane's neighbourhood may differ from NumPy's (since the last flush)
    int hold = 0;                         // SINGLE, float32 frames: iterations a non-integral pixel keeps `inexact` set

    const int r0 = ys - R;                // first average row
    const int n_it = (ye - ys) + 2 * R;   // average rows r0 .. ye + R - 1

    float touch = 0.0f;                   // ITER: keeps the prefetch loads of `curr` alive (never part of a result)
    auto issue_loads = [&](int slot, int r) {
        const unsigned rowe = row_e(r);
        Pr[slot] = load_pix2(prev, rowe);
        if constexpr (UPS) {
            // the taps of the flow upsampling for fine row r: coarse rows y0, y0 + 1 (floor capped at Hc - 2: a sample on the last
            // row is the cell before it with fraction 1), coarse cells xb .. xb + 2 (the last one clamped into the row)
            const double yf = linspace_at(a.up_ly, min(max(r, 0), Hm1));
            const int y0c = __builtin_amdgcn_readfirstlane((int)fmin(floor(yf), (double)max(a.Hc - 2, 0)));   // uniform, but formed on the vector ALU
            const float2 *row0 = csrc + (size_t)y0c * (size_t)a.Wc, *row1 = csrc + (size_t)min(y0c + 1, a.Hc - 1) * (size_t)a.Wc;
            const unsigned ob = (unsigned)xb * 8u, oc = (unsigned)min(xb + 2, a.Wc - 1) * 8u;
            Ua[slot][0] = ld_off<Flow2>(row0, ob);
            Ub[slot][0] = ld_off<float2>(row0, oc);
            Ua[slot][1] = ld_off<Flow2>(row1, ob);
            Ub[slot][1] = ld_off<float2>(row1, oc);
            Cr[slot] = ld_pix<PIX>(scalar_ptr(curr + row_e(r + OFLK_LKS_PFROWS)), (unsigned)c0);
        } else if constexpr (MODE == MODE_ITER) {
            Fr[slot] = load_flow2(fin, rowe);
            // the warp's gathers two rows later find `curr` in the L2 instead of paying the HBM's latency inside the row loop:
            // a coalesced read of the same row brings its lines in (flows of a few pixels stay inside them)
            Cr[slot] = ld_pix<PIX>(scalar_ptr(curr + row_e(r + OFLK_LKS_PFROWS)), (unsigned)c0);
        } else {
            Qr[slot] = load_pix2(curr, rowe);
        }
    };
    // coordinates and gathers of row r (its flow is Fr[slot])
    auto issue_gathers = [&](int slot, int r) {
        if constexpr (MODE == MODE_ITER) {
            const int gy = min(max(r, 0), Hm1);
            const double yd = uint_to_f64_bits(gy);   // scalar ALU: the row is wave-uniform
            float4 f;
            if constexpr (UPS) {
                // upsample_flow at the lane's two pixels of row r: three fused lerps per plane in fp64, float32, then the
                // float32 multiply by float32(scale) of :135-136
                const double yf = linspace_at(a.up_ly, gy);
                const double ury = yf - fmin(floor(yf), (double)max(a.Hc - 2, 0));
                const Flow2 t0 = Ua[slot][0], t1 = Ua[slot][1];
                const float2 e0 = Ub[slot][0], e1 = Ub[slot][1];
                auto lerp = [&](float p00, float p01, float p10, float p11, double rx) {
                    const double A = (double)p00, B = (double)p01, C = (double)p10, D = (double)p11;
                    const double top = __builtin_fma(rx, B - A, A), bot = __builtin_fma(rx, D - C, C);
                    return (float)__builtin_fma(ury, bot - top, top);
                };
                const bool nx = xd != 0;   // the high column's cell is the next one
                f.x = lerp(t0.u0, t0.u1, t1.u0, t1.u1, urx0) * a.up_sx;
                f.y = lerp(t0.v0, t0.v1, t1.v0, t1.v1, urx0) * a.up_sy;
                f.z = lerp(nx ? t0.u1 : t0.u0, nx ? e0.x : t0.u1, nx ? t1.u1 : t1.u0, nx ? e1.x : t1.u1, urx1) * a.up_sx;
                f.w = lerp(nx ? t0.v1 : t0.v0, nx ? e0.y : t0.v1, nx ? t1.v1 : t1.v0, nx ? e1.y : t1.v1, urx1) * a.up_sy;
            } else {
                f = Fr[slot];
                fix_flow2(f);
            }
            ring[(r & (FRING - 1)) * 64] = f;   // (rows above the frame repeat row 0: harmless, they are never output rows)
            // int64 + float32 -> float64, as the reference (lucas_kanade_pyramidal.py:88-95)
            gt[0] = lean_frac_at(lg, yd + (double)f.y, gxd0 + (double)f.x);
            gt[1] = lean_frac_at(lg, yd + (double)f.w, gxd1 + (double)f.z);
            if constexpr (!(OFLK_LKS_ABL & 1)) {
                lean_load<false, PIX>(lg, curr, gt[0], g0[0], g1[0]);
                lean_load<false, PIX>(lg, curr, gt[1], g0[1], g1[1]);
            }
        }
    };

#pragma unroll
    for (int k = 0; k < LD; k++) issue_loads(k, r0 + k);
    issue_gathers(0, r0);

    for (int i0 = 0; i0 < n_it; i0 += U) {
        static_for(std::make_integer_sequence<int, U>{}, [&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            constexpr int jl = jj % LD;                       // = i mod LD: slot of the load rings
            constexpr int j = jj % 3;                         // = i mod 3: slot of the frame-average ring
            constexpr int jq = jj % QR;                       // = i mod QR: slot of the pair-sum ring
            const int i = i0 + jj;   // (the last trip may run up to U - 1 rows past the segment: clamped loads, no stores)
            const int r = r0 + i;
            // ---- second frame of row r (warped if ITER), frame average, It --------------------------------
            float2 p = Pr[jl];
            fix_pix2(p);
            if constexpr (MODE == MODE_ITER) touch = fmaxf(touch, Cr[jl]);
            float2 q;
            if constexpr (MODE == MODE_ITER) {
                if constexpr ((OFLK_LKS_ABL & 1) != 0) {
                    q.x = Cr[jl] + (float)gt[0].rx; q.y = Cr[jl] + (float)gt[1].ry;
                } else if constexpr ((OFLK_LKS_ABL & 32) != 0) {
                    q.x = g0[0].a + g1[0].b + (float)gt[0].rx; q.y = g0[1].a + g1[1].b + (float)gt[1].ry;
                } else if constexpr (WARPV == WARP_LERP64) {
                    q.x = lerp64_finish(gt[0], g0[0], g1[0]);
                    q.y = lerp64_finish(gt[1], g0[1], g1[1]);
                } else {
                    q.x = lean_finish(gt[0], g0[0], g1[0]);
                    q.y = lean_finish(gt[1], g0[1], g1[1]);
                }
            } else {
                q = Qr[jl];
                fix_pix2(q);
                if constexpr (sizeof(PIX) == 4) {
                    // a float32 frame is only promised to be what the verifier makes of 8-bit files (optical_flow_verifier.py:61-65);
                    // a pixel that is not an integer in [0, 255] voids the exactness argument for the windows it touches: the
                    // outputs completed by this and the next 2R iterations, R columns either side
                    auto integral = [](float t) { return t == __builtin_truncf(t) && fabsf(t - 127.5f) <= 127.5f; };
                    if (!(integral(p.x) && integral(p.y) && integral(q.x) && integral(q.y))) hold = 2 * R + 1;
                }
            }
            if constexpr (MODE == MODE_SINGLE) {
                if (hold > 0) {
                    inexact |= 1u;
                    hold--;
                }
            }
            const int o = r - R;                               // the output row this iteration completes
            const bool o_live = o >= ys && o < ye;             // uniform: the pipeline is full and the row is the segment's
            __builtin_amdgcn_sched_barrier(0);                 // the waits above come before the issues below
            // Vector-memory results return in issue order, so a wait for one load is a wait for every load issued before
            // it: the loads of an iteration go out oldest-needed first -- the gathers of the next row, then the coalesced
            // loads three rows ahead -- and every wait leaves the younger ones in flight.
            float4 pf = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if constexpr (MODE == MODE_ITER) {
                if constexpr (!(OFLK_LKS_ABL & 2))
                    if (o_live) pf = ring[(o & (FRING - 1)) * 64];   // written four iterations ago by this lane
                issue_gathers((jl + 1) % LD, r + 1);          // used by the next row, after this row's arithmetic
            }
            issue_loads(jl, r + LD);
            __builtin_amdgcn_sched_barrier(0);
            // (prev + curr) / 2.0 and prev - curr, lucas_kanade_core.py:36, :43
            const float s0 = p.x + q.x, s1 = p.y + q.y;
            Alo[j] = s0 * 0.5f;
            Ahi[j] = s1 * 0.5f;
            const float2 itn = make_float2(p.x - q.x, p.y - q.y);
            AL[j] = wshift<SHR>(Ahi[j]);                       // column 2l - 1
            AR[j] = wshift<SHL>(Alo[j]);                       // column 2l + 2
            // ---- Sobel/8 of row r - 1 in convolve2d's tap order (same operations as k_lkw's) ----------------
            constexpr int jm = (j + 1) % 3, jz = (j + 2) % 3, jp = j;   // rows r-2, r-1, r
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
            float ix[2], iy[2];
            sobel(AL[jm], Alo[jm], Ahi[jm], AL[jz], Ahi[jz], AL[jp], Alo[jp], Ahi[jp], ix[0], iy[0]);
            sobel(Alo[jm], Ahi[jm], AR[jm], Alo[jz], AR[jz], Alo[jp], Ahi[jp], AR[jp], ix[1], iy[1]);
            const float itv[2] = {it1.x, it1.y};
            it1 = itn;
            // ---- products of row g = r - 1, vertical sums of output row o = g - 2 -----------------------------
            float V[5][2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                float pr[5];
                pr[0] = ix[c] * ix[c];
                pr[1] = iy[c] * iy[c];
                pr[2] = ix[c] * iy[c];
                pr[3] = ix[c] * itv[c];
                pr[4] = iy[c] * itv[c];
#pragma unroll
                for (int pl = 0; pl < 5; pl++) {
                    // ring slot of the pair sum q[g'] = p[g'] + p[g'+1]: the iteration that formed it, mod 3
                    if constexpr (HW == 2) {
                        V[pl][c] = (Qv[(jq + 2) % 3][pl][c] + Qv[(jq + 1) % 3][pl][c]) + pr[pl];   // (q[g-4] + q[g-2]) + p[g]
                        Qv[(jq + 2) % 3][pl][c] = P1[pl][c] + pr[pl];                              // q[g-1]
                    } else {
                        V[pl][c] = ((Qv[jq][pl][c] + Qv[(jq + 2) % 6][pl][c]) + Qv[(jq + 4) % 6][pl][c]) + pr[pl];   // ((q[g-6] + q[g-4]) + q[g-2]) + p[g]
                        Qv[(jq + 5) % 6][pl][c] = P1[pl][c] + pr[pl];                                                // q[g-1]
                    }
                    P1[pl][c] = pr[pl];
                }
            }
            // ---- horizontal sums, solve, flow += d ------------------------------------------------------------
            float S[5][2];
#pragma unroll
            for (int pl = 0; pl < 5; pl++) {
                const float lo = V[pl][0], hi = V[pl][1];
                const float pp = lo + hi;
                if constexpr (HW == 2) {
                    const float X = wshift<SHR>(pp) + pp;          // (V[c-2] + V[c-1]) + (V[c] + V[c+1])
                    S[pl][0] = X + wshift<SHL>(lo);                // ... + V[c+2]
                    const float Y = wshift<SHR>(hi) + pp;          // V[c-2] + (V[c-1] + V[c])      (c = the odd column)
                    S[pl][1] = Y + wshift<SHL>(pp);                // ... + (V[c+1] + V[c+2])
                } else {
                    // seven columns: the three lane pairs around the lane plus one column two lanes away
                    const float X = (wshift<SHR>(pp) + pp) + wshift<SHL>(pp);
                    S[pl][0] = X + wshift<SHR>(wshift<SHR>(hi));   // even column c: ... + V[c-3]
                    S[pl][1] = X + wshift<SHL>(wshift<SHL>(lo));   // odd column c:  ... + V[c+3]
                }
                if constexpr ((OFLK_LKS_ABL & 16) != 0) { S[pl][0] = lo; S[pl][1] = hi; }
            }
            const bool oky = o >= HW && o < H - HW;            // borders stay 0 (lucas_kanade_core.py:101-108)
            float du[2], dv[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                float uu, vv;
                if constexpr ((OFLK_LKS_ABL & 8) != 0) { uu = S[0][c] + S[1][c] + S[2][c]; vv = S[3][c] + S[4][c]; }
                else if constexpr (MODE == MODE_SINGLE) {
                    // Where the result is kept (integer frames in [0, 255], Sxx, Syy < 2^16 -- everything else is redone by the
                    // tile kernel) the quotients' operands are in div2_shared_rcp's range: 1e-4 < |det| <= Sxx Syy < 2^32;
                    // a numerator is the rounded difference of two exact multiples of 2^-12 (sums are multiples of 2^-8 and
                    // 2^-4) below 2^16 * 2^20, hence 0 or at least 2^-12 and below 2^37.  In a tile that is redone anyway the
                    // values written here do not matter.
                    lk_solve<true>(S[0][c], S[1][c], S[2][c], S[3][c], S[4][c], uu, vv);
                } else lk_solve(S[0][c], S[1][c], S[2][c], S[3][c], S[4][c], uu, vv);
                const bool interior = oky && (c ? in1 : in0);
                du[c] = interior ? uu : 0.0f;
                dv[c] = interior ? vv : 0.0f;
                if constexpr (MODE == MODE_SINGLE) {
                    if (interior && o_live && !(S[0][c] < kLksExactBound && S[1][c] < kLksExactBound)) inexact |= 1u;
                }
                if constexpr (MODE == MODE_ITER) {
                    if (lane_out && o_live) {   // halo lanes repeat their neighbours' pixels
                        su += fabsf(du[c]);
                        sv += fabsf(dv[c]);
                    }
                }
            }
            if (lane_out && o_live && (!(OFLK_LKS_ABL & 4) || du[0] == 1.25e37f)) {
                const unsigned orow = (unsigned)(o * W);
                float2 ru = make_float2(du[0], du[1]), rv = make_float2(dv[0], dv[1]);
                if constexpr (MODE == MODE_ITER) {
                    ru.x = pf.x + ru.x; ru.y = pf.z + ru.y;
                    rv.x = pf.y + rv.x; rv.y = pf.w + rv.y;
                }
                if constexpr (VEC) {
                    if (x < W) {   // x >= 0 for output lanes; W even: the pair is inside
                        if (planar) {
                            st_off<float2>(ou + orow, 4u * (unsigned)x, ru);
                            st_off<float2>(ov + orow, 4u * (unsigned)x, rv);
                        } else {
                            st_off<float4>(fout + orow, 8u * (unsigned)x, make_float4(ru.x, rv.x, ru.y, rv.y));
                        }
                    }
                } else {
                    if (x < W) {
                        if (planar) {
                            st_off<float>(ou + orow, 4u * (unsigned)x, ru.x);
                            st_off<float>(ov + orow, 4u * (unsigned)x, rv.x);
                        } else {
                            st_off<float2>(fout + orow, 8u * (unsigned)x, make_float2(ru.x, rv.x));
                        }
                    }
                    if (x + 1 < W) {
                        if (planar) {
                            st_off<float>(ou + orow, 4u * (unsigned)(x + 1), ru.y);
                            st_off<float>(ov + orow, 4u * (unsigned)(x + 1), rv.y);
                        } else {
                            st_off<float2>(fout + orow, 8u * (unsigned)(x + 1), make_float2(ru.y, rv.y));
                        }
                    }
                }
            }
            if constexpr (MODE == MODE_SINGLE) {
                // at the last output row of a tile row and at the end of the segment the doubtful windows of the last rows go
                // to the redo list (LkArgs::redo), tile by tile: the wave's columns reach at most four tile columns; a lane
                // vouches for the columns R either side of its pair (a superset of the windows its pixels are in)
                if (a.redo != nullptr && o_live && (((o + 1) % k5TY) == 0 || o == ye - 1)) {
                    const int tiles_x = (W + k5TX - 1) / k5TX, tiles_y = (H + k5TY - 1) / k5TY;
                    const int tlo = min(max(x - R, 0), Wm1) / k5TX, thi = min(max(x + 1 + R, 0), Wm1) / k5TX;
                    const int t_first = min(max(xw - R, 0), Wm1) / k5TX, t_last = min(max(xw + 128 + R, 0), Wm1) / k5TX;
                    const unsigned T = (unsigned)(a.B * tiles_y * tiles_x);
                    for (int t = t_first; t <= t_last; t++) {   // uniform bounds
                        if (__ballot(inexact != 0u && (tlo == t || thi == t)) != 0ull && lane == 0) {
                            const unsigned tile = (unsigned)((b * tiles_y + o / k5TY) * tiles_x + t);
                            if (__hip_atomic_exchange(&a.redo[2u + tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                const unsigned e = __hip_atomic_fetch_add(&a.redo[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                a.redo[2u + T + e] = tile;
                            }
                        }
                    }
                    inexact = 0u;
                }
            }
        });
        if constexpr (MODE == MODE_ITER) {
            // 2 LD fp32 terms per lane, then fp64 (k_lkw: six per thread and tile)
            dsu += (double)su;
            dsv += (double)sv;
            su = 0.0f;
            sv = 0.0f;
        }
    }
    if constexpr (MODE == MODE_ITER) {
        // the wave's totals in the accumulators' fixed point: integer adds commute, so neither the lane order here nor the
        // order of the waves' atomics matters
        const double cap = kAccBlockMax / 64.0;
        dsu = dsu < cap ? dsu : cap;   // (also catches NaN)
        dsv = dsv < cap ? dsv : cap;
        long long tu = __double2ll_rn(dsu * kAccScale), tv = __double2ll_rn(dsv * kAccScale);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            tu += __shfl_xor(tu, m, 64);
            tv += __shfl_xor(tv, m, 64);
        }
        if (touch == 1.5e38f) tu += 1;   // never true for pixel data; makes `touch` observable
        if (lane == 0) {
            unsigned long long *slot = a.acc + acc_index(b, a.level, a.iter, a.L, a.K) + kAccStride * (task & (kAccShards - 1));
            __hip_atomic_fetch_add(slot + 0, (unsigned long long)tu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(slot + 1, (unsigned long long)tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace oflk
