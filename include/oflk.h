/*
 * oflk.h -- C ABI of liboflk.so: dense Lucas-Kanade optical flow on MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of rothej/optical-flow-fpga's
 * Python golden model.  The reference has no FFI of its own: its "interface" is
 * a set of plain Python functions in python/lucas_kanade_core.py and
 * python/lucas_kanade_pyramidal.py, imported by name by the verifier
 * (python/optical_flow_verifier.py:19-20) and the demo CLIs.  Each entry point
 * below cites the reference function it replaces; the Python shims in
 * optical-flow-fpga_amd/python/ (same module and function names as the
 * reference) bind them with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.
 *   - Images and flow fields are C-contiguous row-major float32 [H][W]
 *     (batched: [B][H][W]).
 *   - Every function returns OFLK_OK (0) or a negative OFLK_ERR_* code;
 *     oflk_last_error() returns a human-readable message for the calling thread.
 *   - There is NO CPU fallback: every compute entry point runs hand-written HIP
 *     kernels and fails with OFLK_ERR_NO_DEVICE when no gfx950 GPU is usable.
 *   - "host" entry points take host pointers and are synchronous (H2D, kernels,
 *     D2H inside the call).  "plan" entry points take device pointers, enqueue
 *     on a caller-supplied hipStream_t and return without synchronising.
 */
#ifndef OFLK_H
#define OFLK_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFLK_OK 0
#define OFLK_ERR_INVALID (-1)     /* bad argument (null pointer, non-positive size, ...) */
#define OFLK_ERR_NO_DEVICE (-2)   /* no usable HIP device / HIP runtime error at init */
#define OFLK_ERR_HIP (-3)         /* a HIP runtime call failed (message has details) */
#define OFLK_ERR_UNSUPPORTED (-4) /* parameter outside what the kernels are built for */
#define OFLK_ERR_NOMEM (-5)       /* device or host allocation failed */

#define OFLK_MAX_LEVELS 16
#define OFLK_MAX_WINDOW 45 /* largest window_size (45 x 45 = 2025 products; even sizes round down like the reference) */
#define OFLK_MAX_TILED_WINDOW 11 /* 3x3 ... 11x11 run the tiled kernels; every other size the generic one-thread-per-pixel kernel */

/* ---- library ------------------------------------------------------------ */
const char *oflk_version(void);
/* number of visible HIP devices; 0 when there is none (never fails) */
int oflk_device_count(void);
/* message of the last error raised on the calling thread ("" if none) */
const char *oflk_last_error(void);
/* device used by the host entry points (default 0).  Each device has its own plan cache and
 * staging buffers; calls on different devices (from different host threads) run concurrently. */
int oflk_set_device(int device);

/* ---- host-pointer entry points: one per reference function ---------------- */

/* compute_gradients(frame_prev, frame_curr) -> (Ix, Iy, It)
 * replaces python/lucas_kanade_core.py:15-45 */
int oflk_compute_gradients(const float *prev, const float *curr, int H, int W, float *Ix,
                           float *Iy, float *It);

/* lucas_kanade_from_gradients(Ix, Iy, It, window_size) -> (u, v)
 * replaces python/lucas_kanade_core.py:73-135 */
int oflk_from_gradients(const float *Ix, const float *Iy, const float *It, int H, int W,
                        int window_size, float *u, float *v);

/* lucas_kanade_single_scale(frame_prev, frame_curr, window_size) -> (u, v)
 * replaces python/lucas_kanade_core.py:48-70 (one fused kernel) */
int oflk_single_scale(const float *prev, const float *curr, int H, int W, int window_size,
                      float *u, float *v);

/* level sizes of build_gaussian_pyramid: dims_out[2*l] = H_l, dims_out[2*l+1] = W_l,
 * l = 0 is the coarsest level (python/lucas_kanade_pyramidal.py:51-52, :61) */
int oflk_pyramid_level_dims(int H, int W, int levels, double scale_factor, int *dims_out);

/* build_gaussian_pyramid(image, num_levels, scale_factor) -> [coarse .. fine]
 * replaces python/lucas_kanade_pyramidal.py:23-63; out_levels[l] must hold H_l*W_l floats */
int oflk_build_pyramid(const float *image, int H, int W, int levels, double scale_factor,
                       float *const *out_levels);

/* The same with the Gaussian weights given by the caller: weights[k], k = 0 .. radius, is the normalised weight at distance k
 * of scipy.ndimage.gaussian_filter's kernel for sigma = 1 / scale_factor (radius = int(4 sigma + 0.5)).  SciPy forms them with
 * NumPy's exp, whose last bit differs from libm's for some arguments; oflk_build_pyramid embeds SciPy's table for the
 * reference's default scale_factor 0.5 and falls back to libm elsewhere.  The Python shim computes the weights with NumPy the
 * way SciPy does (lucas_kanade_pyramidal.build_gaussian_pyramid) and calls this entry point, so EVERY scale factor gives the
 * reference's pyramid (tests/golden/pyramid_scales*.npz, made by importing the reference). */
int oflk_build_pyramid_w(const float *image, int H, int W, int levels, double scale_factor, const double *weights,
                         int radius, float *const *out_levels);

/* warp_image(image, flow_u, flow_v) -> warped
 * replaces python/lucas_kanade_pyramidal.py:66-97 */
int oflk_warp(const float *image, const float *flow_u, const float *flow_v, int H, int W,
              float *out);

/* upsample_flow(flow_u, flow_v, (Ht, Wt)) -> (u, v)
 * replaces python/lucas_kanade_pyramidal.py:100-138 */
int oflk_upsample_flow(const float *flow_u, const float *flow_v, int Hc, int Wc, int Ht, int Wt,
                       float *u_out, float *v_out);

/* lucas_kanade_pyramidal(frame_prev, frame_curr, num_levels, window_size, num_iterations) -> (u, v)
 * replaces python/lucas_kanade_pyramidal.py:141-228.
 *   residual_log : [levels][iters][2] floats, mean|du|, mean|dv| of each executed
 *                  iteration (what the reference prints at :215-218); may be NULL
 *   iters_run    : [levels] ints, iterations executed per level (early exit at
 *                  :221-223); may be NULL */
int oflk_pyramidal(const float *prev, const float *curr, int H, int W, int levels,
                   int window_size, int iters, float *u, float *v, float *residual_log,
                   int *iters_run);

/* B independent frame pairs in one call; arrays are [B][H][W], residual_log is
 * [B][levels][iters][2], iters_run is [B][levels] (both may be NULL). */
int oflk_single_scale_batch(const float *prev, const float *curr, int B, int H, int W,
                            int window_size, float *u, float *v);
int oflk_pyramidal_batch(const float *prev, const float *curr, int B, int H, int W, int levels,
                         int window_size, int iters, float *u, float *v, float *residual_log,
                         int *iters_run);

/* The two reads above for the host entry points: they address the plan the last oflk_pyramidal* call
 * of this shape left in the current device's cache (OFLK_ERR_INVALID if there is none). */
int oflk_pyramidal_last_level_flow(int B, int H, int W, int levels, int window_size, int iters, int level,
                                   int pair, float *u, float *v);
int oflk_pyramidal_last_uncertain(int B, int H, int W, int levels, int window_size, int iters, int *uncertain);

/* ---- uint8 frames (the reference's on-disk format) ------------------------- */
/* Same as the batch entry points, for raw 8-bit frames [B][H][W] as generate_test_suite.py
 * writes them (frame_0x.bin, :259-261).  The kernels read the bytes themselves (2 B/px of frame
 * traffic instead of 8; no float32 copy of the frames exists on the device): the uint8 -> float32
 * conversion the verifier does on the host (python/optical_flow_verifier.py:61-65) happens in
 * registers and is exact, so results are identical to converting first. */
int oflk_single_scale_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                         int window_size, float *u, float *v);
int oflk_pyramidal_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                      int levels, int window_size, int iters, float *u, float *v, float *residual_log,
                      int *iters_run);
/* device-side conversion for pipelines that hold uint8 frames in HBM: d_out[i] = (float)d_in[i] */
int oflk_u8_to_f32(const unsigned char *d_in, float *d_out, size_t n, void *stream);

/* ---- reduced precision (BASELINE config 5), opt-in, never the default ------------ */
/* lucas_kanade_single_scale with fp16 gradients and fp16 window accumulators: float32 frames in,
 * float32 flow out, everything between the frame average and the 2x2 solve in half precision, window
 * sums separable.  This is NOT the reference's arithmetic (python/lucas_kanade_core.py:110-133 is fp32
 * throughout): results are close to, not equal to, oflk_single_scale_batch -- how close is measured,
 * per pattern, by tests/test_gpu_fp16.py (mean / median endpoint error against the exact path).
 *   pixel_max : upper bound of the frame values (255 for the reference's 8-bit frames); frames are
 *               scaled by a power of two so that a window sum of Ix^2 stays below fp16's 65504.
 *               Values beyond [0, pixel_max] may overflow to inf / nan. */
int oflk_single_scale_fp16(const float *prev, const float *curr, int B, int H, int W, int window_size,
                           float pixel_max, float *u, float *v);

/* ---- one process, several GPUs ------------------------------------------------- */
/* Frame pairs are independent units (python/lucas_kanade_pyramidal.py:141-228 touches only its two
 * inputs), so a batch spreads over GPUs with no data-path exchange.  How long a pair takes depends on its
 * data (the early exit of :221-223), so devices do not get fixed shards: the B pairs are cut into chunks of
 * consecutive pairs (about four per device) and each device -- a host thread of its own, its own plans --
 * pulls the next chunk from a shared counter until none is left; every chunk is written to its slice of
 * the host output arrays.  n_gpus <= 0 means all visible devices; n_gpus > visible devices is
 * OFLK_ERR_INVALID; with n_gpus == 1 the call is oflk_*_batch on the device of oflk_set_device.  Results
 * do not depend on n_gpus, nor on which device took which chunk.
 * (The multi-process form -- one rank per GPU, torch.distributed over RCCL -- is bench.py's.) */
int oflk_single_scale_batch_multi(const float *prev, const float *curr, int B, int H, int W,
                                  int window_size, int n_gpus, float *u, float *v);
int oflk_pyramidal_batch_multi(const float *prev, const float *curr, int B, int H, int W, int levels,
                               int window_size, int iters, int n_gpus, float *u, float *v,
                               float *residual_log, int *iters_run);
int oflk_pyramidal_u8_multi(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                            int levels, int window_size, int iters, int n_gpus, float *u, float *v,
                            float *residual_log, int *iters_run);
/* Rehearsal of the chunk queue above on a box with fewer GPUs than workers (tests): `workers` > 0 makes the *_multi entry
 * points run that many queue workers, worker i on device i % n_gpus (workers of one device take turns on it); 0 restores
 * one worker per device.  Results do not change. */
int oflk_multi_rehearsal(int workers);
/* [begin, end) of `total` units owned by shard `shard` of `n_shards` (contiguous, sizes differ by at
 * most one): the partition used above and by bench.py's ranks */
void oflk_shard_range(int total, int shard, int n_shards, int *begin, int *end);

/* ---- device-resident plan API (pipelines, bench) -------------------------- */
typedef struct oflk_plan oflk_plan;

/* Allocate the workspace (pyramids, flow ping-pong buffers, reduction scratch)
 * for B pairs of H x W on `device`.  levels = 1 and iters = 0 gives a plan that
 * can only run oflk_plan_single_scale.
 *   window_size : 1 ... 45.  Like the reference (lucas_kanade_core.py:104, :110) a size w uses the (2*(w/2)+1)^2
 *                 window, so 4 and 5 both mean 5x5.  3x3 ... 11x11 (sizes 2 ... 11) run the tiled kernels.  Every other
 *                 size -- 1x1, 13x13 ... 45x45, where np.sum's pairwise order splits into blocks -- runs a generic
 *                 kernel (one thread per output pixel, the sums in NumPy's order for any length; a pyramidal pass
 *                 then runs pair by pair, unfused, with the exit test on the host): the reference's values, slowly.
 *                 Sizes above 45 return OFLK_ERR_UNSUPPORTED; the fp16 mode exists for 2 ... 11 only.
 *   A plan is single-stream: it owns one per-call state block, so at most ONE pass of a plan may be
 *   in flight at a time (enqueue passes of one plan on one stream, or synchronise between streams).
 *   Device pointers: when W % 4 == 0 the kernels move 16 bytes per lane and want every plane
 *   (d_prev, d_curr, d_u, d_v; uint8 frames: 4-byte) 16-byte aligned -- what hipMalloc and
 *   torch allocations give.  Other alignments are accepted and take the element-wise kernels. */
int oflk_plan_create(oflk_plan **plan, int device, int B, int H, int W, int levels,
                     int window_size, int iters);
int oflk_plan_destroy(oflk_plan *plan);
size_t oflk_plan_workspace_bytes(const oflk_plan *plan);

/* Enqueue one pass over the batch on `stream` (a hipStream_t; NULL = default
 * stream).  d_* are device pointers, [B][H][W] float32.  Asynchronous. */
int oflk_plan_single_scale(oflk_plan *plan, const float *d_prev, const float *d_curr,
                           float *d_u, float *d_v, void *stream);
int oflk_plan_pyramidal(oflk_plan *plan, const float *d_prev, const float *d_curr, float *d_u,
                        float *d_v, void *stream);
/* device-resident form of oflk_single_scale_fp16 (any plan; levels / iters are not used) */
int oflk_plan_single_scale_fp16(oflk_plan *plan, const float *d_prev, const float *d_curr, float *d_u, float *d_v,
                                float pixel_max, void *stream);
/* the same passes on device-resident uint8 frames [B][H][W] (see "uint8 frames" above) */
int oflk_plan_single_scale_u8(oflk_plan *plan, const unsigned char *d_prev, const unsigned char *d_curr,
                              float *d_u, float *d_v, void *stream);
int oflk_plan_pyramidal_u8(oflk_plan *plan, const unsigned char *d_prev, const unsigned char *d_curr,
                           float *d_u, float *d_v, void *stream);

/* After oflk_plan_pyramidal: copy the residual log / iteration counts of the last
 * enqueued pass to the host (synchronises `stream`).  Either pointer may be NULL. */
int oflk_plan_read_log(oflk_plan *plan, float *residual_log, int *iters_run, void *stream);

/* After oflk_plan_pyramidal + read_log: uncertain[b*levels + l] has bit k set when the early-exit test
 * after iteration k of level l (python/lucas_kanade_pyramidal.py:221-223) was decided with a mean
 * within the level's band around the 0.01 threshold: max(5e-5, (ceil(npix / 8192) + 32) * 2^-24) relative, i.e.
 * 5e-5 up to 6.9 Mpx levels, 6.2e-5 at 4K, 2.4e-4 at 8K.  The reference sums np.mean in fp32 (pairwise, in
 * 8192-element pieces), the device in exact fixed point; the two can only decide differently inside
 * that band, so 0 everywhere means "provably the reference's iteration counts".  Synchronises. */
int oflk_plan_read_uncertain(oflk_plan *plan, int *uncertain, void *stream);

/* Closes that band: every pair of the last pass with a flagged decision is redone -- that pair alone, with
 * the standalone kernels in the reference's own sequence (pyramid, warp, single-scale LK, flow += d,
 * upsample; value-identical to the fused path) and np.mean(np.abs(d)) evaluated in NumPy's own order
 * (fp32 pairwise inside 8192-element pieces, the pieces added serially), the exit test taken on the host.
 * The pair's slices of d_u / d_v, its log, iteration counts and flags are replaced, so afterwards the
 * whole batch is the reference's result with the reference's iteration counts.  Slow (a host round trip
 * per iteration) and rare (never seen outside constructed inputs).  d_prev / d_curr are the frames the
 * pass was given; *resolved (may be NULL) receives the number of pairs redone.  Synchronises.
 * The host entry points (oflk_pyramidal*, oflk_pyramidal_u8*) do this themselves after every call;
 * oflk_last_resolved() tells how many pairs the calling thread's last such call redid. */
int oflk_plan_resolve_uncertain(oflk_plan *plan, const float *d_prev, const float *d_curr, float *d_u, float *d_v,
                                void *stream, int *resolved);
int oflk_plan_resolve_uncertain_u8(oflk_plan *plan, const unsigned char *d_prev, const unsigned char *d_curr,
                                   float *d_u, float *d_v, void *stream, int *resolved);
int oflk_last_resolved(void);

/* Final flow of a coarser pyramid level (level < levels-1; the finest level's flow is the result) of
 * pair `pair` of the last pass, to host arrays of that level's size -- what the reference hands to
 * visualize_pyramid_level at python/lucas_kanade_pyramidal.py:226.  Synchronises. */
int oflk_plan_read_level_flow(oflk_plan *plan, int level, int pair, float *u, float *v, void *stream);

/* Arithmetic of the plan's fp64 stages.  OFLK_ARITH_EXACT (the default of plans and of the host entry points): SciPy's
 * operation sequence, every operation rounded on its own -- results equal the reference's value for value.
 * OFLK_ARITH_CONTRACTED (opt-in): the Gaussian pyramid (python/lucas_kanade_pyramidal.py:46-59) accumulates with fused
 * multiply-adds, 17 instead of 25 fp64 operations per blurred value on a kernel the fp64 pipe binds.  Intermediates
 * differ from SciPy's by a few 1e-16 relative before they are rounded to float32 where SciPy rounds, so a pyramid value
 * differs from the reference's only where the fp64 value lies that close to a float32 rounding boundary (about one in
 * 10^7, by one ulp); coarse-to-fine LK then amplifies such a difference locally.  Measured on the 13 verification
 * patterns (tests/test_gpu_round3.py, profiles/): mean EPE against the reference far below the 1e-4 bar.  Affects
 * oflk_plan_pyramidal / _u8 only.
 * OFLK_ARITH_TOLERANT (opt-in): everything whose cost in endpoint error against the reference was measured, cell by cell
 * (stage x pyramid level x iteration: tools/experiments/fast_mode_ablation.py, profiles/), to sit at least three times
 * under the north star's bar of 1e-4 px mean EPE on its own and under 5e-5 combined:
 *   - the contracted pyramid (above);
 *   - on the TWO FINEST levels, 5x5 window, the fused iteration runs as a streaming kernel (k_lks) whose window sums are
 *     separable -- five rows added vertically, then five columns horizontally, not np.sum's pairwise order of
 *     python/lucas_kanade_core.py:115-119 -- and whose warp (python/lucas_kanade_pyramidal.py:88-96) forms the bilinear
 *     sample as three fused lerps in fp64 instead of SciPy's 15 operations; gradients, products, the 2x2 solve (IEEE
 *     divisions) and flow += d are the reference's operations;
 *   - coarser levels, the flow upsampling, other windows: exact, as in OFLK_ARITH_EXACT.
 * Measured against dense flows of the reference itself (tests/golden/dense_reference_flows.npz): worst of the 13
 * verification patterns 1.7e-5 px (translate_extreme), the 1080p bench pair 5.6e-7 px.  The arithmetic is stated on the
 * CPU by oracle/oflk_tolerant_model.c (test infrastructure) and the kernels are held to that statement bit for bit
 * (tests/test_gpu_round4.py), so the tolerance is a property of one written-down arithmetic, not of a GPU run.
 * In both opt-in modes the exit-decision flags keep their meaning, and oflk_plan_resolve_uncertain redoes a flagged pair in
 * EXACT arithmetic from the caller's frames (its own exact pyramid): a redone pair is the reference's result, which is
 * inside any tolerance.  Windows without a fused iteration kernel (1x1, 13x13 ...) always run exactly. */
#define OFLK_ARITH_EXACT 0
#define OFLK_ARITH_CONTRACTED 1
#define OFLK_ARITH_TOLERANT 2
int oflk_plan_set_arithmetic(oflk_plan *plan, int mode);
/* The same choice for the host-pointer entry points (oflk_pyramidal*, oflk_pyramidal_u8*, the *_multi forms): process-wide,
 * default OFLK_ARITH_EXACT.  The Python shims call it once when the environment variable OFLK_ARITH is set to "contracted" or
 * "tolerant"; without it the drop-in functions return the reference's values.  Pairs whose exit decision is flagged are
 * redone exactly in every mode (oflk_last_resolved). */
int oflk_set_host_arithmetic(int mode);

/* Which kernel runs a single-scale pass (oflk_plan_single_scale / _u8; results are the reference's either way).
 * The 5x5 and 7x7 windows have a streaming kernel (no LDS; window sums vertical-then-horizontal), which equals np.sum's order exactly
 * wherever the frames are integers in [0, 255] -- what python/optical_flow_verifier.py:61-65 makes of the 8-bit files -- and a
 * window's sum Ix^2, sum Iy^2 stay below 2^16 (every partial sum is then exact in any order; proof in csrc/oflk_stream.hpp);
 * tiles where either is in doubt are flagged on the device and redone by the tile kernel in NumPy's order within the same call.
 * OFLK_KERNELS_AUTO (default): the streaming kernel for launches large enough to fill the chip with it (it walks rows serially
 * inside a wave: a single small pair is faster on the tile kernel), the tile kernel otherwise.  OFLK_KERNELS_TILE: the tile
 * kernel for everything (NumPy's order throughout).  OFLK_KERNELS_STREAM: the streaming kernel whenever the window is 5x5 or 7x7. */
#define OFLK_KERNELS_AUTO 0
#define OFLK_KERNELS_TILE 1
#define OFLK_KERNELS_STREAM 2
int oflk_plan_set_kernels(oflk_plan *plan, int choice);

/* Per-kernel timing with HIP events on the launch stream.  While enabled, every
 * kernel launch of the plan is bracketed by an event pair; oflk_plan_kernel_times
 * synchronises, accumulates and reports per kernel class.
 *   names[i]  : static strings (kernel class names), up to max entries
 *   total_ms  : summed duration per class since profiling was enabled/reset
 *   launches  : launch count per class
 * Returns the number of classes written (>= 0) or an error code.
 * enabled: 0 off, 1 every kernel, 2 only the dominant kernel (fused LK iteration at the
 * finest level; three event pairs per pyramidal call instead of 15). */
int oflk_plan_set_profiling(oflk_plan *plan, int enabled);
int oflk_plan_kernel_times(oflk_plan *plan, const char **names, double *total_ms, long *launches,
                           int max_entries);

/* ---- masked flow metrics on the device -------------------------------------- */
/* compute_all_metrics(u_pred, v_pred, u_true, v_true, mask) of python/flow_metrics.py:166-201
 * (mean_absolute_error :14-40, root_mean_square_error :43-70, endpoint_error :73-103,
 * angular_error :106-163) for the rectangular test regions the verifier builds
 * (mask[y0:y1, x0:x1] = True with NumPy slice semantics, python/optical_flow_verifier.py:96-138).
 *   u_true, v_true : host arrays [B], the constant ground-truth vector of each pair
 *   out            : host array [B][5] = mae_u, mae_v, rmse, epe, aae (degrees)
 * Element-wise arithmetic is the reference's fp32; sums and arccos are fp64, so values agree
 * with the reference's fp32 pairwise means to ~1e-6 relative (not bit for bit).
 * oflk_plan_metrics reads device-resident flows [B][H][W] of the plan's shape and
 * synchronises `stream`; oflk_flow_metrics takes host arrays. */
int oflk_plan_metrics(oflk_plan *plan, const float *d_u, const float *d_v, const float *u_true,
                      const float *v_true, int y0, int y1, int x0, int x1, double *out, void *stream);
int oflk_flow_metrics(const float *u, const float *v, int B, int H, int W, const float *u_true,
                      const float *v_true, int y0, int y1, int x0, int x1, double *out);

/* ---- RTL-bit-accurate integer mode ------------------ */
/* What the reference's single-scale RTL computes for a frame pair streamed by
 * rtl/common/frame_buffer_simple.sv -- rtl/common/line_buffer_5x5.sv:75-151 (window geometry),
 * rtl/unopt/gradient_compute.sv:89-139 (averaged-frame Sobel >>> 3, It = prev - curr),
 * rtl/unopt/window_accumulator.sv:100-189 (25 products per quantity, 32-bit sums),
 * rtl/unopt/flow_solver.sv:82-149 (low-32-bit products, |det| > 1000, (num <<< 7) / det truncating,
 * clamp to +-1024) -- i.e. the values `flow_u` / `flow_v` carry in S8.7 fixed point when the
 * accumulator has ingested element k of the gradient stream.  It replaces an xsim run of
 * tb/tb_optical_flow_top.sv as the golden model of the RTL; python/rtl_golden_model.py turns the
 * per-element values into the testbench's sample sequence, summary and flow_field.txt.
 *   prev, curr : uint8 [B][H][W], 5 <= H <= 512, 5 <= W <= 1024 (flow_y is 9, flow_x 10 bits wide)
 *   u, v       : int16 [B][oflk_rtl_stream_length(H, W)] = [B][(H-4)(W-4)]
 * The RTL's geometry is kept with its quirks (one-pixel stream offset, the accumulator's rows are
 * not image rows, the window of a row's last position is mostly zero): oracle/rtl_model.py
 * states it.  PARITY UNPINNED: the model is held equal to a cycle-by-cycle execution of the modules
 * (oracle/rtl_cycle_sim.py), but no simulator output of the RTL as committed exists to pin either. */
long oflk_rtl_stream_length(int H, int W);
int oflk_rtl_flow_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W, short *u, short *v);
/* device-resident form; enqueues one kernel on `stream` */
int oflk_rtl_flow_u8_device(const unsigned char *d_prev, const unsigned char *d_curr, int B, int H, int W, short *d_u,
                            short *d_v, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OFLK_H */
