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
#define OFLK_MAX_WINDOW 11 /* largest window_size with a compiled kernel (3x3 ... 11x11; even sizes round down like the reference) */

/* ---- library ------------------------------------------------------------ */
const char *oflk_version(void);
/* number of visible HIP devices; 0 when there is none (never fails) */
int oflk_device_count(void);
/* message of the last error raised on the calling thread ("" if none) */
const char *oflk_last_error(void);
/* device used by the host entry points (default 0) */
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

/* ---- uint8 frames (the reference's on-disk format) ------------------------- */
/* Same as the batch entry points, for raw 8-bit frames [B][H][W] as generate_test_suite.py
 * writes them (frame_0x.bin, :259-261).  The uint8 -> float32 conversion the verifier does
 * on the host (python/optical_flow_verifier.py:61-65) runs on the device; results are
 * identical to converting first. */
int oflk_single_scale_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                         int window_size, float *u, float *v);
int oflk_pyramidal_u8(const unsigned char *prev, const unsigned char *curr, int B, int H, int W,
                      int levels, int window_size, int iters, float *u, float *v, float *residual_log,
                      int *iters_run);
/* device-side conversion for pipelines that hold uint8 frames in HBM: d_out[i] = (float)d_in[i] */
int oflk_u8_to_f32(const unsigned char *d_in, float *d_out, size_t n, void *stream);

/* ---- device-resident plan API (pipelines, bench) -------------------------- */
typedef struct oflk_plan oflk_plan;

/* Allocate the workspace (pyramids, flow ping-pong buffers, reduction scratch)
 * for B pairs of H x W on `device`.  levels = 1 and iters = 0 gives a plan that
 * can only run oflk_plan_single_scale. */
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

/* After oflk_plan_pyramidal: copy the residual log / iteration counts of the last
 * enqueued pass to the host (synchronises `stream`).  Either pointer may be NULL. */
int oflk_plan_read_log(oflk_plan *plan, float *residual_log, int *iters_run, void *stream);

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

#ifdef __cplusplus
}
#endif
#endif /* OFLK_H */
