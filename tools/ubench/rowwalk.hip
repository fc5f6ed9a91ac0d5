// Microbenchmark: what the memory system gives a 2-plane-in / 2-plane-out stream over 8K planes, by access
// pattern (development tool).  16 bytes of traffic per pixel, like the single-scale LK kernels.
//   flat4     grid-stride, 16 bytes per lane (what a library elementwise kernel does)
//   walk1     one wave per 64-column strip walking down HS rows, 4 bytes per lane, PF rows of loads in flight
//             (the access pattern of k_lk16s, no halo)
//   walk1h    same with the 8 halo columns of k_lk16s<3>: 64 columns loaded, 56 stored
//   walk4     one wave per 256-column strip, 16 bytes per lane
//   walk1_ld / walk1_st   the walk1 pattern, loads only / stores only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int H = 4320, W = 7680;

__global__ __launch_bounds__(256) void flat4(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ u,
                                             float4 *__restrict__ v, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        u[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        v[i] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
    }
}

// MODE 0: loads + stores, 1: loads only, 2: stores only.  OUTW: columns stored per wave (64 or 56); VEC: floats per lane
template <int MODE, int OUTW, int VEC, int PF>
__global__ __launch_bounds__(256) void walk(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ u,
                                            float *__restrict__ v, int hs, int segs)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int strips = (W + OUTW * VEC - 1) / (OUTW * VEC);
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (task >= strips * segs) return;
    const int seg = task / strips, strip = task - seg * strips;
    const int halo = (64 - OUTW) / 2;
    int x = (strip * OUTW - halo + lane) * VEC;
    const bool out_lane = lane >= halo && lane < 64 - halo && x < W;
    x = min(max(x, 0), W - VEC);
    const int ys = seg * hs, ye = min(ys + hs, H);
    vec_t pa[PF], pb[PF];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const size_t o = (size_t)min(ys + k, H - 1) * W + x;
        if (MODE != 2) {
            pa[k] = *reinterpret_cast<const vec_t *>(a + o);
            pb[k] = *reinterpret_cast<const vec_t *>(b + o);
        }
    }
    for (int r = ys; r < ye; r++) {
        vec_t p = pa[0], q = pb[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; k++) {
            pa[k] = pa[k + 1];
            pb[k] = pb[k + 1];
        }
        if (MODE != 2) {
            const size_t o = (size_t)min(r + PF, H - 1) * W + x;
            pa[PF - 1] = *reinterpret_cast<const vec_t *>(a + o);
            pb[PF - 1] = *reinterpret_cast<const vec_t *>(b + o);
        } else {
            p = q = (vec_t)(float)r;
        }
        const vec_t s = p + q, d = p - q;
        if (MODE == 1) {
            const vec_t sd = s + d;
            acc += sd[0] + sd[VEC - 1];
        } else if (out_lane) {
            const size_t o = (size_t)r * W + x;
            *reinterpret_cast<vec_t *>(u + o) = s;
            *reinterpret_cast<vec_t *>(v + o) = d;
        }
    }
    if (MODE == 1 && acc == 12345.678f) u[0] = acc;
}

template <class F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-58s %8.1f us   %5.2f TB/s\n", name, us, bytes / us / 1e6);
}

int main(int argc, char **argv)
{
    const int hs = argc > 1 ? atoi(argv[1]) : 76;
    const size_t n = (size_t)H * W;
    float *a, *b, *u, *v;
    (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMalloc(&u, n * 4); (void)hipMalloc(&v, n * 4);
    (void)hipMemset(a, 0, n * 4); (void)hipMemset(b, 0, n * 4);
    const int segs = (H + hs - 1) / hs;
    auto grid = [&](int outw, int vec) { return dim3((unsigned)((((W + outw * vec - 1) / (outw * vec)) * segs + 3) / 4)); };
    printf("rows per segment %d\n", hs);
    timeit("flat4: grid-stride, 16 B per lane", 16.0 * n, [&] { hipLaunchKernelGGL(flat4, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)a, (const float4 *)b, (float4 *)u, (float4 *)v, n / 4); });
    timeit("walk1: 64-col strips, 4 B per lane, PF 2", 16.0 * n, [&] { hipLaunchKernelGGL((walk<0, 64, 1, 2>), grid(64, 1), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk1: 64-col strips, 4 B per lane, PF 4", 16.0 * n, [&] { hipLaunchKernelGGL((walk<0, 64, 1, 4>), grid(64, 1), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk1h: 64 cols loaded, 56 stored (k_lk16s<3>), PF 2", 16.0 * n, [&] { hipLaunchKernelGGL((walk<0, 56, 1, 2>), grid(56, 1), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk1h, PF 4", 16.0 * n, [&] { hipLaunchKernelGGL((walk<0, 56, 1, 4>), grid(56, 1), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk2: 128-col strips, 8 B per lane, PF 2", 16.0 * n, [&] { hipLaunchKernelGGL((walk<0, 64, 2, 2>), grid(64, 2), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk4: 256-col strips, 16 B per lane, PF 2", 16.0 * n, [&] { hipLaunchKernelGGL((walk<0, 64, 4, 2>), grid(64, 4), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk1 loads only (8 B per pixel)", 8.0 * n, [&] { hipLaunchKernelGGL((walk<1, 64, 1, 2>), grid(64, 1), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk1 stores only (8 B per pixel)", 8.0 * n, [&] { hipLaunchKernelGGL((walk<2, 64, 1, 2>), grid(64, 1), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    timeit("walk4 stores only (8 B per pixel)", 8.0 * n, [&] { hipLaunchKernelGGL((walk<2, 64, 4, 2>), grid(64, 4), dim3(256), 0, 0, a, b, u, v, hs, segs); });
    return 0;
}
