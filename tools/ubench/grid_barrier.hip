// What does it cost to order two phases of a small grid on MI355X: a dependent kernel launch, or a device-scope barrier
// inside one (co-resident) grid?  Development tool behind DESIGN.md section 7 (one frame pair per call: 14 dependent
// launches; would one launch with grid barriers be faster?).
//
//   boundary : K launches of a kernel whose blocks each write `bytes` bytes and exit, back to back on one stream
//   barrier  : ONE launch; every block writes the same bytes, then release fence + atomic arrive + spin + acquire fence,
//              K times.  The spin is bounded: a block that does not see the others within ~20 ms gives up, counts a
//              failure and goes on, so every wave reaches the end of the kernel whatever happens.
// Grid sizes are the block counts of the pyramid levels of one 1080p pair (96, 345, 1 024 = every slot of the chip at
// four blocks per CU).  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/grid_barrier tools/ubench/grid_barrier.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

__global__ __launch_bounds__(256) void k_phase(float4 *buf, int vec_per_thread, float seed)
{
    float4 *p = buf + (size_t)blockIdx.x * 256 * vec_per_thread + threadIdx.x;
    for (int i = 0; i < vec_per_thread; i++) p[(size_t)i * 256] = make_float4(seed, seed, seed, seed);
}

__global__ __launch_bounds__(256) void k_barrier(float4 *buf, int vec_per_thread, float seed, unsigned *ctr, unsigned *fails,
                                                  int rounds)
{
    float4 *p = buf + (size_t)blockIdx.x * 256 * vec_per_thread + threadIdx.x;
    const unsigned nb = gridDim.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < vec_per_thread; i++) p[(size_t)i * 256] = make_float4(seed + r, seed, seed, seed);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();   // release at device scope: the block's writes are visible to every XCD
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nb;
            int spins = 0;
            // (once any block has given up, nobody waits any more: the launch then ends within microseconds)
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target &&
                   __hip_atomic_load(fails, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++spins > 200000) {   // ~20 ms: give up, never hang
                    atomicAdd(fails, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __threadfence();
        }
        __syncthreads();
    }
}

int main()
{
    const int K = 200;
    const size_t max_bytes = (size_t)1024 * 65536;
    float4 *buf = nullptr;
    unsigned *ctr = nullptr;
    CHECK(hipMalloc((void **)&buf, max_bytes));
    CHECK(hipMalloc((void **)&ctr, 2 * sizeof(unsigned)));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("%-8s %-14s %-22s %-22s\n", "blocks", "bytes/block", "us per launch boundary", "us per grid barrier");
    for (int nb : {96, 345, 1024}) {
        for (int vec : {0, 1, 16}) {   // 0, 4 KB, 64 KB written per block and phase
            // warm-up, then K dependent launches
            for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_phase, dim3(nb), dim3(256), 0, s, buf, vec, 1.0f);
            CHECK(hipStreamSynchronize(s));
            CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_phase, dim3(nb), dim3(256), 0, s, buf, vec, 2.0f + i);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms_launch = 0.0f;
            CHECK(hipEventElapsedTime(&ms_launch, e0, e1));
            // one launch with K barriers (and one warm-up launch before it)
            float ms_bar = 0.0f;
            unsigned fails = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipMemsetAsync(ctr, 0, 2 * sizeof(unsigned), s));
                CHECK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_barrier, dim3(nb), dim3(256), 0, s, buf, vec, 3.0f, ctr, ctr + 1, K);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventElapsedTime(&ms_bar, e0, e1));
                unsigned h[2];
                CHECK(hipMemcpy(h, ctr, sizeof(h), hipMemcpyDeviceToHost));
                fails = h[1];
            }
            std::printf("%-8d %-14d %-22.2f %-22.2f%s\n", nb, vec * 256 * 16, 1e3 * ms_launch / K, 1e3 * ms_bar / K,
                        fails ? "   (spin gave up: blocks not co-resident?)" : "");
        }
    }
    return 0;
}
