// Microbenchmark: cost of the warp's 8-byte pair gathers by alignment (development tool).
// Every lane loads {p[i], p[i+1]} as one 8-byte access; i = lane + shift, so consecutive lanes overlap
// by one element, as the bilinear taps of adjacent cells do.  shift 0: even lanes 8-byte aligned, odd
// lanes 4 bytes off; variants force all-aligned (lane * 2) and all-misaligned (lane * 2 + 1) pairs, and
// the same bytes as two 4-byte loads.  Working set 8 MB (L2 / Infinity Cache resident).
#include <hip/hip_runtime.h>
#include <cstdio>

struct __attribute__((packed, aligned(4))) PairF { float a, b; };

template <int KIND>
__global__ __launch_bounds__(256) void k(const float *__restrict__ p, float *__restrict__ out, int W, int rows, int iters)
{
    const int lane = threadIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        const int row = (blockIdx.x * 7 + it * 13) % rows;
        const float *r = p + (size_t)row * W + (blockIdx.x & 7) * 256;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
            const float *q = r + k2 * 2048;
            if (KIND == 0) { PairF v = *reinterpret_cast<const PairF *>(q + lane); acc += v.a + v.b; }               // overlapping, mixed alignment
            if (KIND == 1) { PairF v = *reinterpret_cast<const PairF *>(q + 2 * lane); acc += v.a + v.b; }           // all 8-byte aligned
            if (KIND == 2) { PairF v = *reinterpret_cast<const PairF *>(q + 2 * lane + 1); acc += v.a + v.b; }       // all 4 bytes off
            if (KIND == 3) { acc += q[lane] + q[lane + 1]; }                                                          // two 4-byte loads
            if (KIND == 4) { acc += q[lane]; }                                                                        // one 4-byte load (coalesced)
            if (KIND == 5) { float2 v = *reinterpret_cast<const float2 *>(q + 2 * lane); acc += v.x + v.y; }         // declared aligned float2
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int KIND>
void run(const char *name, const float *p, float *out, int W, int rows)
{
    const int iters = 400, blocks = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, p, out, W, rows, 20);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, p, out, W, rows, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * 8;
    printf("%-44s %8.3f ms   %6.2f ns per wave-load per CU\n", name, ms, ms * 1e6 / (wave_instr / 256.0));
}

int main()
{
    const int W = 20480, rows = 100;   // 8 MB
    float *p, *out;
    (void)hipMalloc(&p, (size_t)W * rows * 4 + 65536);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(p, 0, (size_t)W * rows * 4 + 65536);
    run<4>("dword, coalesced", p, out, W, rows);
    run<5>("float2 aligned (2 x lane)", p, out, W, rows);
    run<1>("pair, all 8-byte aligned (2 x lane)", p, out, W, rows);
    run<2>("pair, all 4 bytes off (2 x lane + 1)", p, out, W, rows);
    run<0>("pair, overlapping lanes (lane): the warp's", p, out, W, rows);
    run<3>("two dwords, overlapping lanes", p, out, W, rows);
    return 0;
}
