// Microbenchmark: VALU issue rate per SIMD vs waves per SIMD (development tool).
// Each wave runs N iterations of 32 independent ops (fp32 add / pk add / fp64 add / int add).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void k(float *out, int iters)
{
    float a[16];
    double d[8];
    int q[16];
    float2 p[8];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; q[i] = threadIdx.x + i; }
    for (int i = 0; i < 8; i++) { d[i] = threadIdx.x * 0.001 + i; p[i] = make_float2(a[i], a[i + 8]); }
    float inc = out[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = a[i] + inc;
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) { p[i].x += inc; p[i].y += inc; }
#pragma unroll
                for (int i = 0; i < 8; i++) { p[i].x += inc; p[i].y += inc; }
            } else if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) d[i] = d[i] + (double)inc;
#pragma unroll
                for (int i = 0; i < 8; i++) d[i] = d[i] + (double)inc;
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) q[i] = q[i] + (int)inc + i;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + q[i];
    for (int i = 0; i < 8; i++) s += (float)d[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[1] = s;
}

template <int KIND>
void run(const char *name, float *d_out)
{
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 8}) {
        // 256 CUs * 4 SIMDs * wps waves; blocks of 64 threads * (4*wps) waves = one block per CU
        dim3 block(64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps);
        int blocks = 256 * ((64 * 4 * wps) / block.x);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), block, 0, 0, d_out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), block, 0, 0, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double instr_per_wave = (double)iters * 32;
        // cycles per instruction per SIMD at 2.4 GHz nominal (clock may be lower)
        double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * wps);
        printf("%-8s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ns_per_instr_simd, ns_per_instr_simd * 2.4);
    }
}

int main()
{
    float *d_out; (void)hipMalloc(&d_out, 64); (void)hipMemset(d_out, 0, 64);
    run<0>("add_f32", d_out);
    run<1>("pk_add", d_out);
    run<2>("add_f64", d_out);
    run<3>("add_i32", d_out);
    return 0;
}
