// Microbenchmark: issue cost of the fp64-class instructions the exact bilinear uses
// (development tool).  8 waves per SIMD, independent chains, reports ns per wave-instruction
// per SIMD; compare with v_add_f32 = ~0.93 ns.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void k(float *out, int iters)
{
    double d[8];
    float f[8];
    int q[8];
    for (int i = 0; i < 8; i++) { d[i] = threadIdx.x * 0.001 + i + out[2]; f[i] = (float)d[i]; q[i] = threadIdx.x + i; }
    double inc = out[0] + 1.0000001;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) d[i] = d[i] + inc;
                else if (KIND == 1) d[i] = d[i] * inc;
                else if (KIND == 2) d[i] = __builtin_fma(d[i], inc, inc);
                else if (KIND == 3) { d[i] = (double)f[i]; asm volatile("" : "+v"(d[i])); f[i] += 1.0f; }   // cvt_f64_f32 (+ add_f32)
                else if (KIND == 4) { f[i] = (float)d[i]; asm volatile("" : "+v"(f[i])); }                    // cvt_f32_f64
                else if (KIND == 5) { d[i] = __builtin_floor(d[i]); asm volatile("" : "+v"(d[i])); }          // floor_f64
                else if (KIND == 6) { q[i] = (int)d[i]; asm volatile("" : "+v"(q[i])); }                      // cvt_i32_f64
                else if (KIND == 7) { d[i] = (double)q[i]; asm volatile("" : "+v"(d[i])); }                   // cvt_f64_i32
                else if (KIND == 8) { f[i] = f[i] / (f[(i + 1) & 7] + 3.0f); }                                // IEEE f32 divide (+ add)
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += (float)d[i] + f[i] + q[i];
    if (s == 12345.678f) out[1] = s;
}

template <int KIND>
void run(const char *name, float *d_out, double per_iter)
{
    const int iters = 4000, wps = 8;
    dim3 block(1024);
    int blocks = 256 * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), block, 0, 0, d_out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), block, 0, 0, d_out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / ((double)iters * per_iter * wps);
    printf("%-22s %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, ms, ns);
}

int main()
{
    float *d_out; (void)hipMalloc(&d_out, 64); (void)hipMemset(d_out, 0, 64);
    run<0>("v_add_f64", d_out, 32);
    run<1>("v_mul_f64", d_out, 32);
    run<2>("v_fma_f64", d_out, 32);
    run<3>("cvt_f64_f32 (+add_f32)", d_out, 32);
    run<4>("cvt_f32_f64", d_out, 32);
    run<5>("floor_f64", d_out, 32);
    run<6>("cvt_i32_f64", d_out, 32);
    run<7>("cvt_f64_i32", d_out, 32);
    run<8>("div_f32 IEEE (+add)", d_out, 32);
    return 0;
}
