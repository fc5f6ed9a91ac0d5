// Microbenchmark: is v_cndmask_b32 slow on gfx950?  Several forms, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k(float *out, int iters)
{
    float a = threadIdx.x * 0.001f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f, e = a + 4.0f;
    for (int it = 0; it < iters; it++) {
        // 0: e32 form, vcc set once per group by s_mov
        if (KIND == 0) { REP8(asm volatile("s_mov_b64 vcc, exec\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
        // 1: e64 form with an SGPR pair mask
        if (KIND == 1) { REP8(asm volatile("s_mov_b64 s[40:41], exec\n v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_cndmask_b32_e64 %1, %1, %4, s[40:41]\n v_cndmask_b32_e64 %2, %2, %4, s[40:41]\n v_cndmask_b32_e64 %3, %3, %4, s[40:41]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s40", "s41");) }
        // 2: v_cmp -> v_cndmask pairs (the usual select)
        if (KIND == 2) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
        // 3: same number of instructions, but max instead of cmp+cndmask
        if (KIND == 3) { REP8(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %1, %1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        // 4: cndmask interleaved with independent adds
        if (KIND == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %1, %1, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
        // 5: compiler-generated selects (C++ ternary), whatever form hipcc picks
        if (KIND == 5) {
#pragma unroll
            for (int r = 0; r < 8; r++) { a = a > e ? a : b; b = b > e ? b + 1.0f : c; c = c > e ? c : d; d = d > e ? d : a * 0.5f; }
        }
    }
    float s = a + b + c + d + e;
    if (s == 12345.678f) out[1] = s;
}
template <int KIND>
void run(const char *name, float *d_out, int per_iter)
{
    const int iters = 4000, wps = 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * 4 * wps), 0, 0, d_out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * 4 * wps), 0, 0, d_out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.2f ns per wave-instr per SIMD (%d instr/iter)\n", name, ms * 1e6 / ((double)iters * per_iter * wps), per_iter);
}
int main()
{
    float *d_out; (void)hipMalloc(&d_out, 64); (void)hipMemset(d_out, 0, 64);
    run<0>("cndmask e32, vcc = exec (s_mov)", d_out, 40);
    run<1>("cndmask e64, sgpr pair", d_out, 40);
    run<2>("v_cmp + cndmask pairs", d_out, 32);
    run<3>("v_max x2 (reference)", d_out, 32);
    run<4>("cndmask + add interleaved", d_out, 32);
    run<5>("C++ ternaries (hipcc)", d_out, 64);
    return 0;
}
