// Co-run partners for tools/corun.py (development tool): kernels that occupy exactly one pipe of the chip, to be
// run on a second stream beside the LK kernels.  Whichever partner slows the LK step in proportion to what it
// takes is competing for the pipe that binds the LK kernels.
//   spin_valu : every wave loops over 32 independent v_add_f32 (mode 1), 32 v_add_f64 (mode 2), or only sleeps
//               (mode 0: the residency control -- same registers / wave slots taken, no instruction issue);
//               `nops` s_nop 7 per 32 instructions thin the stream
//   spin_hbm  : grid-stride float4 copy over a buffer far larger than the Infinity Cache
// Each wave writes {cycles, realtime ticks, HW_ID, XCC_ID, start, end} so the caller can see where and how fast
// the partner ran.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libspin.so tools/ubench/spinner.hip
#include <hip/hip_runtime.h>

#define OP32(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x));
#define OP64(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(dx));
#define R4(X) X X X X
#define I_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %8\n"

template <int MODE>
__global__ void k_spin(unsigned *out, long iters, int nops, float seed)
{
    float r[8];
    double d[8];
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 0.001f + i + seed; d[i] = r[i] * 1.0000001; }
    float x = seed + 1.0000001f;
    double dx = x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();
    for (long it = 0; it < iters; it++) {
        if (MODE == 0) { __builtin_amdgcn_s_sleep(2); }
        if (MODE == 1) { R4(OP32(I_ADD_F32)) }
        if (MODE == 2) { R4(OP64(I_ADD_F64)) }
        for (int n = 0; n < nops; n++) asm volatile("s_nop 7");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long q1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i] + (float)d[i];
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 8;
        o[0] = (unsigned)((t1 - t0) >> 8) + (s == 12345.678f);   // cycles / 256
        o[1] = (unsigned)(q1 - q0);
        o[2] = (unsigned)q0;
        o[3] = (unsigned)q1;
        o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
}

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n, int reps)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (int rep = 0; rep < reps; rep++)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

extern "C" {
__attribute__((visibility("default"))) int spin_valu(void *stream, int mode, int blocks, int threads, long iters, int nops,
                                                     unsigned *d_out)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_spin<0>, dim3(blocks), dim3(threads), 0, s, d_out, iters, nops, 1.0f);
    else if (mode == 1) hipLaunchKernelGGL(k_spin<1>, dim3(blocks), dim3(threads), 0, s, d_out, iters, nops, 1.0f);
    else hipLaunchKernelGGL(k_spin<2>, dim3(blocks), dim3(threads), 0, s, d_out, iters, nops, 1.0f);
    return (int)hipGetLastError();
}
__attribute__((visibility("default"))) int spin_hbm(void *stream, const void *src, void *dst, size_t n_vec, int reps, int blocks)
{
    hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n_vec, reps);
    return (int)hipGetLastError();
}
}
