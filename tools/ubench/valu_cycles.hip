// Microbenchmark: SIMD cycles per wave-instruction of the VALU forms the LK kernels are made of,
// measured IN CYCLES with s_memtime inside the kernel (a clock-independent figure: the chip's
// clock differs between a short microbenchmark and a sustained kernel), at 1, 2, 4 and 8 waves
// per SIMD, eight independent chains per wave.  Development tool; its table feeds
// tools/valu_floor.py (the VALU-issue bound bench.py reports beside the HBM bound).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_cycles tools/ubench/valu_cycles.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define R4(X) X X X X
// body: 8 instructions on 8 independent chains (r0..r7 32-bit, d0..d7 64-bit)
#define OP32(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
    : "v"(x), "v"(y) : "vcc");
#define OP64(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
    : "v"(dx), "v"(dy) : "vcc");
// mixed: 32-bit result from 64-bit source or the reverse
#define OPM(INS) asm volatile(INS(0, 8) INS(1, 9) INS(2, 10) INS(3, 11) INS(4, 12) INS(5, 13) INS(6, 14) INS(7, 15) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
      "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
    : "v"(x), "v"(y) : "vcc");

#define I_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_MED3(i) "v_med3_i32 %" #i ", %" #i ", 0, %8\n"
#define I_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define I_CMP_F32(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_DIV_SCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %9\n"
#define I_DIV_FMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define I_DIV_FIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD_DPP(i) "v_add_f32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_PK_ADD_F16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define I_PK_FMA_F16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define I_PK_MUL_F16(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
#define I_CVT_F16(i) "v_cvt_f16_f32 %" #i ", %" #i "\n"
#define I_CVT_PKRTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n"
#define I_CVT_F32_UB0(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define I_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define I_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define I_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define I_MIN_F64(i) "v_min_f64 %" #i ", %" #i ", %8\n"
#define I_FLOOR_F64(i) "v_floor_f64 %" #i ", %" #i "\n"
#define I_CMP_U64(i) "v_cmp_le_u64 vcc, %" #i ", %8\n"
#define I_PK_ADD_F32(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define I_PK_MUL_F32(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define I_PK_FMA_F32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_CVT_F64_F32(i, j) "v_cvt_f64_f32 %" #j ", %" #i "\n"
#define I_CVT_F32_F64(i, j) "v_cvt_f32_f64 %" #i ", %" #j "\n"
#define I_CVT_F64_I32(i, j) "v_cvt_f64_i32 %" #j ", %" #i "\n"
#define I_CVT_I32_F64(i, j) "v_cvt_i32_f64 %" #i ", %" #j "\n"

enum { K_ADD_F32, K_FMA_F32, K_MUL_F32, K_MOV, K_ADD_U32, K_MUL24, K_MAD24, K_MED3, K_LSHL_ADD, K_CMP_F32, K_CNDMASK,
       K_RCP, K_DIV_SCALE, K_DIV_FMAS, K_DIV_FIXUP, K_MOV_DPP, K_ADD_DPP, K_PK_ADD_F16, K_PK_FMA_F16, K_PK_MUL_F16,
       K_CVT_F16, K_CVT_PKRTZ, K_CVT_UB0, K_ADD_F64, K_MUL_F64, K_FMA_F64, K_MIN_F64, K_FLOOR_F64, K_CMP_U64, K_PK_ADD_F32,
       K_PK_MUL_F32, K_PK_FMA_F32, K_CVT_F64_F32, K_CVT_F32_F64, K_CVT_F64_I32, K_CVT_I32_F64, K_MIX_F64_F32, K_COUNT };
const char *kNames[K_COUNT] = {"v_add_f32", "v_fma_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_mul_u32_u24", "v_mad_u32_u24",
    "v_med3_i32", "v_lshl_add_u32", "v_cmp_gt_f32", "v_cndmask_b32", "v_rcp_f32", "v_div_scale_f32", "v_div_fmas_f32",
    "v_div_fixup_f32", "v_mov_b32_dpp", "v_add_f32_dpp", "v_pk_add_f16", "v_pk_fma_f16", "v_pk_mul_f16", "v_cvt_f16_f32",
    "v_cvt_pkrtz_f16_f32", "v_cvt_f32_ubyte0", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_min_f64", "v_floor_f64", "v_cmp_le_u64",
    "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f64_i32", "v_cvt_i32_f64",
    "mix add_f64/add_f32 1:1"};

template <int KIND>
__global__ void k(unsigned *out, int iters, float seed)
{
    float r[8];
    double d[8];
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 0.001f + i + seed; d[i] = r[i] * 1.0000001; }
    float x = seed + 1.0000001f, y = seed + 0.5f;
    double dx = x, dy = y;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it++) {
        if (KIND == K_ADD_F32) { R4(OP32(I_ADD_F32)) }
        if (KIND == K_FMA_F32) { R4(OP32(I_FMA_F32)) }
        if (KIND == K_MUL_F32) { R4(OP32(I_MUL_F32)) }
        if (KIND == K_MOV) { R4(OP32(I_MOV)) }
        if (KIND == K_ADD_U32) { R4(OP32(I_ADD_U32)) }
        if (KIND == K_MUL24) { R4(OP32(I_MUL24)) }
        if (KIND == K_MAD24) { R4(OP32(I_MAD24)) }
        if (KIND == K_MED3) { R4(OP32(I_MED3)) }
        if (KIND == K_LSHL_ADD) { R4(OP32(I_LSHL_ADD)) }
        if (KIND == K_CMP_F32) { R4(OP32(I_CMP_F32)) }
        if (KIND == K_CNDMASK) { R4(OP32(I_CNDMASK)) }
        if (KIND == K_RCP) { R4(OP32(I_RCP)) }
        if (KIND == K_DIV_SCALE) { R4(OP32(I_DIV_SCALE)) }
        if (KIND == K_DIV_FMAS) { R4(OP32(I_DIV_FMAS)) }
        if (KIND == K_DIV_FIXUP) { R4(OP32(I_DIV_FIXUP)) }
        if (KIND == K_MOV_DPP) { R4(OP32(I_MOV_DPP)) }
        if (KIND == K_ADD_DPP) { R4(OP32(I_ADD_DPP)) }
        if (KIND == K_PK_ADD_F16) { R4(OP32(I_PK_ADD_F16)) }
        if (KIND == K_PK_FMA_F16) { R4(OP32(I_PK_FMA_F16)) }
        if (KIND == K_PK_MUL_F16) { R4(OP32(I_PK_MUL_F16)) }
        if (KIND == K_CVT_F16) { R4(OP32(I_CVT_F16)) }
        if (KIND == K_CVT_PKRTZ) { R4(OP32(I_CVT_PKRTZ)) }
        if (KIND == K_CVT_UB0) { R4(OP32(I_CVT_F32_UB0)) }
        if (KIND == K_ADD_F64) { R4(OP64(I_ADD_F64)) }
        if (KIND == K_MUL_F64) { R4(OP64(I_MUL_F64)) }
        if (KIND == K_FMA_F64) { R4(OP64(I_FMA_F64)) }
        if (KIND == K_MIN_F64) { R4(OP64(I_MIN_F64)) }
        if (KIND == K_FLOOR_F64) { R4(OP64(I_FLOOR_F64)) }
        if (KIND == K_CMP_U64) { R4(OP64(I_CMP_U64)) }
        if (KIND == K_PK_ADD_F32) { R4(OP64(I_PK_ADD_F32)) }
        if (KIND == K_PK_MUL_F32) { R4(OP64(I_PK_MUL_F32)) }
        if (KIND == K_PK_FMA_F32) { R4(OP64(I_PK_FMA_F32)) }
        if (KIND == K_CVT_F64_F32) { R4(OPM(I_CVT_F64_F32)) }
        if (KIND == K_CVT_F32_F64) { R4(OPM(I_CVT_F32_F64)) }
        if (KIND == K_CVT_F64_I32) { R4(OPM(I_CVT_F64_I32)) }
        if (KIND == K_CVT_I32_F64) { R4(OPM(I_CVT_I32_F64)) }
        if (KIND == K_MIX_F64_F32) { OP64(I_ADD_F64) OP32(I_ADD_F32) OP64(I_ADD_F64) OP32(I_ADD_F32) }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long q1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i] + (float)d[i];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (unsigned)(t1 - t0) + (s == 12345.678f);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[16384] = (unsigned)(t1 - t0); out[16385] = (unsigned)(q1 - q0); }
}

template <int KIND>
void run(unsigned *d_out)
{
    const int iters = 20000;
    printf("%-26s", kNames[KIND]);
    for (int wps : {1, 2, 4, 8}) {
        // wps waves on each SIMD: one block per CU (two of 1024 threads for wps = 8)
        const int threads = wps == 8 ? 1024 : 256 * wps, blocks = wps == 8 ? 512 : 256;
        const int nw = blocks * threads / 64;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 200, 1.0f);   // warm the clock
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0f);
        std::vector<unsigned> h(nw);
        (void)hipMemcpy(h.data(), d_out, nw * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per_wave = (double)h[nw / 2] / ((double)iters * 32.0);   // cycles between a wave's own instructions
        printf("  w%d: %5.2f", wps, per_wave / wps);                      // SIMD ticks per wave-instruction
        if (wps == 8) {
            unsigned tr[2];
            (void)hipMemcpy(tr, d_out + 16384, 8, hipMemcpyDeviceToHost);
            printf("  [s_memtime ticks per ns: %.3f]", (double)tr[0] / ((double)tr[1] * 10.0));
        }
    }
    printf("   (SIMD cycles per wave-instruction)\n");
    fflush(stdout);
}

template <int K0>
void run_all(unsigned *d_out)
{
    if constexpr (K0 < K_COUNT) {
        run<K0>(d_out);
        run_all<K0 + 1>(d_out);
    }
}

int main()
{
    unsigned *d_out;
    (void)hipMalloc(&d_out, (16384 + 16) * sizeof(unsigned));
    run_all<0>(d_out);
    return 0;
}
