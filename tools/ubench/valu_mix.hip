// Microbenchmark: issue cost of the VALU instruction forms that dominate k_lk5, at 4 waves per
// SIMD (the kernel's occupancy).  Inline asm so the exact opcode/encoding is what is timed.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k(float *out, int iters)
{
    float a = threadIdx.x * 0.001f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    float e = a + 4.0f, f = a + 5.0f, g = a + 6.0f, h = a + 7.0f;
    int ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;
    double da = a, db = b, dc = c, dd = d;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (KIND == 2) { REP8(asm volatile("v_fma_f32 %0, %0, %4, 0.5\n v_fma_f32 %1, %1, %4, 0.5\n v_fma_f32 %2, %2, %4, 0.5\n v_fma_f32 %3, %3, %4, 0.5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 3) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
        if (KIND == 4) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
        if (KIND == 5) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));) }
        if (KIND == 6) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));) }
        if (KIND == 7) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));) }
        if (KIND == 8) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %0\n v_lshl_add_u64 %1, %1, 2, %1\n v_lshl_add_u64 %2, %2, 2, %2\n v_lshl_add_u64 %3, %3, 2, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));) }
        if (KIND == 9) { REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 10) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da));) }
        if (KIND == 11) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da));) }
        if (KIND == 12) { REP8(asm volatile("v_med3_i32 %0, %0, 0, %4\n v_med3_i32 %1, %1, 0, %4\n v_med3_i32 %2, %2, 0, %4\n v_med3_i32 %3, %3, 0, %4" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));) }
        if (KIND == 13) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 14) { REP8(asm volatile("v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (KIND == 15) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %5\n v_div_scale_f32 %1, vcc, %1, %4, %5\n v_div_scale_f32 %2, vcc, %2, %4, %5\n v_div_scale_f32 %3, vcc, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");) }
        if (KIND == 16) { REP8(asm volatile("v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");) }
        if (KIND == 17) { REP8(asm volatile("v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da));) }
        if (KIND == 18) { REP8(asm volatile("v_cmp_le_u64 vcc, %0, %4\n v_cmp_le_u64 vcc, %1, %4\n v_cmp_le_u64 vcc, %2, %4\n v_cmp_le_u64 vcc, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da) : "vcc");) }
        if (KIND == 19) { REP8(asm volatile("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 20) { REP8(asm volatile("v_add_f32_dpp %0, %4, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %4, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 21) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da));) }
        if (KIND == 22) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da));) }
        if (KIND == 23) { REP8(asm volatile("v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_add_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 24) { REP8(asm volatile("v_pk_fma_f16 %0, %0, %4, %4\n v_pk_fma_f16 %1, %1, %4, %4\n v_pk_fma_f16 %2, %2, %4, %4\n v_pk_fma_f16 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 25) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(a), "v"(b), "v"(c), "v"(d));) }
        if (KIND == 26) { REP8(asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));) }
        if (KIND == 27) { REP8(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(da), "v"(db), "v"(dc), "v"(dd));) }
        if (KIND == 28) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(da), "v"(db), "v"(dc), "v"(dd));) }
        if (KIND == 29) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(da));) }
        if (KIND == 30) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %1, %1, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
    }
    float s = a + b + c + d + e + f + g + h + ia + ib + ic + id + (float)(da + db + dc + dd);
    if (s == 12345.678f) out[1] = s;
}

template <int KIND>
void run(const char *name, float *d_out)
{
    const int iters = 4000;
    for (int wps : {1, 4}) {
        dim3 block(64 * 4 * wps);
        int blocks = 256;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), block, 0, 0, d_out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), block, 0, 0, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double ns = ms * 1e6 / ((double)iters * 32 * wps);
        printf("%-22s waves/SIMD=%d  %.2f ns per wave-instr per SIMD\n", name, wps, ns);
    }
}

int main()
{
    float *d_out; (void)hipMalloc(&d_out, 64); (void)hipMemset(d_out, 0, 64);
    run<0>("v_add_f32 (VOP2)", d_out);
    run<1>("v_fma_f32 3 vgpr", d_out);
    run<2>("v_fma_f32 const", d_out);
    run<3>("v_cndmask_b32 vcc", d_out);
    run<4>("v_cmp_gt_f32 vcc", d_out);
    run<5>("v_add_u32", d_out);
    run<6>("v_mad_u32_u24", d_out);
    run<7>("v_mul_lo_u32", d_out);
    run<8>("v_lshl_add_u64", d_out);
    run<9>("v_mov_b32", d_out);
    run<10>("v_pk_add_f32", d_out);
    run<11>("v_pk_fma_f32", d_out);
    run<12>("v_med3_i32", d_out);
    run<13>("v_rcp_f32", d_out);
    run<14>("v_div_fixup_f32", d_out);
    run<15>("v_div_scale_f32", d_out);
    run<16>("v_div_fmas_f32", d_out);
    run<17>("v_min_f64", d_out);
    run<18>("v_cmp_le_u64 vcc", d_out);
    run<19>("v_mov_b32_dpp row_shr", d_out);
    run<20>("v_add_f32_dpp wave_shr", d_out);
    run<21>("v_mul_f64", d_out);
    run<22>("v_pk_mul_f32", d_out);
    run<23>("v_pk_add_f16", d_out);
    run<24>("v_pk_fma_f16", d_out);
    run<25>("v_cvt_f64_f32", d_out);
    run<26>("v_floor_f64", d_out);
    run<27>("v_cvt_i32_f64", d_out);
    run<28>("v_cvt_f32_f64", d_out);
    run<29>("v_add_f64", d_out);
    run<30>("cndmask/add mix", d_out);
    return 0;
}
