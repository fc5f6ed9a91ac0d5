// Microbenchmark: vector-ALU throughput of gfx950 checked against WALL-CLOCK (HIP events), with a census of
// where the waves ran (HW_ID / XCC_ID per wave) instead of an assumed waves-per-SIMD figure.
// Round 2's tools/ubench/valu_cycles.hip divided per-wave s_memtime deltas by an assumed occupancy and priced
// v_add_f32 at 0.95 SIMD cycles per wave64 instruction (> 64 lanes per clock); this tool answers whether that
// time base is right: for every instruction form it prints
//   wall     : launch duration from HIP events
//   Ginstr/s : wave64 instructions per second over the whole chip
//   cyc/SIMD : SIMD cycles per wave-instruction = (SIMDs in use x clock x wall) / wave-instructions, with the
//              clock measured inside the kernel (s_memtime ticks per s_memrealtime tick x 100 MHz)
//   TFLOP/s  : lanes x flops per instruction x instructions / wall (157.3 TF is the fp32 vector spec)
//   census   : SIMDs that hosted a wave, and resident waves per SIMD (min / median / max) at mid-launch
// Development tool.  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_wall tools/ubench/valu_wall.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#define OP32(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
    : "v"(x), "v"(y) : "vcc");
#define OP64(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
    : "v"(dx), "v"(dy) : "vcc");
#define OPM(INS) asm volatile(INS(0, 8) INS(1, 9) INS(2, 10) INS(3, 11) INS(4, 12) INS(5, 13) INS(6, 14) INS(7, 15) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
      "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
    : "v"(x), "v"(y) : "vcc");
#define R4(X) X X X X
#define OP32S(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
    : "v"(x), "v"(y), "s"(smask) : "vcc");
#define OP32SW(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+s"(smask) \
    : "v"(x), "v"(y) : "vcc");

#define I_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define I_ADD_DPP(i) "v_add_f32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD_WSHR(i) "v_add_f32_dpp %" #i ", %8, %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define I_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define I_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define I_PK_ADD_F32(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define I_PK_FMA_F32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_CVT_F64_F32(i, j) "v_cvt_f64_f32 %" #j ", %" #i "\n"
#define I_CVT_F32_F64(i, j) "v_cvt_f32_f64 %" #i ", %" #j "\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_FLOOR_F64(i) "v_floor_f64 %" #i ", %" #i "\n"
#define I_CND_SGPR(i) "v_cndmask_b32_e64 %" #i ", %8, %9, %10\n"
#define I_CND_ZERO(i) "v_cndmask_b32_e64 %" #i ", 0, %9, vcc\n"
#define I_CND_SELF(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define I_CMP_CND(i) "v_cmp_gt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define I_CMP_F32(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define I_CMP_4CND(i) "v_cmp_gt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %8, %9, vcc\nv_cndmask_b32 %" #i ", %9, %" #i ", vcc\nv_cndmask_b32 %" #i ", %8, %" #i ", vcc\nv_cndmask_b32 %" #i ", %9, %" #i ", vcc\n"
#define I_CMP_4CND64(i) "v_cmp_gt_f32 vcc, %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %8, %9, vcc\nv_cndmask_b32_e64 %" #i ", %9, %" #i ", vcc\nv_cndmask_b32_e64 %" #i ", %8, %" #i ", vcc\nv_cndmask_b32_e64 %" #i ", %9, %" #i ", vcc\n"
#define I_CMP_SGPR(i) "v_cmp_gt_f32_e64 %8, %" #i ", %9\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define I_MED3(i) "v_med3_i32 %" #i ", %" #i ", 0, %8\n"
#define I_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define I_MUL24(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define I_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_ADD_LSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 2\n"
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 2, %" #i "\n"
#define I_DIV_SCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %9\n"
#define I_DIV_FMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define I_DIV_FIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MIN_F64(i) "v_min_f64 %" #i ", %" #i ", %8\n"
#define I_CMP_U64(i) "v_cmp_ge_u64 vcc, %" #i ", %8\n"
#define I_CVT_F64_I32(i, j) "v_cvt_f64_i32 %" #j ", %" #i "\n"
#define I_CVT_I32_F64(i, j) "v_cvt_i32_f64 %" #i ", %" #j "\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_FLOOR_F32(i) "v_floor_f32 %" #i ", %" #i "\n"
#define I_CVT_F32_U8(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"

enum { K_ADD_F32, K_FMA_F32, K_MUL_F32, K_ADD_U32, K_CNDMASK, K_ADD_DPP, K_ADD_WSHR, K_ADD_F64, K_MUL_F64, K_FMA_F64,
       K_PK_ADD_F32, K_PK_FMA_F32, K_CVT_F64_F32, K_CVT_F32_F64, K_RCP, K_FLOOR_F64, K_MIX,
       K_CND_SGPR, K_CND_ZERO, K_CND_SELF, K_CMP_CND, K_CMP_F32, K_CMP_SGPR, K_AND, K_FMAC, K_MED3, K_MAD24, K_MUL24, K_MUL_LO, K_ADD_LSHL, K_LSHL,
       K_DIV_SCALE, K_DIV_FMAS, K_DIV_FIXUP, K_MIN_F64, K_CMP_U64, K_CVT_F64_I32, K_CVT_I32_F64, K_MOV, K_MOV_DPP, K_FLOOR_F32, K_CVT_F32_U8, K_CMP_4CND, K_CMP_4CND64, K_COUNT };
struct Kind { const char *name; double flops; };   // flops per lane and instruction
const Kind kKinds[K_COUNT] = {
    {"v_add_f32", 1}, {"v_fma_f32", 2}, {"v_mul_f32", 1}, {"v_add_u32", 1}, {"v_cndmask_b32 (vcc)", 1},
    {"v_add_f32_dpp row_shr:1", 1}, {"v_add_f32_dpp wave_shr:1", 1}, {"v_add_f64", 1}, {"v_mul_f64", 1}, {"v_fma_f64", 2},
    {"v_pk_add_f32", 2}, {"v_pk_fma_f32", 4}, {"v_cvt_f64_f32", 1}, {"v_cvt_f32_f64", 1}, {"v_rcp_f32", 1}, {"v_floor_f64", 1},
    {"mix 3 add_f32 : 1 add_f64", 1},
    {"v_cndmask_b32_e64 (sgpr mask)", 1}, {"v_cndmask_b32_e64 0, v, vcc", 1}, {"v_cndmask_b32 dst=src0, vcc", 1}, {"v_cmp_gt_f32 + v_cndmask (pair = 2)", 1},
    {"v_cmp_gt_f32 vcc", 1}, {"v_cmp_gt_f32 sgpr", 1}, {"v_and_b32", 1}, {"v_fmac_f32", 2}, {"v_med3_i32", 1}, {"v_mad_i32_i24", 1}, {"v_mul_i32_i24", 1},
    {"v_mul_lo_u32", 1}, {"v_add_lshl_u32", 1}, {"v_lshlrev_b32", 1}, {"v_div_scale_f32", 1}, {"v_div_fmas_f32", 1}, {"v_div_fixup_f32", 1},
    {"v_min_f64", 1}, {"v_cmp_ge_u64", 1}, {"v_cvt_f64_i32", 1}, {"v_cvt_i32_f64", 1}, {"v_mov_b32", 1}, {"v_mov_b32_dpp row_shr:1", 1},
    {"v_floor_f32", 1}, {"v_cvt_f32_ubyte0", 1}, {"v_cmp + 4 v_cndmask_e32 (5 instr)", 1}, {"v_cmp + 4 v_cndmask_e64 vcc (5 instr)", 1}};

constexpr int kWordsPerWave = 8;

template <int KIND>
__global__ void k(unsigned *out, int iters, float seed)
{
    float r[8];
    double d[8];
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 0.001f + i + seed; d[i] = r[i] * 1.0000001; }
    float x = seed + 1.0000001f, y = seed + 0.5f;
    double dx = x, dy = y;
    unsigned long long smask = __builtin_amdgcn_read_exec() ^ (0x5555555555555555ull * (unsigned long long)(seed != 7.0f));
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it++) {
        if (KIND == K_ADD_F32) { R4(OP32(I_ADD_F32)) }
        if (KIND == K_FMA_F32) { R4(OP32(I_FMA_F32)) }
        if (KIND == K_MUL_F32) { R4(OP32(I_MUL_F32)) }
        if (KIND == K_ADD_U32) { R4(OP32(I_ADD_U32)) }
        if (KIND == K_CNDMASK) { R4(OP32(I_CNDMASK)) }
        if (KIND == K_ADD_DPP) { R4(OP32(I_ADD_DPP)) }
        if (KIND == K_ADD_WSHR) { R4(OP32(I_ADD_WSHR)) }
        if (KIND == K_ADD_F64) { R4(OP64(I_ADD_F64)) }
        if (KIND == K_MUL_F64) { R4(OP64(I_MUL_F64)) }
        if (KIND == K_FMA_F64) { R4(OP64(I_FMA_F64)) }
        if (KIND == K_PK_ADD_F32) { R4(OP64(I_PK_ADD_F32)) }
        if (KIND == K_PK_FMA_F32) { R4(OP64(I_PK_FMA_F32)) }
        if (KIND == K_CVT_F64_F32) { R4(OPM(I_CVT_F64_F32)) }
        if (KIND == K_CVT_F32_F64) { R4(OPM(I_CVT_F32_F64)) }
        if (KIND == K_RCP) { R4(OP32(I_RCP)) }
        if (KIND == K_FLOOR_F64) { R4(OP64(I_FLOOR_F64)) }
        if (KIND == K_CND_SGPR) { R4(OP32S(I_CND_SGPR)) }
        if (KIND == K_CND_ZERO) { R4(OP32(I_CND_ZERO)) }
        if (KIND == K_CND_SELF) { R4(OP32(I_CND_SELF)) }
        if (KIND == K_CMP_CND) { R4(OP32(I_CMP_CND)) }
        if (KIND == K_CMP_F32) { R4(OP32(I_CMP_F32)) }
        if (KIND == K_AND) { R4(OP32(I_AND)) }
        if (KIND == K_FMAC) { R4(OP32(I_FMAC)) }
        if (KIND == K_MED3) { R4(OP32(I_MED3)) }
        if (KIND == K_MAD24) { R4(OP32(I_MAD24)) }
        if (KIND == K_MUL24) { R4(OP32(I_MUL24)) }
        if (KIND == K_MUL_LO) { R4(OP32(I_MUL_LO)) }
        if (KIND == K_ADD_LSHL) { R4(OP32(I_ADD_LSHL)) }
        if (KIND == K_LSHL) { R4(OP32(I_LSHL)) }
        if (KIND == K_DIV_SCALE) { R4(OP32(I_DIV_SCALE)) }
        if (KIND == K_DIV_FMAS) { R4(OP32(I_DIV_FMAS)) }
        if (KIND == K_DIV_FIXUP) { R4(OP32(I_DIV_FIXUP)) }
        if (KIND == K_MIN_F64) { R4(OP64(I_MIN_F64)) }
        if (KIND == K_CMP_U64) { R4(OP64(I_CMP_U64)) }
        if (KIND == K_CVT_F64_I32) { R4(OPM(I_CVT_F64_I32)) }
        if (KIND == K_CVT_I32_F64) { R4(OPM(I_CVT_I32_F64)) }
        if (KIND == K_MOV) { R4(OP32(I_MOV)) }
        if (KIND == K_MOV_DPP) { R4(OP32(I_MOV_DPP)) }
        if (KIND == K_FLOOR_F32) { R4(OP32(I_FLOOR_F32)) }
        if (KIND == K_CVT_F32_U8) { R4(OP32(I_CVT_F32_U8)) }
        if (KIND == K_CMP_SGPR) { R4(OP32SW(I_CMP_SGPR)) }
        if (KIND == K_CMP_4CND) { R4(OP32(I_CMP_4CND)) }
        if (KIND == K_CMP_4CND64) { R4(OP32(I_CMP_4CND64)) }
        if (KIND == K_MIX) { OP32(I_ADD_F32) OP32(I_ADD_F32) OP32(I_ADD_F32) OP64(I_ADD_F64) }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long q1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i] + (float)d[i];
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kWordsPerWave;
        o[0] = (unsigned)(t1 - t0) + (s == 12345.678f) + (smask == 12345ull);
        o[1] = (unsigned)(q1 - q0);
        o[2] = (unsigned)q0;
        o[3] = (unsigned)q1;
        o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID
        o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
    }
}

template <int KIND>
void run(unsigned *d_out, int max_waves)
{
    const int iters = 10000;
    for (int wps : {1, 4, 8}) {
        // 256-thread blocks: one wave per SIMD of a CU per block; wps blocks per CU
        const int threads = 256, blocks = 256 * wps;
        const int nw = blocks * threads / 64;
        if (nw > max_waves) continue;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 400, 1.0f);   // warm the clock
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h((size_t)nw * kWordsPerWave);
        (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        // clock: median over waves of memtime ticks per realtime tick (100 MHz)
        std::vector<double> clk(nw), cyc(nw);
        unsigned qmin = ~0u, qmax = 0;
        for (int w = 0; w < nw; w++) {
            const unsigned *o = &h[(size_t)w * kWordsPerWave];
            clk[w] = (double)o[0] / ((double)o[1] * 10.0);   // GHz
            cyc[w] = (double)o[0];
            qmin = std::min(qmin, o[2]);
            qmax = std::max(qmax, o[3]);
        }
        std::sort(clk.begin(), clk.end());
        std::sort(cyc.begin(), cyc.end());
        const double ghz = clk[nw / 2];
        const unsigned qmid = qmin + (qmax - qmin) / 2;
        // census: SIMD key = XCC | SE | SH | CU | SIMD; waves resident at mid-launch
        std::map<unsigned, int> per_simd;
        for (int w = 0; w < nw; w++) {
            const unsigned *o = &h[(size_t)w * kWordsPerWave];
            const unsigned hw = o[4];
            const unsigned key = ((hw >> 4) & 3u) | (((hw >> 8) & 0xffu) << 2) | ((o[5] & 0xfu) << 10);
            if ((int)(o[2] - qmid) <= 0 && (int)(o[3] - qmid) >= 0) per_simd[key]++;
            else per_simd[key] += 0;
        }
        std::vector<int> occ;
        for (auto &kv : per_simd) occ.push_back(kv.second);
        std::sort(occ.begin(), occ.end());
        const double n_instr = (double)nw * iters * 32.0 * (KIND == K_CMP_CND ? 2.0 : (KIND == K_CMP_4CND || KIND == K_CMP_4CND64) ? 5.0 : 1.0);
        const double wall_s = ms * 1e-3;
        const double simds = (double)per_simd.size();
        printf("%-28s w%d  wall %8.1f us  %7.2f Ginstr/s  clock %.2f GHz  cyc/SIMD %5.2f  per-wave cadence %5.2f cyc  %6.1f TFLOP/s  "
               "census: %zu SIMDs, waves/SIMD %d / %d / %d\n",
               kKinds[KIND].name, wps, ms * 1e3, n_instr / wall_s * 1e-9, ghz, simds * ghz * 1e9 * wall_s / n_instr,
               cyc[nw / 2] / (iters * 32.0), 64.0 * kKinds[KIND].flops * n_instr / wall_s * 1e-12, per_simd.size(), occ.front(),
               occ[occ.size() / 2], occ.back());
        fflush(stdout);
    }
}

template <int K0>
void run_all(unsigned *d_out, int max_waves)
{
    if constexpr (K0 < K_COUNT) {
        run<K0>(d_out, max_waves);
        run_all<K0 + 1>(d_out, max_waves);
    }
}

int main()
{
    const int max_waves = 256 * 8 * 4;
    unsigned *d_out;
    (void)hipMalloc(&d_out, (size_t)max_waves * kWordsPerWave * sizeof(unsigned));
    run_all<0>(d_out, max_waves);
    return 0;
}
