// Round 3 issue-rate probe with control rows (gfx950): tools/valu_microbench2.hip plus v_fma_f32 / v_pk_fma_f32 (the guide's reference
// points), v_perm_b32, v_pk_sub_i16, the row DPP forms the packed kernel uses, and one launch per (kind, occupancy) so that
// `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` gives every row its TRUE cycle count (tools/valu_clock_summary.py): the event-timed
// column assumes the nominal 2.4 GHz.
// (based on the) second issue-rate probe (gfx950): which int32 VALU opcodes run at the rate of v_add_u32 (~2.8 cycles of SIMD time per
// wave64 instruction) and which at ~4.2, and what a DPP source costs on top -- the DP cell's cost model.
// Same scheme as tools/valu_microbench.hip: ITER x 64 instructions of one kind on 8 registers per lane, 3 and 8 waves per SIMD.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench2.hip -o /tmp/vm2 && /tmp/vm2
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(x) x x x x x x x x
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0) :: "vcc", "s20", "s21", "s22", "s23"
#define BIN(op) op " %0, %0, %1\n" op " %2, %2, %3\n" op " %4, %4, %5\n" op " %6, %6, %7\n" op " %1, %1, %0\n" op " %3, %3, %2\n" op " %5, %5, %4\n" op " %7, %7, %6"
#define BINX(op, x) op " %0, %0, %1 " x "\n" op " %2, %2, %3 " x "\n" op " %4, %4, %5 " x "\n" op " %6, %6, %7 " x "\n" op " %1, %1, %0 " x "\n" op " %3, %3, %2 " x "\n" op " %5, %5, %4 " x "\n" op " %7, %7, %6 " x
#define TER(op) op " %0, %0, %1, %2\n" op " %2, %2, %3, %4\n" op " %4, %4, %5, %6\n" op " %6, %6, %7, %0\n" op " %1, %1, %0, %3\n" op " %3, %3, %2, %5\n" op " %5, %5, %4, %7\n" op " %7, %7, %6, %1"
#define TERI(op, imm) op " %0, %0, %1, " imm "\n" op " %2, %2, %3, " imm "\n" op " %4, %4, %5, " imm "\n" op " %6, %6, %7, " imm "\n" op " %1, %1, %0, " imm "\n" op " %3, %3, %2, " imm "\n" op " %5, %5, %4, " imm "\n" op " %7, %7, %6, " imm
#define DPPS "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"

template <int KIND>
__global__ __launch_bounds__(64) void k(int* out, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b0 = a0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {1.0f + a0, 0.5f}, p1 = {0.25f, 1.5f}, p2 = {0.75f, 0.125f}, p3 = {1.25f, 0.0625f};
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { REP8(asm volatile(BIN("v_add_u32") REGS);) }
        if (KIND == 1) { REP8(asm volatile(BIN("v_sub_u32") REGS);) }
        if (KIND == 2) { REP8(asm volatile(BIN("v_and_b32") REGS);) }
        if (KIND == 3) { REP8(asm volatile(BIN("v_xor_b32") REGS);) }
        if (KIND == 4) { REP8(asm volatile(BIN("v_lshlrev_b32") REGS);) }
        if (KIND == 5) { REP8(asm volatile(BIN("v_ashrrev_i32") REGS);) }
        if (KIND == 6) { REP8(asm volatile(BIN("v_max_i32") REGS);) }
        if (KIND == 7) { REP8(asm volatile(BIN("v_min_u32") REGS);) }
        if (KIND == 8) { REP8(asm volatile(TERI("v_alignbit_b32", "31") REGS);) }
        if (KIND == 9) { REP8(asm volatile(TER("v_lshl_add_u32") REGS);) }
        if (KIND == 10) { REP8(asm volatile(TER("v_lshl_or_b32") REGS);) }
        if (KIND == 11) { REP8(asm volatile(TER("v_and_or_b32") REGS);) }
        if (KIND == 12) { REP8(asm volatile(TER("v_bfi_b32") REGS);) }
        if (KIND == 13) { REP8(asm volatile(BINX("v_add_u32_dpp", DPPS) REGS);) }
        if (KIND == 14) { REP8(asm volatile(BINX("v_max_i32_dpp", DPPS) REGS);) }
        if (KIND == 15) { REP8(asm volatile("v_mov_b32_dpp %0, %1 " DPPS "\n v_add_u32 %2, %2, %0\n v_mov_b32_dpp %4, %5 " DPPS "\n v_add_u32 %6, %6, %4\n v_mov_b32_dpp %1, %0 " DPPS "\n v_add_u32 %3, %3, %1\n v_mov_b32_dpp %5, %4 " DPPS "\n v_add_u32 %7, %7, %5" REGS);) }
        if (KIND == 16) { REP8(asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cmp_gt_i32 s[20:21], %2, %3\n v_cmp_gt_i32 vcc, %4, %5\n v_cmp_gt_i32 s[22:23], %6, %7\n v_cmp_gt_i32 vcc, %1, %0\n v_cmp_gt_i32 s[20:21], %3, %2\n v_cmp_gt_i32 vcc, %5, %4\n v_cmp_gt_i32 s[22:23], %7, %6" REGS);) }
        if (KIND == 17) { REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_addc_co_u32 %7, vcc, %7, %7, vcc" REGS);) }
        if (KIND == 18) { REP8(asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n v_sub_co_u32 %2, vcc, %2, %3\n v_sub_co_u32 %4, vcc, %4, %5\n v_sub_co_u32 %6, vcc, %6, %7\n v_sub_co_u32 %1, vcc, %1, %0\n v_sub_co_u32 %3, vcc, %3, %2\n v_sub_co_u32 %5, vcc, %5, %4\n v_sub_co_u32 %7, vcc, %7, %6" REGS);) }
        if (KIND == 19) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %7, %7, %6, vcc" REGS);) }
        if (KIND == 20) { REP8(asm volatile(TERI("v_bfe_i32", "4") REGS);) }
        if (KIND == 21) { REP8(asm volatile(TER("v_max3_i32") REGS);) }
        if (KIND == 22) { REP8(asm volatile(TER("v_add3_u32") REGS);) }
        if (KIND == 23) { REP8(asm volatile(TER("v_mad_i32_i24") REGS);) }
        if (KIND == 24) { REP8(asm volatile(BIN("v_pk_add_i16") REGS);) }
        if (KIND == 25) { REP8(asm volatile(BIN("v_pk_max_i16") REGS);) }
        if (KIND == 26) { REP8(asm volatile(BIN("v_or_b32") REGS);) }
        if (KIND == 27) { REP8(asm volatile(BIN("v_lshrrev_b32") REGS);) }
        if (KIND == 28) { REP8(asm volatile(TER("v_fma_f32") REGS);) }
        if (KIND == 29) { REP8(asm volatile(TER("v_perm_b32") REGS);) }
        if (KIND == 30) { REP8(asm volatile(BIN("v_pk_sub_i16") REGS);) }
        if (KIND == 31) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" REGS);) }
        if (KIND == 32) { REP8(asm volatile(BIN("v_pk_add_u16") REGS);) }
        if (KIND == 33) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n v_mov_b32 %1, %0\n v_mov_b32 %3, %2\n v_mov_b32 %5, %4\n v_mov_b32 %7, %6" REGS);) }
        if (KIND == 34) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %3, %0\n v_pk_fma_f32 %2, %2, %0, %1\n v_pk_fma_f32 %3, %3, %1, %2" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)b0 + (int)(p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y);
}

template <int KIND>
double run(int waves_per_simd, int* d_out) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 2);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * ITER * 64.0);
}

int main() {
    int* d_out; (void)hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(int));
    const char* names[] = {"v_add_u32", "v_sub_u32", "v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_max_i32", "v_min_u32",
                           "v_alignbit_b32 (imm 31)", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfi_b32", "v_add_u32_dpp", "v_max_i32_dpp",
                           "v_mov_dpp + v_add_u32 (pair/2)", "v_cmp_gt_i32 -> sgpr/vcc", "v_addc_co_u32 (vcc in/out)", "v_sub_co_u32", "v_cndmask_b32",
                           "v_bfe_i32", "v_max3_i32", "v_add3_u32", "v_mad_i32_i24", "v_pk_add_i16", "v_pk_max_i16", "v_or_b32", "v_lshrrev_b32",
                           "v_fma_f32 (control)", "v_perm_b32", "v_pk_sub_i16", "v_mov_b32_dpp row_shr/row_shl", "v_pk_add_u16", "v_mov_b32", "v_pk_fma_f32 (control)"};
    printf("cycles (at 2.4 GHz) per wave64 instruction per SIMD\n%-32s %8s %8s\n", "instruction", "3 waves", "8 waves");
#define ROW(K) printf("%-32s %8.2f %8.2f   kind %d\n", names[K], run<K>(3, d_out), run<K>(8, d_out), K);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19)
    ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31) ROW(32) ROW(33) ROW(34)
    return 0;
}
