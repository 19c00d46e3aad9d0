// Issue-rate microbenchmark for the int32 VALU instruction mix of the DP kernel (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o gpurun_out/valu_microbench ; run on the GPU box.
// Each kernel runs ITER iterations of 64 instructions of one kind on 8 independent registers per lane; the host reports
// cycles per wave-instruction per SIMD for 1, 2, 4 waves per SIMD (256 CUs x 4 SIMDs, grid sized accordingly).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(64) void k(int* out, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b0 = a0;
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n v_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n v_add_u32 %5, %5, %4\n v_add_u32 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 1) { REP8(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %3\n v_max_i32 %4, %4, %5\n v_max_i32 %6, %6, %7\n v_max_i32 %1, %1, %0\n v_max_i32 %3, %3, %2\n v_max_i32 %5, %5, %4\n v_max_i32 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 2) { REP8(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %2, %2, %3, %4\n v_max3_i32 %4, %4, %5, %6\n v_max3_i32 %6, %6, %7, %0\n v_max3_i32 %1, %1, %0, %3\n v_max3_i32 %3, %3, %2, %5\n v_max3_i32 %5, %5, %4, %7\n v_max3_i32 %7, %7, %6, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 3) { REP8(asm volatile("v_cmp_gt_i32 s[20:21], %0, %1\n v_cmp_gt_i32 s[22:23], %2, %3\n v_cmp_gt_i32 s[24:25], %4, %5\n v_cmp_gt_i32 vcc, %6, %7\n v_addc_co_u32 %8, s[20:21], %8, %8, s[20:21]\n v_addc_co_u32 %8, s[22:23], %8, %8, s[22:23]\n v_addc_co_u32 %8, s[24:25], %8, %8, s[24:25]\n v_addc_co_u32 %8, vcc, %8, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0) :: "vcc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (KIND == 4) { REP8(asm volatile("v_bfe_i32 %0, %1, %2, 4\n v_bfe_i32 %3, %4, %2, 4\n v_bfe_i32 %5, %6, %2, 4\n v_bfe_i32 %7, %0, %2, 4\n v_bfe_i32 %1, %3, %2, 4\n v_bfe_i32 %4, %5, %2, 4\n v_bfe_i32 %6, %7, %2, 4\n v_bfe_i32 %0, %1, %2, 4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 5) { REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 6) { REP8(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %2, %2, %3\n v_pk_add_i16 %4, %4, %5\n v_pk_max_i16 %6, %6, %7\n v_pk_add_i16 %1, %1, %0\n v_pk_max_i16 %3, %3, %2\n v_pk_add_i16 %5, %5, %4\n v_pk_max_i16 %7, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 7) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %1, %1, %0, %3\n v_add3_u32 %3, %3, %2, %5\n v_add3_u32 %5, %5, %4, %7\n v_add3_u32 %7, %7, %6, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)b0;
}

template <int KIND>
double run(int waves_per_simd, int* d_out) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double inst_per_simd = (double)waves_per_simd * ITER * 64.0;
    return ms * 1e-3 * 2.4e9 / inst_per_simd;          // cycles @2.4 GHz per wave-instruction per SIMD
}

int main() {
    int* d_out; hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(int));
    const char* names[] = {"v_add_u32", "v_max_i32", "v_max3_i32", "4x v_cmp + 4x v_addc", "v_bfe_i32", "v_mov_b32_dpp wave_shr:1", "v_pk_add/max_i16", "v_add3_u32"};
    printf("cycles (at 2.4 GHz) per wave64 instruction per SIMD, by waves per SIMD\n%-28s %8s %8s %8s %8s\n", "instruction", "1", "2", "4", "8");
#define ROW(K) printf("%-28s %8.2f %8.2f %8.2f %8.2f\n", names[K], run<K>(1, d_out), run<K>(2, d_out), run<K>(4, d_out), run<K>(8, d_out));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
    return 0;
}
