// Feasibility probe for DESIGN.md 7(1)(i): can the DP's four compare masks per cell (SGPR pairs written by v_cmp) go to
// memory with scalar stores instead of being re-packed per lane with four v_addc?  gfx950 assembles s_store_dwordx2; this
// measures what it costs next to the v_cmp + v_addc form and checks that the stored masks are what the lanes computed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/sstore_microbench.hip -o gpurun_out/sstore_microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 2048

// KIND 2: 4 x v_cmp into s[20:27] + 2 x s_store_dwordx4 (immediate offset on the second);
// KIND 0: 4 x v_cmp + 4 x v_addc (what c2_push4 does);  KIND 1: 4 x v_cmp + 4 x s_store_dwordx2 (+ one s_add for the offset)
template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* masks, int* out, int seed) {
    int a0 = threadIdx.x * 7 + seed, a1 = threadIdx.x * 5 + 3, a2 = threadIdx.x ^ 21, a3 = 40 - (int)threadIdx.x;
    unsigned bits = 0;
    unsigned long long* mine = masks + (size_t)blockIdx.x * ITER * 4;
    unsigned off = 0;
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {
            asm volatile("v_cmp_gt_i32 s[20:21], %1, %2\n v_cmp_gt_i32 s[22:23], %2, %3\n v_cmp_gt_i32 s[24:25], %3, %4\n v_cmp_gt_i32 vcc, %4, %1\n"
                         "v_addc_co_u32 %0, s[20:21], %0, %0, s[20:21]\n v_addc_co_u32 %0, s[22:23], %0, %0, s[22:23]\n"
                         "v_addc_co_u32 %0, s[24:25], %0, %0, s[24:25]\n v_addc_co_u32 %0, vcc, %0, %0, vcc"
                         : "+v"(bits) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
        } else if (KIND == 2) {
            asm volatile("v_cmp_gt_i32 s[20:21], %2, %3\n v_cmp_gt_i32 s[22:23], %3, %4\n v_cmp_gt_i32 s[24:25], %4, %5\n v_cmp_gt_i32 s[26:27], %5, %2\n"
                         "s_nop 1\n"
                         "s_store_dwordx4 s[20:23], %1, %0\n s_store_dwordx4 s[24:27], %1, %0 offset:16\n s_add_u32 %0, %0, 32"
                         : "+s"(off) : "s"(mine), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory");
        } else {
            asm volatile("v_cmp_gt_i32 s[20:21], %2, %3\n v_cmp_gt_i32 s[22:23], %3, %4\n v_cmp_gt_i32 s[24:25], %4, %5\n v_cmp_gt_i32 s[26:27], %5, %2\n"
                         "s_nop 1\n"
                         "s_store_dwordx2 s[20:21], %1, %0\n s_add_u32 %0, %0, 8\n s_store_dwordx2 s[22:23], %1, %0\n s_add_u32 %0, %0, 8\n"
                         "s_store_dwordx2 s[24:25], %1, %0\n s_add_u32 %0, %0, 8\n s_store_dwordx2 s[26:27], %1, %0\n s_add_u32 %0, %0, 8"
                         : "+s"(off) : "s"(mine), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory");
        }
        a0 += 3; a1 -= 1; a2 += 2; a3 += 1;                      // (4 more VALU per iteration in both kinds)
    }
    if (KIND >= 1) asm volatile("s_waitcnt lgkmcnt(0)\n s_dcache_wb" ::: "memory");
    out[blockIdx.x * 64 + threadIdx.x] = (int)bits + a0 + a1 + a2 + a3;
}

template <int KIND>
double run(int waves_per_simd, unsigned long long* d_masks, int* d_out) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d_masks, d_out, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d_masks, d_out, 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * ITER);   // cycles @2.4 GHz per iteration ("cell") per SIMD
}

int main() {
    const size_t nwaves = 256 * 4 * 8;
    unsigned long long* d_masks; int* d_out;
    hipMalloc(&d_masks, nwaves * ITER * 4 * sizeof(unsigned long long));
    hipMalloc(&d_out, nwaves * 64 * sizeof(int));
    hipMemset(d_masks, 0, nwaves * ITER * 4 * sizeof(unsigned long long));
    printf("cycles (at 2.4 GHz) per cell (4 compares + 4 VALU of loop work + packing or storing) per SIMD, by waves per SIMD\n%-34s %8s %8s %8s %8s\n", "form", "1", "2", "4", "8");
    printf("%-34s %8.1f %8.1f %8.1f %8.1f\n", "4 v_cmp + 4 v_addc", run<0>(1, d_masks, d_out), run<0>(2, d_masks, d_out), run<0>(4, d_masks, d_out), run<0>(8, d_masks, d_out));
    printf("%-34s %8.1f %8.1f %8.1f %8.1f\n", "4 v_cmp + 4 s_store_dwordx2", run<1>(1, d_masks, d_out), run<1>(2, d_masks, d_out), run<1>(4, d_masks, d_out), run<1>(8, d_masks, d_out));
    printf("%-34s %8.1f %8.1f %8.1f %8.1f\n", "4 v_cmp + 2 s_store_dwordx4", run<2>(1, d_masks, d_out), run<2>(2, d_masks, d_out), run<2>(4, d_masks, d_out), run<2>(8, d_masks, d_out));
    // correctness of the stored masks of wave 0 (seed 1): recompute on the host
    std::vector<unsigned long long> h((size_t)ITER * 4);
    hipMemcpy(h.data(), d_masks, h.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < ITER; ++i) {
        unsigned long long m[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) {
            const int a0 = l * 7 + 1 + 3 * i, a1 = l * 5 + 3 - i, a2 = (l ^ 21) + 2 * i, a3 = 40 - l + i;
            if (a0 > a1) m[0] |= 1ull << l;
            if (a1 > a2) m[1] |= 1ull << l;
            if (a2 > a3) m[2] |= 1ull << l;
            if (a3 > a0) m[3] |= 1ull << l;
        }
        for (int q = 0; q < 4; ++q) bad += (h[(size_t)i * 4 + q] != m[q]);
    }
    printf("stored masks of wave 0: %zu of %d differ from the host's\n", bad, ITER * 4);
    return 0;
}
