// Third issue-rate probe (gfx950): do the cheap int32 opcodes (v_add/v_sub/v_and/v_or/v_xor/right shifts: ~2.4-2.9 cycles
// alone) overlap with the 4.2-cycle opcodes (max, cmp, addc, bfe, VOP3, DPP) when the two kinds are mixed in one
// instruction stream, or do their costs add up?  Streams of 8 instructions on independent registers, F of them v_max_i32
// ("full") and 8 - F v_add_u32 / v_and_b32 ("simple"), interleaved.  Reported: cycles of SIMD time per 8 instructions.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench3.hip -o /tmp/vm3 && /tmp/vm3
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(x) x x x x x x x x
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
#define F(n, m) "v_max_i32 %" #n ", %" #n ", %" #m "\n"
#define S(n, m) "v_add_u32 %" #n ", %" #n ", %" #m "\n"
#define A(n, m) "v_and_b32 %" #n ", %" #n ", %" #m "\n"
#define X(n, m) "v_sub_u32 %" #n ", %" #n ", %" #m "\n"

template <int KIND>
__global__ __launch_bounds__(64) void k(int* out, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    int c0 = a0 ^ 5, c1 = a1 ^ 6, c2 = a2 ^ 7, c3 = a3 ^ 8, c4 = a4 ^ 9, c5 = a5 ^ 10, c6 = a6 ^ 11, c7 = a7 ^ 12;
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { REP8(asm volatile(F(0, 8) F(1, 9) F(2, 10) F(3, 11) F(4, 12) F(5, 13) F(6, 14) F(7, 15) REGS);) }       // 8 full
        if (KIND == 1) { REP8(asm volatile(F(0, 8) S(1, 9) F(2, 10) S(3, 11) F(4, 12) S(5, 13) F(6, 14) S(7, 15) REGS);) }       // 4 full + 4 add
        if (KIND == 2) { REP8(asm volatile(F(0, 8) S(1, 9) S(2, 10) S(3, 11) F(4, 12) S(5, 13) S(6, 14) S(7, 15) REGS);) }       // 2 full + 6 add
        if (KIND == 3) { REP8(asm volatile(F(0, 8) F(1, 9) F(2, 10) S(3, 11) F(4, 12) F(5, 13) F(6, 14) S(7, 15) REGS);) }       // 6 full + 2 add
        if (KIND == 4) { REP8(asm volatile(S(0, 8) S(1, 9) S(2, 10) S(3, 11) S(4, 12) S(5, 13) S(6, 14) S(7, 15) REGS);) }       // 8 add
        if (KIND == 5) { REP8(asm volatile(F(0, 8) A(1, 9) F(2, 10) X(3, 11) F(4, 12) A(5, 13) F(6, 14) X(7, 15) REGS);) }       // 4 full + 4 and/sub
        if (KIND == 6) { REP8(asm volatile(F(0, 8) F(1, 9) F(2, 10) F(3, 11) S(4, 12) S(5, 13) S(6, 14) S(7, 15) REGS);) }       // 4 full then 4 add (blocked)
        if (KIND == 7) { REP8(asm volatile(F(0, 8) S(0, 9) F(2, 10) S(2, 11) F(4, 12) S(4, 13) F(6, 14) S(6, 15) REGS);) }       // 4 x (full, dependent add)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
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
    return ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * ITER * 8.0);      // cycles per 8-instruction stream
}

int main() {
    int* d_out; (void)hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(int));
    const char* names[] = {"8 max", "4 max + 4 add, interleaved", "2 max + 6 add", "6 max + 2 add", "8 add", "4 max + 2 and + 2 sub", "4 max, then 4 add", "4 x (max, dependent add)"};
    printf("cycles (at 2.4 GHz) of SIMD time per stream of 8 wave64 instructions\n%-30s %8s %8s %8s %8s\n", "stream", "1 wave", "2 waves", "3 waves", "8 waves");
#define ROW(K) printf("%-30s %8.1f %8.1f %8.1f %8.1f\n", names[K], run<K>(1, d_out), run<K>(2, d_out), run<K>(3, d_out), run<K>(8, d_out));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
    return 0;
}
