// Microbenchmark: do v_dot4_u32_u8 / v_dot2_u32_u16 / v_pk_* share the issue slot of plain vector ALU instructions on gfx950?
// Each kernel runs ITER iterations of an unrolled block of independent instructions on 8 accumulators per lane, 256 threads x
// many workgroups; prints ns per wave-instruction for blocks of: 64 adds, 64 dot4, 32 adds + 32 dot4 (interleaved), 64 pk_mad,
// 32 adds + 32 pk_mad, 64 dot2.  If a mix runs as fast as its halves alone, the two kinds issue on different pipes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define DOT2(i) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define PKMAD(i) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            if (MODE == 0) { REP8(ADD) }
            if (MODE == 1) { REP8(DOT4) }
            if (MODE == 2) { ADD(0) DOT4(1) ADD(2) DOT4(3) ADD(4) DOT4(5) ADD(6) DOT4(7) }
            if (MODE == 3) { REP8(PKMAD) }
            if (MODE == 4) { ADD(0) PKMAD(1) ADD(2) PKMAD(3) ADD(4) PKMAD(5) ADD(6) PKMAD(7) }
            if (MODE == 5) { REP8(DOT2) }
            if (MODE == 6) { ADD(0) DOT2(1) ADD(2) DOT2(3) ADD(4) DOT2(5) ADD(6) DOT2(7) }
            if (MODE == 7) { REP8(PERM) }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
template <int MODE> static double run(unsigned* d, int wgs, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096);
    const int wgs = 256 * 8, iters = 20000;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    const char* names[] = {"64 v_add_u32", "64 v_dot4_u32_u8", "32 add + 32 dot4", "64 v_pk_mad_u16", "32 add + 32 pk_mad", "64 v_dot2_u32_u16", "32 add + 32 dot2", "64 v_perm_b32"};
    double ms[8] = {run<0>(d, wgs, iters), run<1>(d, wgs, iters), run<2>(d, wgs, iters), run<3>(d, wgs, iters), run<4>(d, wgs, iters), run<5>(d, wgs, iters), run<6>(d, wgs, iters), run<7>(d, wgs, iters)};
    for (int m = 0; m < 8; m++) {
        // wave-instructions per SIMD: wgs * 4 waves * iters * 64 / (256 CUs * 4 SIMDs)
        const double per_simd = (double)wgs * 4 * iters * 64 / 1024.0;
        printf("%-22s %8.3f ms  = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
