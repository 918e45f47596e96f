// Issue cost of v_exp_f32 / v_exp_f16 / v_cvt_pk_f16_f32 / v_max3_f32 on gfx950: cycles per instruction of ONE wave on a SIMD
// (s_memtime around 4096 independent instructions), round 4: is a half-precision exponential cheaper for the softmax?
// Build: hipcc --offload-arch=gfx950 -O2 -o exp_lab.bin exp_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k(unsigned long long* out, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    unsigned long long t0, t1, t2, t3, t4;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < 32; ++i) {
        asm volatile(REP16("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    for (int i = 0; i < 32; ++i) {
        asm volatile(REP16("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\tv_exp_f16 %4, %4\n\tv_exp_f16 %5, %5\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
    for (int i = 0; i < 32; ++i) {
        asm volatile(REP16("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_cvt_pk_f16_f32 %1, %2, %3\n\tv_cvt_pk_f16_f32 %2, %3, %4\n\tv_cvt_pk_f16_f32 %3, %4, %5\n\tv_cvt_pk_f16_f32 %4, %5, %6\n\tv_cvt_pk_f16_f32 %5, %6, %7\n\tv_cvt_pk_f16_f32 %6, %7, %0\n\tv_cvt_pk_f16_f32 %7, %0, %1\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3));
    for (int i = 0; i < 32; ++i) {
        asm volatile(REP16("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %1, %2, %3, %4\n\tv_max3_f32 %2, %3, %4, %5\n\tv_max3_f32 %3, %4, %5, %6\n\tv_max3_f32 %4, %5, %6, %7\n\tv_max3_f32 %5, %6, %7, %0\n\tv_max3_f32 %6, %7, %0, %1\n\tv_max3_f32 %7, %0, %1, %2\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t4));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[4] = 1;
}
int main() {
    unsigned long long* d; unsigned long long h[5] = {0};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0.5f);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const double n = 32.0 * 16 * 8;
    printf("one wave, independent instructions, cycles each: v_exp_f32 %.2f  v_exp_f16 %.2f  v_cvt_pk_f16_f32 %.2f  v_max3_f32 %.2f\n",
           h[0] / n, h[1] / n, h[2] / n, h[3] / n);
    return 0;
}
