// Do the vector-ALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?  And how many VALU instructions
// hide between the MFMAs of the SAME wave, with one and with two waves per SIMD?  (attention's softmax next to its QK^T / PV MFMAs)
//   hipcc -O3 --offload-arch=gfx950 -w tools/lab/coexec_lab.hip -o /tmp/coexec_lab && /tmp/coexec_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2000;

// ROLE per wave: waves [0, n_mfma) run MFMA-only loops, waves [n_mfma, n_mfma + n_valu) VALU-only loops (8 instructions per
// iteration: NEXP v_exp_f32 + the rest v_fma_f32), the others exit.  8 MFMAs 32x32x16 per iteration.
template <int NEXP>
__global__ __launch_bounds__(512) void roles(int n_mfma, int n_valu, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.001f * (lane + e);
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < n_mfma) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
        }
    } else if (wave < n_mfma + n_valu) {
        for (int it = 0; it < ITERS * 4; ++it) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < NEXP) v[e] = __builtin_amdgcn_exp2f(v[e]);
                else v[e] = __builtin_fmaf(v[e], 0.999f, 0.001f);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
// One instruction stream: per MFMA 32x32x16, NV vector instructions (NE of them v_exp_f32) behind it; NW waves per workgroup
template <int NV, int NE, int NW>
__global__ __launch_bounds__(NW * 64) void inter(float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.001f * (lane + e);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (e < NE) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(0.999f), "v"(0.001f));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
// The candidate attention stream: per MFMA 32x32x16 two v_exp_f32, one v_max3_f32, one v_cvt_pk_f16_f32 (24 issue cycles), an LDS
// fragment read every other gap (RD), a workgroup barrier every 14 MFMAs (BAR); NW waves per workgroup
template <int NW, bool RD, bool BAR>
__global__ __launch_bounds__(NW * 64) void cand(float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += NW * 64) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.001f * (lane + e);
    unsigned pk = 0;
    const char* rp = smem + lane * 144;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS * 8 / 14; ++it) {
        if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m & 1 ? b : a, b, acc[m & 3], 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[1]));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[2]) : "v"(v[3]), "v"(v[4]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[5]), "v"(v[6]));
            if (RD && (m & 1)) a = *(const h8*)(rp + (m >> 1) * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
        v[7] += __builtin_bit_cast(float, pk);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = (t1 - t0) * 14 / 14;
}
float* g_out; long long* g_cyc;
template <typename F> void timeit(const char* name, F launch, int nw_report) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%-64s", name);
    for (int w = 0; w < nw_report; ++w) printf(" w%d %6.1f", w, (double)h[w] / (ITERS * 8));
    printf("   (cycles per MFMA-slot: 8 MFMAs, or 32 VALU, per iteration)\n");
}
int main() {
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 256 * 8 * 8);
    printf("roles: cycles per iteration-eighth of wave 0 (MFMA role) and wave 4 (VALU role: 4 VALU per eighth)\n");
    timeit("4 MFMA waves alone (1 per SIMD)", [] { hipLaunchKernelGGL(roles<0>, dim3(256), dim3(512), 0, 0, 4, 0, g_out, g_cyc); }, 1);
    timeit("8 MFMA waves (2 per SIMD)", [] { hipLaunchKernelGGL(roles<0>, dim3(256), dim3(512), 0, 0, 8, 0, g_out, g_cyc); }, 5);
    timeit("4 VALU waves alone (fma only), waves 0-3", [] { hipLaunchKernelGGL(roles<0>, dim3(256), dim3(512), 0, 0, 0, 4, g_out, g_cyc); }, 1);
    timeit("4 VALU waves alone (4 of 8 exp)", [] { hipLaunchKernelGGL(roles<4>, dim3(256), dim3(512), 0, 0, 0, 4, g_out, g_cyc); }, 1);
    timeit("4 MFMA waves + 4 VALU waves (fma only)", [] { hipLaunchKernelGGL(roles<0>, dim3(256), dim3(512), 0, 0, 4, 4, g_out, g_cyc); }, 5);
    timeit("4 MFMA waves + 4 VALU waves (4 of 8 exp)", [] { hipLaunchKernelGGL(roles<4>, dim3(256), dim3(512), 0, 0, 4, 4, g_out, g_cyc); }, 5);
    printf("interleaved in ONE stream: cycles per MFMA\n");
    timeit("1 wave/SIMD: MFMA + 0 VALU", [] { hipLaunchKernelGGL((inter<0, 0, 4>), dim3(256), dim3(256), 0, 0, g_out, g_cyc); }, 1);
    timeit("1 wave/SIMD: MFMA + 4 fma", [] { hipLaunchKernelGGL((inter<4, 0, 4>), dim3(256), dim3(256), 0, 0, g_out, g_cyc); }, 1);
    timeit("1 wave/SIMD: MFMA + 5 (2 exp + 3 fma)  [attention's mix]", [] { hipLaunchKernelGGL((inter<5, 2, 4>), dim3(256), dim3(256), 0, 0, g_out, g_cyc); }, 1);
    timeit("1 wave/SIMD: MFMA + 6 (3 exp + 3 fma)", [] { hipLaunchKernelGGL((inter<6, 3, 4>), dim3(256), dim3(256), 0, 0, g_out, g_cyc); }, 1);
    timeit("2 waves/SIMD: MFMA + 0 VALU", [] { hipLaunchKernelGGL((inter<0, 0, 8>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    timeit("2 waves/SIMD: MFMA + 4 fma", [] { hipLaunchKernelGGL((inter<4, 0, 8>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    timeit("2 waves/SIMD: MFMA + 5 (2 exp + 3 fma)  [attention's mix]", [] { hipLaunchKernelGGL((inter<5, 2, 8>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    timeit("2 waves/SIMD: MFMA + 6 (3 exp + 3 fma)", [] { hipLaunchKernelGGL((inter<6, 3, 8>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    timeit("2 waves/SIMD: MFMA + 8 (4 exp + 4 fma)", [] { hipLaunchKernelGGL((inter<8, 4, 8>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    printf("candidate attention stream (2 exp + max3 + cvt_pk per MFMA): cycles per MFMA\n");
    timeit("1 wave/SIMD", [] { hipLaunchKernelGGL((cand<4, false, false>), dim3(256), dim3(256), 0, 0, g_out, g_cyc); }, 1);
    timeit("1 wave/SIMD + LDS reads", [] { hipLaunchKernelGGL((cand<4, true, false>), dim3(256), dim3(256), 0, 0, g_out, g_cyc); }, 1);
    timeit("2 waves/SIMD", [] { hipLaunchKernelGGL((cand<8, false, false>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    timeit("2 waves/SIMD + LDS reads", [] { hipLaunchKernelGGL((cand<8, true, false>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    timeit("2 waves/SIMD + LDS reads + barrier / 14 MFMAs", [] { hipLaunchKernelGGL((cand<8, true, true>), dim3(256), dim3(512), 0, 0, g_out, g_cyc); }, 5);
    return 0;
}
