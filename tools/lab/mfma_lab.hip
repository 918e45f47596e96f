// Where do the cycles (and the clock) of an LDS-read + MFMA K-step go?  Bare variants of the igemm2 inner loop.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -w tools/lab/mfma_lab.hip -o /tmp/mfma_lab && /tmp/mfma_lab
// Every workgroup (8 waves, one per CU) runs STEPS K-steps of a 256 x 320 x 32 tile: 40 MFMAs 16x16x32 (or 20 of 32x32x16) per wave.
// Each variant is launched back to back for WARM seconds before it is timed (the clock the chip holds under a load settles slowly),
// printed: wall per step, TF/s, shader cycles per step per wave (s_memtime), clock = cycles / wall.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int STEPS = 180, STAGE = 36864, ABYTES = 16384;
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_wave_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// R: 0 no LDS reads, 1 reads hoisted, 2 reads interleaved.  BAR: barrier per step.  MS: 16 / 32.  DMA: 0 none, else 5 LDS-DMA pieces
// per wave and step (2 "activation" pieces from a per-workgroup region, 3 "weight" pieces from a region all workgroups share),
// interleaved when R == 2.  DMA = 1: every activation byte is new (HBM stream); 9: each 16 KB activation chunk is fetched in 9
// consecutive steps (the 3x3 taps: L2 hits); 100: activations from one 16 KB chunk per workgroup (always L2); 200: weights only
// SEG: 0 = a DMA wave-instruction fetches 1 KiB contiguous; 64 = 16 segments of 64 B, 640 B (activations) / 5760 B (weights) apart,
// as igemm2 with BK = 32 does; 128 = 8 segments of 128 B (BK = 64: whole cache lines)
template <int R, bool BAR, int MS, int DMA, int SEG = 0>
__global__ __launch_bounds__(512) void lab(const h8* src, const char* stream, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = 64 / MS, NI = 160 / MS, KS = MS == 32 ? 2 : 1, AR = MS == 32 ? 16 : 4;
    typedef float acc_t __attribute__((ext_vector_type(AR)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * STAGE / 16; i += 512) ((h8*)smem)[i] = src[(i + blockIdx.x) & 4095];
    __syncthreads();
    acc_t acc[NI][MI];
    for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < AR; ++r) acc[j][i][r] = 0.f;
    const int lrow = lane & (MS - 1), lh = lane / MS;
    auto swz_of = [](int r) { return MS == 32 ? ((r >> 2) & 3) : (((r >> 3) & 1) << 1); };
    int koff[KS];
    for (int ks = 0; ks < KS; ++ks) koff[ks] = lrow * 64 + ((((MS == 32 ? ks * 2 : 0) + lh) ^ swz_of(lrow)) << 4);
    const int wm = wave / 2, wn = wave % 2;
    constexpr int NF = (MI + NI) * KS;      // fragments per step; index f: order of consumption
    // fragment list in consumption order: for ks: W j then ... we keep it simple: X frags of a k-substep first, then W frags
    auto frag_addr = [&](int kt, int f) {
        const int ks = f / (MI + NI), r = f % (MI + NI);
        const char* base = smem + (kt & 3) * STAGE;
        return r < MI ? base + (wm * 64 + r * MS) * 64 + koff[ks] : base + ABYTES + (wn * 160 + (r - MI) * MS) * 64 + koff[ks];
    };
    h8 fa[NF], fb[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) fa[f] = *(const h8*)frag_addr(0, f);
    // segmented forms: lane -> (row, 16-byte chunk of the row's segment); a piece's rows are consecutive pixels / weight rows
    const int lpr = SEG ? SEG / 16 : 64, srow = lane / lpr, scol = (lane % lpr) * 16, rpp = 64 / lpr;   // rows per piece
    const char* my_stream = stream + (size_t)blockIdx.x * (STEPS * 16384) + (SEG ? (wave * rpp + srow) * 640 + scol : (wave * 64 + lane) * 16);
    const char* shared = stream + (size_t)256 * STEPS * 16384 + (SEG ? (wave * rpp + srow) * 5760 + scol : (wave * 64 + lane) * 16);
    auto dma_piece = [&](int kt, int p) {
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS const void*)smem + ((kt + 3) & 3) * STAGE);
        const int chunk = DMA == 1 ? kt : DMA == 9 ? kt / 9 : 0;
        if (DMA == 200 && p < 2) return;
        if (SEG) {
            // activations: 256 pixel rows of 640 B, this step's segment at column (chunk % (640 / SEG)) * SEG of them, image
            // after image; weights: 320 rows of 5760 B, segment (kt % (5760 / SEG))
            if (p < 2) glds16(my_stream + (size_t)(chunk / (640 / SEG)) * 163840 + (chunk % (640 / SEG)) * SEG + p * (8 * rpp * 640), base + (wave + p * 8) * 1024);
            else glds16(shared + (kt % (5760 / SEG)) * SEG + (size_t)(((p - 2) * 8 * rpp) % (320 - 8 * rpp + 1)) * 5760, base + ABYTES + ((wave + (p - 2) * 8) % 20) * 1024);
            return;
        }
        if (p < 2) glds16(my_stream + (size_t)chunk * 16384 + p * 8192, base + (wave + p * 8) * 1024);
        else glds16(shared + (size_t)(kt % 64) * 20480 + (p - 2) * 8192 % 20480, base + ABYTES + ((wave + (p - 2) * 8) % 20) * 1024);
    };
    auto mfma = [&](int j, int i, int ks, h8 (&fr)[NF]) {
        const h8 x = fr[ks * (MI + NI) + i], w = fr[ks * (MI + NI) + MI + j];
        if constexpr (MS == 32) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[j][i], 0, 0, 0);
        else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc[j][i], 0, 0, 0);
    };
    long long t0 = __builtin_amdgcn_s_memtime();
    auto step = [&](int kt, h8 (&cur)[NF], h8 (&nxt)[NF]) {
        if (DMA) wait_vmcnt<DMA == 200 ? 3 : DMA ? 5 : 0>();
        if (BAR) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }
        if (R == 1) {
            if (DMA) { for (int p = 0; p < 5; ++p) dma_piece(kt, p); }
#pragma unroll
            for (int f = 0; f < NF; ++f) cur[f] = *(const h8*)frag_addr(kt, f);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int NG = NI * KS;        // MFMA groups (one W fragment each)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = g / NI, j = g % NI;
            if (R == 2) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) if (f * NG / NF == g) nxt[f] = *(const h8*)frag_addr(kt + 1, f);
                if (DMA) {
#pragma unroll
                    for (int p = 0; p < 5; ++p) if (p * NG / 5 == g) dma_piece(kt, p);
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) mfma(j, i, ks, cur);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < STEPS; kt += 2) {
        if (R == 2) { step(kt, fa, fb); step(kt + 1, fb, fa); } else { step(kt, fa, fa); step(kt + 1, fa, fa); }
    }
    wait_vmcnt<0>();
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i) s += acc[j][i][0] + acc[j][i][AR - 1];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
static double g_warm = 1.0;
template <int R, bool BAR, int MS, int DMA, int SEG = 0>
void run(const char* name, const h8* src, const char* stream, float* out, long long* cyc) {
    auto k = lab<R, BAR, MS, DMA, SEG>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count() < g_warm) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 4 * STAGE, 0, src, stream, out, cyc);
        hipDeviceSynchronize();
    }
    hipEventRecord(e0);
    const int it = 200;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 4 * STAGE, 0, src, stream, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double flops = 2.0 * 256 * 320 * 32 * STEPS * 256.0;
    printf("%-52s %7.1f us %6.0f TF/s  %6.0f cyc/step  %.2f GHz\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12, avg / STEPS,
           avg / (ms * 1e-3) / 1e9);
    fflush(stdout);
}
int main() {
    std::vector<_Float16> hs(4096 * 8);
    unsigned st = 1;
    const bool zeros = getenv("LAB_ZEROS") != nullptr;
    if (getenv("LAB_WARM")) g_warm = atof(getenv("LAB_WARM"));
    for (auto& v : hs) { st = st * 1664525u + 1013904223u; v = zeros ? (_Float16)0.f : (_Float16)(((st >> 9) & 0xffff) / 32768.0f - 1.0f); }
    h8* src; float* out; long long* cyc; char* stream;
    const size_t sbytes = (size_t)257 * STEPS * 16384 + (1 << 22);
    hipMalloc(&src, hs.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&stream, sbytes);
    hipMemcpy(src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
    {   // random (or zero) fp16 stream
        std::vector<_Float16> big(sbytes / 2);
        for (auto& v : big) { st = st * 1664525u + 1013904223u; v = zeros ? (_Float16)0.f : (_Float16)(((st >> 9) & 0xffff) / 32768.0f - 1.0f); }
        hipMemcpy(stream, big.data(), sbytes, hipMemcpyHostToDevice);
    }
    printf("%s operands, warm-up %.1f s per variant; MFMA-bound = 1280 cycles/step\n", zeros ? "zero" : "random", g_warm);
    run<0, true, 16, 0>("mfma16 + barrier", src, stream, out, cyc);
    run<2, true, 16, 0>("mfma16 + reads interleaved", src, stream, out, cyc);
    run<2, true, 16, 9>("... + DMA act x9, 1 KiB contiguous pieces", src, stream, out, cyc);
    run<2, true, 16, 9, 64>("... + DMA act x9, 16 x 64 B segments", src, stream, out, cyc);
    run<2, true, 16, 9, 128>("... + DMA act x9, 8 x 128 B segments", src, stream, out, cyc);
    run<2, true, 16, 1>("... + DMA act stream, 1 KiB contiguous pieces", src, stream, out, cyc);
    run<2, true, 16, 1, 64>("... + DMA act stream, 16 x 64 B segments", src, stream, out, cyc);
    run<2, true, 16, 1, 128>("... + DMA act stream, 8 x 128 B segments", src, stream, out, cyc);
    run<1, true, 16, 9, 64>("hoisted reads + DMA act x9, 16 x 64 B segments", src, stream, out, cyc);
    return 0;
}
