// GroupNorm(32) (+SiLU) and LayerNorm over NHWC / token-major fp16 tensors; fp32 statistics.
// HBM-bound kernels: 16-byte vector loads/stores, wave-shuffle + LDS reductions, deterministic
// (fixed-order) partial sums so results do not depend on block scheduling or rank count.
//
// Reference semantics: GroupNorm32 (ldm/modules/diffusionmodules/util.py:223-225, eps 1e-5),
// Normalize (ldm/modules/attention.py:76-77, eps 1e-6), nn.LayerNorm (attention.py:226-228, eps 1e-5).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

// pixels per partial-sum chunk of the two-kernel path (round 4: 64 -> 256, the reduction tail of a stats workgroup was as long as its
// loads: 64x64 x 320 channels, B = 32: 50.1 -> 45.5 us for both kernels, B = 16: 32.1 -> 29.3; profiles/r04_gn_chunk.txt)
#define GN_PIX_PER_CHUNK 256

__device__ __forceinline__ const half_t* src_octet(const half_t* x0, int C0, const half_t* x1, int C1,
                                                   size_t pix, int o) {
    const int c = o << 3;
    return c < C0 ? x0 + pix * C0 + c : x1 + pix * C1 + (c - C0);
}

// partial[b][chunk][32][2] = (sum, sumsq) of group g over the chunk's pixels.
// Per-thread per-channel sums go to LDS and are reduced in a fixed order (no atomics -> bitwise reproducible).
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x0, int C0,
                                                        const half_t* __restrict__ x1, int C1, int HW,
                                                        float* __restrict__ partial, int pix_per_chunk) {
    extern __shared__ float red[];   // [PI][C][2]
    const int C = C0 + C1, P = C >> 3, cpg = C >> 5;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int tid = threadIdx.x;
    const int p_begin = chunk * pix_per_chunk;
    const int p_end = min(HW, p_begin + pix_per_chunk);
    // thread -> (pixel lane, octet); P <= 256: one octet per thread, several pixel lanes;
    // P > 256: one pixel lane, up to two octets per thread
    const int PI = P <= 256 ? 256 / P : 1;
    const int nslot = P <= 256 ? 1 : 2;
    for (int slot = 0; slot < nslot; ++slot) {
        int o, pi;
        bool active;
        if (P <= 256) { o = tid % P; pi = tid / P; active = pi < PI; }
        else { o = tid + slot * 256; pi = 0; active = o < P; }
        if (active) {
            float s[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
            // four independent 16-byte loads in flight per thread (a dependent one-load loop leaves HBM idle)
            int p = p_begin + pi;
            for (; p + 3 * PI < p_end; p += 4 * PI) {
                h8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const h8*)src_octet(x0, C0, x1, C1, (size_t)b * HW + p + u * PI, o);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e]; s[e] += f; q[e] += f * f; }
            }
            for (; p < p_end; p += PI) {
                const h8 v = *(const h8*)src_octet(x0, C0, x1, C1, (size_t)b * HW + p, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] += f * f; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((size_t)pi * C + (o << 3) + e) * 2] = s[e];
                red[((size_t)pi * C + (o << 3) + e) * 2 + 1] = q[e];
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int g = tid >> 1, which = tid & 1;
        float acc = 0.f;
        for (int pi = 0; pi < PI; ++pi)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) acc += red[((size_t)pi * C + c) * 2 + which];
        partial[((size_t)b * nchunk + chunk) * 64 + tid] = acc;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x0, int C0,
                                                        const half_t* __restrict__ x1, int C1, int HW,
                                                        const float* __restrict__ partial, int nchunk,
                                                        float inv_count, float eps,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu,
                                                        half_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // scale[C], shift[C], then stats[32][2]
    const int C = C0 + C1, P = C >> 3, cpg = C >> 5;
    const int b = blockIdx.y;
    float* scale = sc;
    float* shift = sc + C;
    float* stats = sc + 2 * C;
    // every block reduces the per-chunk partial sums of its sample itself (fixed order, double accumulation:
    // the E[x^2] - mean^2 form must not lose digits when |mean| >> std) -- no separate finalize launch
    {   // 8 threads per group, each sums every 8th chunk; the 8 sub-sums meet in a fixed xor-shuffle tree
        const int g = threadIdx.x >> 3, part = threadIdx.x & 7;
        double s = 0.0, q = 0.0;
        for (int c = part; c < nchunk; c += 8) {
            s += (double)partial[((size_t)b * nchunk + c) * 64 + g * 2];
            q += (double)partial[((size_t)b * nchunk + c) * 64 + g * 2 + 1];
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
        if (part == 0) {
            const double mean = s * inv_count;
            double var = q * inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            stats[g * 2] = (float)mean;
            stats[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        const float mean = stats[g * 2], rstd = stats[g * 2 + 1];
        const float w = gamma[c] * rstd;
        scale[c] = w;
        shift[c] = beta[c] - mean * w;
    }
    __syncthreads();
    const unsigned total = (unsigned)HW * (unsigned)P;     // < 2^31 (groupnorm_launch checks): 32-bit index arithmetic -- the
    // 64-bit `i / P` of the first version expanded to ~60 instructions per octet
    // four independent 16-byte loads in flight per thread
    for (unsigned i0 = blockIdx.x * 1024u + threadIdx.x; i0 < total; i0 += gridDim.x * 1024u) {
        h8 v[4];
        unsigned pix[4];
        int oct[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + (unsigned)u * 256u;
            pix[u] = i / (unsigned)P;
            oct[u] = (int)(i - pix[u] * (unsigned)P);
            if (i < total) v[u] = *(const h8*)src_octet(x0, C0, x1, C1, (size_t)b * HW + pix[u], oct[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + (unsigned)u * 256u >= total) break;
            h8 r;
            const f32x4 sc0 = *(const f32x4*)(scale + (oct[u] << 3)), sc1 = *(const f32x4*)(scale + (oct[u] << 3) + 4);
            const f32x4 sh0 = *(const f32x4*)(shift + (oct[u] << 3)), sh1 = *(const f32x4*)(shift + (oct[u] << 3) + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[u][e] * (e < 4 ? sc0[e & 3] : sc1[e & 3]) + (e < 4 ? sh0[e & 3] : sh1[e & 3]);
                if (silu) f = silu_f(f);
                r[e] = (half_t)f;
            }
            *(h8*)(out + ((size_t)b * HW + pix[u]) * C + (oct[u] << 3)) = r;
        }
    }
}

// Small feature maps (the 32x32 / 16x16 / 8x8 levels): ONE kernel per GroupNorm.  A workgroup owns NG whole groups of one
// sample, keeps that [HW][NG * cpg] slice in LDS (<= 96 KB), so x is read from HBM once instead of twice and the second
// launch disappears (these calls were latency-bound: 24 us for a 5 MB tensor at the 8x8 level).  Same fixed-order
// reductions as the two-kernel path -> bitwise independent of the batch.
// Twin layers (the UNet encoder's and every ControlNet's GroupNorm of the same shape) share ONE launch of the single-pass kernels:
// blockIdx.y selects the problem (engine.hip: replay_group).  Same bodies -> same bits.
struct GnProb { const half_t* x0; const half_t* x1; const float* gamma; const float* beta; half_t* out; };
struct GnArgsG { GnProb p[FGDM_MAX_GROUP]; int C0, C1, B, HW, NG, silu; float eps; };
struct GnRec { GnProb p; int C0, C1, B, HW, NG, silu; float eps; int np_sel; unsigned smem; };     // what a recorded launch keeps (<= FGDM_GROUP_BLOB)
static_assert(sizeof(GnRec) <= FGDM_GROUP_BLOB, "GroupNorm record must fit the recorder's blob");

template <int NT>   // threads per workgroup: 256 when 3+ workgroups share a CU, 512 when one 80 KB slice owns it (memory-level parallelism)
__device__ __forceinline__ void gn_fused_body(const half_t* __restrict__ x0, int C0,
                                                        const half_t* __restrict__ x1, int C1, int B, int HW, int NG,
                                                        float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu,
                                                        half_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int C = C0 + C1, cpg = C >> 5, CW = NG * cpg, OW = CW >> 3;     // slice width in channels / octets
    // the 32 / NG slices of ONE sample share an XCD (workgroup ids are dealt round-robin over the eight): see gn_reg_kernel
    const int NS = 32 / NG, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int b = (k / NS) * 8 + xcd, c_lo = (k % NS) * CW;
    if (b >= B) return;
    const int tid = threadIdx.x;
    half_t* tile = (half_t*)gsm;                                   // [HW][CW]
    float* red = (float*)(gsm + (size_t)HW * CW * 2);              // [PI][CW][2]
    const int PI = NT / OW;                                        // pixel lanes (OW <= 40)
    float* stats = red + (size_t)PI * CW * 2;                      // [NG][2]
    float* scale = stats + 8;                                      // [CW], then shift [CW]
    float* shift = scale + CW;
    const int o = tid % OW, pi = tid / OW;
    if (pi < PI) {
        float sm[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
        // eight independent 16-byte loads in flight per thread: one workgroup per CU must still pull its whole slice
        int p = pi;
        for (; p + 7 * PI < HW; p += 8 * PI) {
            h8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const h8*)src_octet(x0, C0, x1, C1, (size_t)b * HW + p + u * PI, (c_lo >> 3) + o);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                *(h8*)(tile + (size_t)(p + u * PI) * CW + (o << 3)) = v[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e]; sm[e] += f; sq[e] += f * f; }
            }
        }
        for (; p < HW; p += PI) {
            const h8 v = *(const h8*)src_octet(x0, C0, x1, C1, (size_t)b * HW + p, (c_lo >> 3) + o);
            *(h8*)(tile + (size_t)p * CW + (o << 3)) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sm[e] += f; sq[e] += f * f; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)pi * CW + (o << 3) + e) * 2] = sm[e];
            red[((size_t)pi * CW + (o << 3) + e) * 2 + 1] = sq[e];
        }
    }
    __syncthreads();
    // two-level fixed-order reduction: per channel over the pixel lanes (all threads), then per group over its channels
    for (int i = tid; i < 2 * CW; i += NT) {
        float acc = 0.f;
        for (int l = 0; l < PI; ++l) acc += red[(size_t)l * CW * 2 + i];
        red[i] = acc;                                              // lane-0 row now holds the per-channel totals
    }
    __syncthreads();
    if (tid < 2 * NG) {
        const int g = tid >> 1, which = tid & 1;
        double acc = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) acc += (double)red[c * 2 + which];
        stats[tid] = (float)(acc / ((double)HW * cpg));            // E[x], E[x^2] of group g
    }
    __syncthreads();
    if (tid < NG) {                                                // in place: (E[x], E[x^2]) -> (mean, rstd)
        const double mean = stats[2 * tid];
        double var = (double)stats[2 * tid + 1] - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = tid; c < CW; c += NT) {
        const int g = c / cpg;
        const float w = gamma[c_lo + c] * stats[2 * g + 1];
        scale[c] = w;
        shift[c] = beta[c_lo + c] - stats[2 * g] * w;
    }
    __syncthreads();
    const int total = HW * OW;
    for (int i = tid; i < total; i += NT) {
        const int p = i / OW, oo = i - p * OW;
        const h8 v = *(const h8*)(tile + (size_t)p * CW + (oo << 3));
        h8 r;
        const f32x4 sc0 = *(const f32x4*)(scale + (oo << 3)), sc1 = *(const f32x4*)(scale + (oo << 3) + 4);
        const f32x4 sh0 = *(const f32x4*)(shift + (oo << 3)), sh1 = *(const f32x4*)(shift + (oo << 3) + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * (e < 4 ? sc0[e & 3] : sc1[e & 3]) + (e < 4 ? sh0[e & 3] : sh1[e & 3]);
            if (silu) f = silu_f(f);
            r[e] = (half_t)f;
        }
        *(h8*)(out + ((size_t)b * HW + p) * C + c_lo + (oo << 3)) = r;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void gn_fused_kernel(const half_t* __restrict__ x0, int C0, const half_t* __restrict__ x1, int C1, int B, int HW,
                                                        int NG, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int silu, half_t* __restrict__ out) {
    gn_fused_body<NT>(x0, C0, x1, C1, B, HW, NG, eps, gamma, beta, silu, out);
}
template <int NT>
__global__ __launch_bounds__(NT) void gn_fused_group_kernel(const GnArgsG g) {
    const GnProb& p = g.p[blockIdx.y];
    gn_fused_body<NT>(p.x0, g.C0, p.x1, g.C1, g.B, g.HW, g.NG, g.eps, p.gamma, p.beta, g.silu, p.out);
}

// Large feature maps (the 64x64 / 32x32 levels: a slice of NG whole groups of one sample is 160-330 KB, more than LDS holds): the
// slice lives in the REGISTERS of one 1024-thread workgroup (NP 16-byte pieces per thread: 84 of the 128 registers a thread has at
// that occupancy), so x is read once instead of twice and the second launch disappears, exactly as in the kernel above; only the
// per-channel partial sums go through LDS.  One workgroup per CU; all of them load (every piece requested before the first use), then
// reduce, then store: HBM sees a pure read phase and a pure write phase.  Same fixed-order reductions -> independent of the batch.
template <int NP>
__device__ __forceinline__ void gn_reg_body(const half_t* __restrict__ x0, int C0,
                                                      const half_t* __restrict__ x1, int C1, int B, int HW, int NG,
                                                      float eps, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int silu,
                                                      half_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int C = C0 + C1, cpg = C >> 5, CW = NG * cpg, OW = CW >> 3;
    // a slice is 2 CW bytes of every 2 C-byte pixel row: the 32 / NG slices of ONE sample go to one XCD (workgroup ids are dealt
    // round-robin over the eight), neighbours in time, so its L2 fetches every 128-byte line of x once instead of once per slice
    // that owns a piece of it (without this the 64x64 level read 1.6x its bytes and lost to the two-kernel path)
    const int NS = 32 / NG, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int b = (k / NS) * 8 + xcd, c_lo = (k % NS) * CW;
    if (b >= B) return;
    const int tid = threadIdx.x;
    const int PI = 1024 / OW;                                      // pixel lanes; NP * PI >= HW (host)
    float* red = (float*)gsm;                                      // [PI][CW][2]
    float* stats = red + (size_t)PI * CW * 2;                      // [NG][2]
    float* scale = stats + 8;                                      // [CW], then shift [CW]
    float* shift = scale + CW;
    const int o = tid % OW, pi = tid / OW;
    const bool active = pi < PI;
    h8 v[NP];
    if (active) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = pi + u * PI;
            v[u] = (h8)(half_t)0;
            if (p < HW) v[u] = *(const h8*)src_octet(x0, C0, x1, C1, (size_t)b * HW + p, (c_lo >> 3) + o);
        }
        float sm[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
        for (int u = 0; u < NP; ++u)      // pixels past HW hold zeros: they add nothing
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e]; sm[e] += f; sq[e] += f * f; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)pi * CW + (o << 3) + e) * 2] = sm[e];
            red[((size_t)pi * CW + (o << 3) + e) * 2 + 1] = sq[e];
        }
    }
    __syncthreads();
    // two-level fixed-order reduction, as in gn_fused_kernel: per channel over the pixel lanes, then per group over its channels
    for (int i = tid; i < 2 * CW; i += 1024) {
        float acc = 0.f;
        for (int l = 0; l < PI; ++l) acc += red[(size_t)l * CW * 2 + i];
        stats[8 + 2 * CW + i] = acc;                               // per-channel totals, behind scale / shift
    }
    __syncthreads();
    const float* tot = stats + 8 + 2 * CW;
    if (tid < 2 * NG) {
        const int g = tid >> 1, which = tid & 1;
        double acc = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) acc += (double)tot[c * 2 + which];
        stats[tid] = (float)(acc / ((double)HW * cpg));            // E[x], E[x^2] of group g
    }
    __syncthreads();
    if (tid < NG) {
        const double mean = stats[2 * tid];
        double var = (double)stats[2 * tid + 1] - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = tid; c < CW; c += 1024) {
        const int g = c / cpg;
        const float w = gamma[c_lo + c] * stats[2 * g + 1];
        scale[c] = w;
        shift[c] = beta[c_lo + c] - stats[2 * g] * w;
    }
    __syncthreads();
    if (active) {
        const f32x4 sc0 = *(const f32x4*)(scale + (o << 3)), sc1 = *(const f32x4*)(scale + (o << 3) + 4);
        const f32x4 sh0 = *(const f32x4*)(shift + (o << 3)), sh1 = *(const f32x4*)(shift + (o << 3) + 4);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = pi + u * PI;
            if (p < HW) {
                h8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[u][e] * (e < 4 ? sc0[e & 3] : sc1[e & 3]) + (e < 4 ? sh0[e & 3] : sh1[e & 3]);
                    if (silu) f = silu_f(f);
                    r[e] = (half_t)f;
                }
                *(h8*)(out + ((size_t)b * HW + p) * C + c_lo + (o << 3)) = r;
            }
        }
    }
}

template <int NP>
__global__ __launch_bounds__(1024) void gn_reg_kernel(const half_t* __restrict__ x0, int C0, const half_t* __restrict__ x1, int C1, int B, int HW,
                                                      int NG, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int silu, half_t* __restrict__ out) {
    gn_reg_body<NP>(x0, C0, x1, C1, B, HW, NG, eps, gamma, beta, silu, out);
}
template <int NP>
__global__ __launch_bounds__(1024) void gn_reg_group_kernel(const GnArgsG g) {
    const GnProb& p = g.p[blockIdx.y];
    gn_reg_body<NP>(p.x0, g.C0, p.x1, g.C1, g.B, g.HW, g.NG, g.eps, p.gamma, p.beta, g.silu, p.out);
}

// one launch of a single-pass kernel, or (while the engine records) its record with the grouped form attached
static unsigned long long gn_shape_hash(const GnRec& r, int kind) {
    unsigned long long h = 1469598103934665603ull;
    const int v[10] = {kind, r.C0, r.C1, r.B, r.HW, r.NG, r.silu, r.np_sel, (int)r.smem, 0};
    for (int i = 0; i < 9; ++i) { h ^= (unsigned)v[i]; h *= 1099511628211ull; }
    unsigned e; memcpy(&e, &r.eps, 4);
    h ^= e; h *= 1099511628211ull;
    return h;
}
static bool gn_same_shape(const GnRec& a, const GnRec& b) {
    return a.C0 == b.C0 && a.C1 == b.C1 && a.B == b.B && a.HW == b.HW && a.NG == b.NG && a.silu == b.silu && a.eps == b.eps &&
           a.np_sel == b.np_sel && a.smem == b.smem;
}
template <typename SingleFn, typename GroupFn>
static int gn_group_dispatch(const void* const* args, int n, unsigned grid_x, hipStream_t s, unsigned block, SingleFn single, GroupFn group) {
    const GnRec& r0 = *(const GnRec*)args[0];
    if (n == 1) return single(r0, grid_x, s);
    bool same = n <= FGDM_MAX_GROUP;
    for (int k = 1; k < n && same; ++k) same = gn_same_shape(r0, *(const GnRec*)args[k]);
    if (!same) {            // (a hash collision: cannot happen for two real layers, but then each problem simply runs alone)
        for (int k = 0; k < n; ++k) { const int rc = single(*(const GnRec*)args[k], grid_x, s); if (rc != FGDM_OK) return rc; }
        return FGDM_OK;
    }
    GnArgsG g{};
    for (int k = 0; k < n; ++k) g.p[k] = ((const GnRec*)args[k])->p;
    g.C0 = r0.C0; g.C1 = r0.C1; g.B = r0.B; g.HW = r0.HW; g.NG = r0.NG; g.silu = r0.silu; g.eps = r0.eps;
    return group(g, r0, grid_x, (unsigned)n, s);
}
#define GN_SINGLE(KERNEL) [](const GnRec& r, unsigned gx, hipStream_t st) -> int {                                                       \
        hipLaunchKernelGGL(KERNEL, dim3(gx), dim3(GN_BLOCK), r.smem, st, r.p.x0, r.C0, r.p.x1, r.C1, r.B, r.HW, r.NG, r.eps, r.p.gamma,   \
                           r.p.beta, r.silu, r.p.out);                                                                                 \
        return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP; }
#define GN_GROUP(KERNEL) [](const GnArgsG& g, const GnRec& r, unsigned gx, unsigned n, hipStream_t st) -> int {                            \
        hipLaunchKernelGGL(KERNEL, dim3(gx, n), dim3(GN_BLOCK), r.smem, st, g);                                                       \
        return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP; }
// np_sel: 6 / 11 / 16 / 21 = gn_reg_kernel<NP>; 256 / 512 = gn_fused_kernel<NT>
static int gn_group_fn(const void* const* args, int n, unsigned grid_x, hipStream_t s) {
    const int sel = ((const GnRec*)args[0])->np_sel;
    switch (sel) {
#define GN_BLOCK 1024
        case 6: return gn_group_dispatch(args, n, grid_x, s, 1024, GN_SINGLE(gn_reg_kernel<6>), GN_GROUP(gn_reg_group_kernel<6>));
        case 11: return gn_group_dispatch(args, n, grid_x, s, 1024, GN_SINGLE(gn_reg_kernel<11>), GN_GROUP(gn_reg_group_kernel<11>));
        case 16: return gn_group_dispatch(args, n, grid_x, s, 1024, GN_SINGLE(gn_reg_kernel<16>), GN_GROUP(gn_reg_group_kernel<16>));
        case 21: return gn_group_dispatch(args, n, grid_x, s, 1024, GN_SINGLE(gn_reg_kernel<21>), GN_GROUP(gn_reg_group_kernel<21>));
#undef GN_BLOCK
#define GN_BLOCK 256
        case 256: return gn_group_dispatch(args, n, grid_x, s, 256, GN_SINGLE(gn_fused_kernel<256>), GN_GROUP(gn_fused_group_kernel<256>));
#undef GN_BLOCK
#define GN_BLOCK 512
        case 512: return gn_group_dispatch(args, n, grid_x, s, 512, GN_SINGLE(gn_fused_kernel<512>), GN_GROUP(gn_fused_group_kernel<512>));
#undef GN_BLOCK
        default: return FGDM_ERR_ARG;
    }
}
static thread_local bool g_gn_group = true;        // FGDM_GN_GROUP (read at fgdm_create, set by the engine per call): A/B knob, same bits either way
void groupnorm_set_group(bool on) { g_gn_group = on; }
// launch now, or record (with the grouped form attached when FGDM_GN_GROUP allows)
static int gn_single_pass_launch(const GnRec& r, unsigned grid_x, hipStream_t s) {
    const void* args[1] = {&r};
    if (!fgdm_recording()) return gn_group_fn(args, 1, grid_x, s);
    const GnRec rc = r;
    fgdm_record_generic([rc, grid_x](hipStream_t rs) -> int { const void* a1[1] = {&rc}; return gn_group_fn(a1, 1, grid_x, rs); },
                        (const void*)&gn_group_fn, g_gn_group ? gn_group_fn : nullptr, &r, sizeof(r), grid_x, gn_shape_hash(r, r.np_sel));
    return FGDM_OK;
}

size_t groupnorm_ws_floats(int B, int HW) {
    const int nchunk = (HW + 63) / 64;   // upper bound for every chunk size >= 64
    return (size_t)B * nchunk * 64 + (size_t)B * 64;
}

int groupnorm_launch(const half_t* x0, int C0, const half_t* x1, int C1, int B, int HW, const float* gamma,
                     const float* beta, float eps, int silu, half_t* out, float* ws, hipStream_t s) {
    const int C = C0 + C1;
    if ((C & 31) || (C0 & 7) || (C1 & 7) || C > 4096 || B <= 0 || HW <= 0 || (size_t)HW * (C >> 3) >= (1u << 31)) return FGDM_ERR_ARG;
    // register-resident single pass (gn_reg_kernel): the widest whole-group slice whose pieces fit a thread's registers.
    // Returns 1 when no slice fits (the caller goes on to the other paths), else the launch status.
    auto try_reg = [&]() -> int {
        // measured, B = 32 / 16 (tools/bench_norm.py, us, two kernels -> this one): 32x32 C = 640 28.8 -> 20.5 / 20.6 -> 16.9,
        // 640+640 45.0 -> 34.4 / 31.2 -> 27.8, 1280+640 67.1 -> 55.0.  The 64x64 level stays on two kernels: C = 320 50.7 -> 45.8 at
        // B = 32 but 32.1 -> 34.9 at B = 16, 320+320 (sixteen slices per sample, two rounds of workgroups) 94.4 -> 107.3, and nothing
        // end to end (which kernel runs must not depend on the batch: a sample's bits must not)
        static const int reg_hw = getenv("FGDM_GN_REG") ? atoi(getenv("FGDM_GN_REG")) : 1024;      // A/B knob: largest HW taken (0 = off)
        const bool reg_on = HW <= reg_hw;
        const int cpg = C >> 5;
        static const int ng_max = getenv("FGDM_GN_REG_NG") ? atoi(getenv("FGDM_GN_REG_NG")) : 4;        // experiment knob: widest slice, in groups
        for (int NG = ng_max; reg_on && NG >= 1; NG >>= 1) {           // the widest slice that fits: longer runs per pixel row
            const int CW = NG * cpg;
            if (CW & 7) continue;
            const int OW = CW >> 3;
            if (OW < 5 || OW > 64) continue;                       // (narrower: 16-64 byte runs per pixel row, the autoencoder's widths)
            const int PI = 1024 / OW, np = (HW + PI - 1) / PI;
            if (np > 21) continue;
            const size_t smem = ((size_t)PI * CW * 2 + 8 + 2 * (size_t)CW + 2 * (size_t)CW) * sizeof(float);
            if (smem > 150 * 1024) continue;
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)gn_reg_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_reg_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_reg_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_reg_kernel<21>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                    return FGDM_ERR_HIP;
                attr_set = true;
            }
            static bool attr2_set = false;
            if (!attr2_set) {
                if (hipFuncSetAttribute((const void*)gn_reg_group_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_reg_group_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_reg_group_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_reg_group_kernel<21>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                    return FGDM_ERR_HIP;
                attr2_set = true;
            }
            GnRec r{{x0, x1, gamma, beta, out}, C0, C1, B, HW, NG, silu, eps, np <= 6 ? 6 : np <= 11 ? 11 : np <= 16 ? 16 : 21, (unsigned)smem};
            return gn_single_pass_launch(r, (unsigned)((32 / NG) * 8 * ((B + 7) / 8)), s);
        }
        return 1;
    };
    // from 16x16 x 1280 channels (or 8x8 x 2560) upwards the register kernel is the faster single pass (tools/bench_norm.py, B = 32,
    // LDS kernel -> register kernel: 16x16 C = 1280 15.6 -> 14.4 us, 2560 33.5 -> 22.5; 8x8 C = 2560 13.4 -> 12.1, C = 1280 8.9 -> 9.1)
    if ((size_t)HW * C >= 64 * 2560) { const int rc = try_reg(); if (rc != 1) return rc; }
    // single-kernel path when NG whole groups of one sample fit in LDS
    {
        const int cpg = C >> 5;
        // first choice: slices of <= 40 KB (3+ workgroups per CU overlap their load / apply phases), else up to 64 KB (two per
        // CU).  Fatter slices (82 KB at the 32x32 level: one workgroup per CU, its load, reduce and store phases back to back)
        // measured 37.5 us for an 84 MB pass where the two-kernel path below moves 126 MB in about 25
        static const int max_kb = getenv("FGDM_GN_FUSED_MAXKB") ? atoi(getenv("FGDM_GN_FUSED_MAXKB")) : 64;      // tuning knob
        for (int pass = 0; pass < 2; ++pass)
        for (int NG = 4; NG >= 1; NG >>= 1) {
            const int CW = NG * cpg;
            const size_t slice = (size_t)HW * CW * 2;
            if ((CW & 7) || CW > 320 || slice > (size_t)std::min(pass == 0 ? 40 : 64, max_kb) * 1024) continue;
            const int NT = slice > 40 * 1024 ? 512 : 256;      // one fat slice per CU: twice the threads (and loads in flight)
            const int OW = CW >> 3, PI = NT / OW;
            const size_t smem = slice + ((size_t)PI * CW * 2 + 8 + 2 * (size_t)CW) * sizeof(float);
            if (smem > 150 * 1024) continue;
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)gn_fused_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_fused_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                    return FGDM_ERR_HIP;
                attr_set = true;
            }
            static bool attr2_set = false;
            if (!attr2_set) {
                if (hipFuncSetAttribute((const void*)gn_fused_group_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)gn_fused_group_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                    return FGDM_ERR_HIP;
                attr2_set = true;
            }
            GnRec r{{x0, x1, gamma, beta, out}, C0, C1, B, HW, NG, silu, eps, NT, (unsigned)smem};
            return gn_single_pass_launch(r, (unsigned)((32 / NG) * 8 * ((B + 7) / 8)), s);
        }
    }
    { const int rc = try_reg(); if (rc != 1) return rc; }
    // pixels per partial-sum chunk of the two-kernel path: a constant of the build (a sample's bits must not depend on it varying);
    // FGDM_GN_CHUNK is a tuning knob for experiments (multiples of 64, so that the workspace bound above holds)
    static const int ppc_env = getenv("FGDM_GN_CHUNK") ? atoi(getenv("FGDM_GN_CHUNK")) : GN_PIX_PER_CHUNK;
    const int ppc = (ppc_env >= 64 && ppc_env % 64 == 0) ? ppc_env : GN_PIX_PER_CHUNK;
    const int nchunk = (HW + ppc - 1) / ppc;
    float* partial = ws;
    const int P = C >> 3, PI = P <= 256 ? 256 / P : 1;
    FGDM_LAUNCH(gn_stats_kernel, dim3(nchunk, B), dim3(256), (size_t)PI * C * 2 * sizeof(float), s, x0, C0, x1,
                       C1, HW, partial, ppc);
    const float inv_count = 1.0f / ((float)HW * (float)(C / 32));
    const size_t total = (size_t)HW * (C >> 3);
    int gx = (int)((total + 1023) / 1024);
    // few, fat blocks: every block first reduces the sample's partial sums, so that prefix must be amortised
    // (measured on C3: 32 blocks per sample at B >= 8 is 6 % faster than 256)
    const int cap = 256 / (B < 8 ? B : 8);
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    FGDM_LAUNCH(gn_apply_kernel, dim3(gx, B), dim3(256), (2 * C + 64) * sizeof(float), s, x0, C0, x1, C1, HW,
                       partial, nchunk, inv_count, eps, gamma, beta, silu, out);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

// ---------------------------------------------------------------- LayerNorm: one wave per token row
template <int NO>   // octets per lane: C <= NO * 512
__global__ __launch_bounds__(256) void ln_kernel(const half_t* __restrict__ x, int rows, int C,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float eps, half_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int P = C >> 3;
    h8 v[NO];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const int o = lane + k * 64;
        if (o < P) {
            v[k] = *(const h8*)(x + (size_t)row * C + (o << 3));
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[k][e];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const int o = lane + k * 64;
        if (o < P) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[k][e] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const int o = lane + k * 64;
        if (o < P) {
            h8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = (o << 3) + e;
                r[e] = (half_t)(((float)v[k][e] - mean) * rstd * gamma[c] + beta[c]);
            }
            *(h8*)(out + (size_t)row * C + (o << 3)) = r;
        }
    }
}

int layernorm_launch(const half_t* x, int rows, int C, const float* gamma, const float* beta, float eps,
                     half_t* out, hipStream_t s) {
    if ((C & 7) || C > 2048 || rows <= 0) return FGDM_ERR_ARG;
    const dim3 grid((rows + 3) / 4), block(256);
    if (C <= 512) FGDM_LAUNCH(ln_kernel<1>, grid, block, 0, s, x, rows, C, gamma, beta, eps, out);
    else if (C <= 1024) FGDM_LAUNCH(ln_kernel<2>, grid, block, 0, s, x, rows, C, gamma, beta, eps, out);
    else FGDM_LAUNCH(ln_kernel<4>, grid, block, 0, s, x, rows, C, gamma, beta, eps, out);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

// LayerNorm over an fp32 token matrix (CLIP's fp32 residual stream): one wave per row, fp32 statistics; the result is stored
// as fp16 (it feeds a GEMM: autocast rounds it there) or as fp32 (final_layer_norm: the encoder's output)
template <typename OUT>
__global__ __launch_bounds__(256) void ln32_kernel(const float* __restrict__ x, int rows, int C, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, OUT* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float sum = 0.f;
    for (int c = lane * 4; c < C; c += 256) { const f32x4 v = *(const f32x4*)(xr + c); sum += v[0] + v[1] + v[2] + v[3]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; sq += d * d; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq / (float)C + eps);
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(size_t)row * C + c + e] = (OUT)((v[e] - mean) * rstd * gamma[c + e] + beta[c + e]);
    }
}
int layernorm32_launch(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, half_t* out16,
                       float* out32, hipStream_t s) {
    if ((C & 3) || rows <= 0 || (!out16 == !out32)) return FGDM_ERR_ARG;
    const dim3 grid((rows + 3) / 4), block(256);
    if (out16) FGDM_LAUNCH(ln32_kernel<half_t>, grid, block, 0, s, x, rows, C, gamma, beta, eps, out16);
    else FGDM_LAUNCH(ln32_kernel<float>, grid, block, 0, s, x, rows, C, gamma, beta, eps, out32);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

// LayerNorm partial sums of every token row, for a consumer GEMM that has the LayerNorm folded in (IgemmArgs::ln_stats),
// when the producing GEMM could not emit them from its own epilogue (2-stage kernel, split-K).  SAME slots (160 columns
// each; one slot when C is not a multiple of 160) and the SAME order of fp32 additions as the pipelined kernel's epilogue --
// eight channels of a 16-byte chunk, then the slot's chunks in order -- so the statistics, and with them every output bit, do
// not depend on which kernel produced them (a sample's result stays independent of the batch it is evaluated in).
__global__ __launch_bounds__(256) void row_stats_kernel(const half_t* __restrict__ x, int rows, int C, int slots,
                                                        float* __restrict__ stats) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * slots) return;
    const size_t row = i / slots;
    const int slot = (int)(i - row * slots), w = C / slots;
    const half_t* p = x + row * C + (size_t)slot * w;
    float sm = 0.f, sq = 0.f;
    for (int k = 0; k < (w >> 3); ++k) {
        const h8 v = *(const h8*)(p + (k << 3));
        float cs = 0.f, cq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; cs += f; cq += f * f; }
        sm += cs; sq += cq;
    }
    stats[i * 2] = sm; stats[i * 2 + 1] = sq;
}
int row_stats_slots(int C) { return (C % 160) == 0 ? C / 160 : 1; }
int row_stats_launch(const half_t* x, int rows, int C, float* stats, hipStream_t s) {
    if ((C & 7) || rows <= 0) return FGDM_ERR_ARG;
    const int slots = row_stats_slots(C);
    const size_t n = (size_t)rows * slots;
    FGDM_LAUNCH(row_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, rows, C, slots, stats);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}
