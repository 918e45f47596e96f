// Fused (flash-style) multi-head attention for gfx950: O = softmax(Q K^T d^-1/2) V, never materialising scores.
// Reference semantics: CrossAttention.forward, ldm/modules/attention.py:177-202 (self: context = x; cross: 77 tokens).
//
// Layout: Q [B, T, ldq], K [B, Tk, ldk] with head h at columns [h*D, (h+1)*D); V is consumed TRANSPOSED,
// Vt [B, H*D, ldvt] (keys contiguous), which the V-projection GEMM writes directly (OUT_F16_T epilogue).
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
//   S^T = K Q^T   : v_mfma_f32_32x32x16_f16 with A = K tile rows (keys), B = Q^T kept in registers, so each lane
//                   holds one query column -> the softmax row reduction is in-lane + one cross-half shuffle.
//   O^T += Vt P^T : the S^T accumulator registers, converted to fp16, ARE the B operand (k order permuted as
//                   16s + 8(j>>2) + 4h + (j&3)); the A operand reads Vt from LDS with the same key permutation.
// Online softmax state (m, l) and the O^T accumulators are per-lane scalars of that lane's query.
#include "common.h"

template <int D>
__global__ __launch_bounds__(256) void attn_kernel(const half_t* __restrict__ Q, int ldq,
                                                   const half_t* __restrict__ K, int ldk,
                                                   const half_t* __restrict__ Vt, int ldvt,
                                                   half_t* __restrict__ O, int ldo,
                                                   int H, int T, int Tk, float sl2e) {
    constexpr int DP = (D + 15) / 16 * 16;   // QK^T contraction length, padded to MFMA K
    constexpr int NKS = DP / 16;
    constexpr int DT = (D + 31) / 32;         // 32-row tiles of O^T
    constexpr int KS = DP * 2 + 16;           // K-tile row stride in bytes: odd multiple of 16 -> b128 conflict-free
    constexpr int VS = 64 * 2 + 8;            // Vt-tile row stride in bytes: 34 dwords -> b64 conflict-free
    constexpr int DC = D / 8;                 // 16-byte chunks per K row
    __shared__ __attribute__((aligned(16))) char Ks[64 * KS];
    __shared__ __attribute__((aligned(16))) char Vs[DT * 32 * VS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + lq;

    // zero the K pad columns [D, DP) once (tile loads never touch them) and the Vt pad rows [D, DT*32)
    if constexpr (DP > D) {
        for (int i = tid; i < 64 * (DP - D) / 8; i += 256) {
            const int key = i / ((DP - D) / 8), c = i % ((DP - D) / 8);
            *(h8*)(Ks + key * KS + (D + c * 8) * 2) = (h8)(half_t)0;
        }
    }
    for (int i = tid; i < (DT * 32 - D) * 16; i += 256) {
        const int r = D + i / 16, c = i % 16;
        *(h4*)(Vs + r * VS + c * 8) = (h4)(half_t)0;
    }

    // Q^T fragments (B operand): lane holds Q[q][16s + 8h .. +7]
    h8 qf[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const int c = 16 * s + 8 * lh;
        qf[s] = (h8)(half_t)0;
        if (c < D && q < T) qf[s] = *(const h8*)(Q + ((size_t)b * T + q) * ldq + head * D + c);
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const half_t* Kb = K + (size_t)b * Tk * ldk + head * D;
    const half_t* Vb = Vt + ((size_t)b * H + head) * D * ldvt;

    for (int k0 = 0; k0 < Tk; k0 += 64) {
        __syncthreads();   // previous tile fully consumed (also orders the pad-zeroing before the first tile)
        for (int i = tid; i < 64 * DC; i += 256) {
            const int key = i / DC, c = i % DC;
            h8 v = (h8)(half_t)0;
            if (k0 + key < Tk) v = *(const h8*)(Kb + (size_t)(k0 + key) * ldk + c * 8);
            *(h8*)(Ks + key * KS + c * 16) = v;
        }
        for (int i = tid; i < D * 8; i += 256) {
            const int r = i >> 3, c = i & 7;
            const h8 v = *(const h8*)(Vb + (size_t)r * ldvt + k0 + c * 8);
            h4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
            *(h4*)(Vs + r * VS + c * 16) = lo;
            *(h4*)(Vs + r * VS + c * 16 + 8) = hi;
        }
        __syncthreads();

        // ---- S^T = K Q^T for two 32-key sub-tiles
        f32x16 sacc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const h8 kf = *(const h8*)(Ks + (sub * 32 + lq) * KS + (16 * s + 8 * lh) * 2);
                sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[sub], 0, 0, 0);
            }
        }
        if (k0 + 64 > Tk) {   // ragged last tile: keys >= Tk get -inf
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= Tk) sacc[sub][r] = -INFINITY;
                }
        }
        // ---- online softmax for this lane's query (keys split over the two lane halves)
        float mx = sacc[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[sub][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2e);
        const float mb = m_new * sl2e;
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[sub][r] * sl2e - mb);
                sacc[sub][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;

        // ---- O^T += Vt P^T
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[sub][8 * s + j];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const char* vp = Vs + (t * 32 + lq) * VS + (sub * 32 + 16 * s + 4 * lh) * 2;
                    const h4 v0 = *(const h4*)vp;
                    const h4 v1 = *(const h4*)(vp + 16);
                    const h8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[t], 0, 0, 0);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (q < T) {
        half_t* op = O + ((size_t)b * T + q) * ldo + head * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = t * 32 + 8 * g + 4 * lh;
                if (dd < D) {
                    h4 pk = {(half_t)(oacc[t][4 * g] * inv), (half_t)(oacc[t][4 * g + 1] * inv),
                             (half_t)(oacc[t][4 * g + 2] * inv), (half_t)(oacc[t][4 * g + 3] * inv)};
                    *(h4*)(op + dd) = pk;
                }
            }
    }
}

int attention_launch(const half_t* Q, int ldq, const half_t* K, int ldk, const half_t* Vt, int ldvt, half_t* O,
                     int ldo, int B, int H, int T, int Tk, int d, hipStream_t s) {
    if (B <= 0 || H <= 0 || T <= 0 || Tk <= 0) return FGDM_ERR_ARG;
    if (ldvt < (Tk + 63) / 64 * 64 || (ldvt & 7) || (ldq & 7) || (ldk & 7) || (ldo & 3)) return FGDM_ERR_ARG;
    const float sl2e = 1.4426950408889634f / sqrtf((float)d);
    const dim3 grid((T + 127) / 128, H, B), block(256);
    switch (d) {
        case 40: hipLaunchKernelGGL(attn_kernel<40>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        case 80: hipLaunchKernelGGL(attn_kernel<80>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        case 160: hipLaunchKernelGGL(attn_kernel<160>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        default: return FGDM_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}
