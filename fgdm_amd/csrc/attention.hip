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
// The softmax is VALU-bound at d = 40 (one v_exp per score against 14 MFMAs per 64 keys), so the VALU work is cut:
//   * the row sum l = sum_k p rides on the MFMA: the spare rows of the padded O^T tile (d = 40 -> 64, 80 -> 96)
//     get a row of ONES in the Vt LDS tile, so O^T[row D] accumulates sum_k fp16(p) -- exactly the normaliser of
//     the fp16 probabilities used in the numerator -- and is rescaled together with O for free;
//   * deferred-max rescale: the running max is only raised (and O rescaled) when some query of the wave exceeds it
//     by more than 2^8; p <= 256 stays exact-range in fp16 and the final division by l cancels the offset;
//   * K / Vt tiles are double-buffered in LDS and prefetched into registers one tile ahead (one barrier per tile);
//   * d = 40 (padded to 48 in the contraction) has spare k slots: Q is pre-multiplied by log2(e) d^-1/2 and slot 40
//     carries K = 1, Q = -m (the running max, kept fp16-representable), so the MFMA itself delivers
//     s log2(e) d^-1/2 - m and the per-score work shrinks to max3 + exp2 + pack (FOLD).
//
// Measured and rejected (MI355X, C3): software-pipelining the loop inside a wave (QK^T of tile k+1 issued before the
// softmax of tile k, ping-pong score registers, K / V buffers out of phase, still one barrier per tile) needs 194 VGPRs
// at d = 40 -> 2 waves per SIMD instead of 3 -> 505 TF/s against 600; forced to 168 VGPRs it spills -> 436 TF/s.  The
// overlap of MFMA and VALU phases comes from the three resident waves here.  One 32-key sub-tile at a time (16 live
// score registers, 128 VGPRs, 4 waves per SIMD, no spill) is not faster either (549 vs 557-606 TF/s): v_exp_f32 issues
// at a quarter of the VALU rate, the 33 exp2 per 64 keys are ~530 of the ~1000 cycles a wave-tile takes on its SIMD.
#include "common.h"

#define ATT_THR 8.0f

template <int D>
__global__ __launch_bounds__(256) void attn_kernel(const half_t* __restrict__ Q, int ldq,
                                                   const half_t* __restrict__ K, int ldk,
                                                   const half_t* __restrict__ Vt, int ldvt,
                                                   half_t* __restrict__ O, int ldo,
                                                   int H, int T, int Tk, float sl2e) {
    constexpr int DP = (D + 15) / 16 * 16;   // QK^T contraction length, padded to MFMA K
    constexpr int NKS = DP / 16;
    constexpr int DT = (D + 31) / 32;         // 32-row tiles of O^T
    constexpr bool ONES = (DT * 32 > D);      // a spare O^T row exists: row D carries the softmax denominator
    constexpr bool FOLD = (DP > D) && (D % 8 == 0);   // a spare contraction slot exists: scale and -max ride on the MFMA
    constexpr int PS = D / 16, PH = (D % 16) / 8;     // fragment / lane half that hold contraction slot D (element 0)
    constexpr int KS = DP * 2 + 16;           // K-tile row stride in bytes: odd multiple of 16 -> b128 conflict-free
    constexpr int VS = 64 * 2 + 8;            // Vt-tile row stride in bytes: 34 dwords -> b64 conflict-free
    constexpr int DC = D / 8;                 // 16-byte chunks per K row
    constexpr int KCH = (64 * DC + 255) / 256;   // K chunks per thread per tile
    constexpr int VCH = (D * 8 + 255) / 256;     // Vt chunks per thread per tile
    constexpr int KBYTES = 64 * KS, VBYTES = DT * 32 * VS;
    __shared__ __attribute__((aligned(16))) char smem[2 * (KBYTES + VBYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    // XCD-aware mapping: workgroups b and b+8 share an XCD (and its 4 MiB L2).  Give every XCD a contiguous range of
    // (batch, head, q-block) triples with the q-block fastest, so all q-blocks of one (batch, head) stream the SAME
    // K / Vt through ONE L2 instead of eight (PMC: 389 MB fetched per launch against 252 MB of Q+K+V before).
    const int nqb = (T + 127) / 128;
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qd = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
    }
    const int qblk = logical % nqb, bh = logical / nqb;
    const int b = bh / H, head = bh - b * H;
    const int q = qblk * 128 + wave * 32 + lq;

    // pad regions of both buffers, written once: K pad columns [D, DP) = 0; Vt pad rows [D, DT*32) = 0, row D = 1
    for (int buf = 0; buf < 2; ++buf) {
        char* Ksb = smem + buf * (KBYTES + VBYTES);
        char* Vsb = Ksb + KBYTES;
        if constexpr (DP > D) {
            for (int i = tid; i < 64 * (DP - D) / 8; i += 256) {
                const int key = i / ((DP - D) / 8), c = i % ((DP - D) / 8);
                h8 pad = (h8)(half_t)0;
                if (FOLD && c == 0) pad[0] = (half_t)1;      // K[key][D] = 1: multiplies the -max kept in Q[q][D]
                *(h8*)(Ksb + key * KS + (D + c * 8) * 2) = pad;
            }
        }
        for (int i = tid; i < (DT * 32 - D) * 16; i += 256) {
            const int r = D + i / 16, c = i % 16;
            *(h4*)(Vsb + r * VS + c * 8) = (r == D) ? (h4)(half_t)1 : (h4)(half_t)0;
        }
    }

    // Q^T fragments (B operand): lane holds Q[q][16s + 8h .. +7]
    h8 qf[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const int c = 16 * s + 8 * lh;
        qf[s] = (h8)(half_t)0;
        if (c < D && q < T) qf[s] = *(const h8*)(Q + ((size_t)b * T + q) * ldq + head * D + c);
        if constexpr (FOLD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (half_t)((float)qf[s][j] * sl2e);
        }
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = FOLD ? 0.f : -INFINITY, l_run = 0.f;   // l_run only used when there is no spare row (D = 160)

    const half_t* Kb = K + (size_t)b * Tk * ldk + head * D;
    const half_t* Vb = Vt + ((size_t)b * H + head) * D * ldvt;

    // register prefetch of one K / Vt tile
    h8 kreg[KCH], vreg[VCH];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int u = 0; u < KCH; ++u) {
            const int i = tid + u * 256;
            const int key = i / DC, c = i - key * DC;
            kreg[u] = (h8)(half_t)0;
            if (i < 64 * DC && k0 + key < Tk) kreg[u] = *(const h8*)(Kb + (size_t)(k0 + key) * ldk + c * 8);
        }
#pragma unroll
        for (int u = 0; u < VCH; ++u) {
            const int i = tid + u * 256;
            if (i < D * 8) vreg[u] = *(const h8*)(Vb + (size_t)(i >> 3) * ldvt + k0 + (i & 7) * 8);
        }
    };
    auto store_tile = [&](int buf) {
        char* Ksb = smem + buf * (KBYTES + VBYTES);
        char* Vsb = Ksb + KBYTES;
#pragma unroll
        for (int u = 0; u < KCH; ++u) {
            const int i = tid + u * 256;
            const int key = i / DC, c = i - key * DC;
            if (i < 64 * DC) *(h8*)(Ksb + key * KS + c * 16) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < VCH; ++u) {
            const int i = tid + u * 256;
            if (i < D * 8) {
                const int r = i >> 3, c = i & 7;
                const h8 v = vreg[u];
                h4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
                *(h4*)(Vsb + r * VS + c * 16) = lo;
                *(h4*)(Vsb + r * VS + c * 16 + 8) = hi;
            }
        }
    };

    load_tile(0);
    store_tile(0);
    int cur = 0;
    for (int k0 = 0; k0 < Tk; k0 += 64) {
        const bool more = k0 + 64 < Tk;
        if (more) load_tile(k0 + 64);      // global loads fly while this tile is computed
        __syncthreads();                   // buffer `cur` is complete; buffer cur^1 is no longer read by anyone
        const char* Ks = smem + cur * (KBYTES + VBYTES);
        const char* Vs = Ks + KBYTES;

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
        // ---- online softmax for this lane's query (keys split over the two lane halves), log2 domain
        float psum = 0.f;
        if constexpr (FOLD) {
            // scores arrive as s log2(e) d^-1/2 - m_run; mx is therefore relative to the running max
            float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const bool first = (k0 == 0);
            if (first || !__all(mx <= ATT_THR)) {
                const float m_new = first ? mx : m_run + fmaxf(mx, 0.f);
                const float m_hat = (float)(half_t)m_new;          // the offset must be exactly what Q[q][D] can hold
                const float delta = m_run - m_hat;
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(delta);
                m_run = m_hat;
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[sub][r] += delta;   // this tile was taken against the old offset
                if (lh == PH) qf[PS][0] = (half_t)(-m_hat);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[sub][r] = __builtin_amdgcn_exp2f(sacc[sub][r]);
        } else {
        float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sacc[0][r], sacc[1][r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * sl2e;
        // raise the running max only when some query of this wave outgrew it by more than ATT_THR (deferred rescale)
        if (!__all(mx - m_run <= ATT_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[sub][r] * sl2e - m_run);
                sacc[sub][r] = p;
                if constexpr (!ONES) psum += p;
            }
        }
        if constexpr (!ONES) l_run += psum;

        // ---- O^T += Vt P^T   (with ONES: row D of O^T accumulates sum_k p)
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
        if (more) store_tile(cur ^ 1);     // safe: everyone passed this iteration's barrier after reading cur^1
        cur ^= 1;
    }

    float l_tot;
    if constexpr (ONES) {
        // row D of O^T lives in tile D/32, register (D%32 -> (r&3)+8(r>>2)+4h): fetch it from the lane half that owns it
        constexpr int rr = D % 32;
        constexpr int reg = (rr & 3) + 4 * (rr >> 3);     // register index within the half that has 4*lh == rr & 4
        constexpr int owner_half = (rr >> 2) & 1;
        const float mine = oacc[D / 32][reg];
        const float other = __shfl_xor(mine, 32);
        l_tot = (lh == owner_half) ? mine : other;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
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
                     int ldo, int B, int H, int T, int Tk, int d, int q_prescaled, hipStream_t s) {
    if (B <= 0 || H <= 0 || T <= 0 || Tk <= 0) return FGDM_ERR_ARG;
    if (ldvt < (Tk + 63) / 64 * 64 || (ldvt & 7) || (ldq & 7) || (ldk & 7) || (ldo & 3)) return FGDM_ERR_ARG;
    // q_prescaled: the to_q weights were packed with log2(e) d^-1/2 folded in (fgdm_finalize_weights), so Q arrives in the
    // log2 domain with ONE fp16 rounding; the kernels then multiply by exactly 1
    const float sl2e = q_prescaled ? 1.0f : 1.4426950408889634f / sqrtf((float)d);
    const dim3 grid(((T + 127) / 128) * H * B), block(256);
    switch (d) {
        case 40: hipLaunchKernelGGL(attn_kernel<40>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        case 80: hipLaunchKernelGGL(attn_kernel<80>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        case 160: hipLaunchKernelGGL(attn_kernel<160>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        default: return FGDM_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}
