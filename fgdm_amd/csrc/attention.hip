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
// (variants that were measured and rejected, and what bounds these kernels: DESIGN.md section 4.2 / 4.3)
#include "common.h"
#include <type_traits>
#include <algorithm>
#include <stdlib.h>

#pragma clang diagnostic ignored "-Winline-asm"     // M0 named as an asm clobber (att_dma16): reserved register, on purpose
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
    // Cross-attention (Tk <= 128: K / Vt of a head are 6-25 KB, nothing to share) is bound by its Q reads and O writes, and a
    // head's slice of a token row is 2 D bytes of a 2 H D-byte row: there the HEAD runs fastest, so the H workgroups that touch
    // the same 128 rows are neighbours on one XCD and every 128-byte line of Q and O crosses the fabric once instead of once per
    // head that owns a piece of it (only Tk <= 64 still comes here with few keys; the text tokens have attn_cross_kernel below)
    int qblk, b, head;
    if (Tk <= 128) {
        head = logical % H;
        const int rest = logical / H;
        qblk = rest % nqb; b = rest / nqb;
    } else {
        qblk = logical % nqb;
        const int bh = logical / nqb;
        b = bh / H; head = bh - b * H;
    }
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

    // register prefetch of one K / Vt tile.  Everything per-thread about the copy is loop-invariant and computed here, the
    // tile base is wave-uniform (scalar), chunk validity is wave-uniform too (64 * DC and D * 8 are multiples of 64), and
    // the K loop below is unrolled over the two LDS buffers so that their offsets are immediates: the loop carries no
    // address arithmetic or per-lane predicates (PMC, round 2: 114 VALU instructions per 64-key tile of which 64 were the
    // softmax itself; the kernel is vector-issue bound, so the other 50 cost as much as a fifth of the tile).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int k_goff[KCH], k_loff[KCH], k_key[KCH], v_goff[VCH], v_loff[VCH];
#pragma unroll
    for (int u = 0; u < KCH; ++u) {
        const int i = tid + u * 256, key = i / DC, c = i - key * DC;
        k_key[u] = key; k_goff[u] = key * ldk + c * 8; k_loff[u] = key * KS + c * 16;
    }
#pragma unroll
    for (int u = 0; u < VCH; ++u) {
        const int i = tid + u * 256, r = i >> 3, c = i & 7;
        v_goff[u] = r * ldvt + c * 8; v_loff[u] = r * VS + c * 16;
    }
    h8 kreg[KCH], vreg[VCH];
    auto load_tile = [&](int k0, bool ragged) {
        const half_t* kb = Kb + (size_t)k0 * ldk;
        const half_t* vb = Vb + k0;
#pragma unroll
        for (int u = 0; u < KCH; ++u)
            if ((u * 4 + wave_u) * 64 < 64 * DC) {
                int off = k_goff[u];
                // rows past Tk: re-read the last valid row (their scores are masked to -inf below; never out of bounds)
                if (ragged) off += (min(k_key[u], Tk - 1 - k0) - k_key[u]) * ldk;
                kreg[u] = *(const h8*)(kb + off);
            }
#pragma unroll
        for (int u = 0; u < VCH; ++u)
            if ((u * 4 + wave_u) * 64 < D * 8) vreg[u] = *(const h8*)(vb + v_goff[u]);     // Vt rows are padded to 64 keys
    };
    auto store_tile = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        char* Ksb = smem + buf * (KBYTES + VBYTES);
        char* Vsb = Ksb + KBYTES;
#pragma unroll
        for (int u = 0; u < KCH; ++u)
            if ((u * 4 + wave_u) * 64 < 64 * DC) *(h8*)(Ksb + k_loff[u]) = kreg[u];
#pragma unroll
        for (int u = 0; u < VCH; ++u)
            if ((u * 4 + wave_u) * 64 < D * 8) {
                const h8 v = vreg[u];
                h4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
                *(h4*)(Vsb + v_loff[u]) = lo;
                *(h4*)(Vsb + v_loff[u] + 8) = hi;
            }
    };
    // fragment addresses inside a tile (bytes): loop-invariant too
    const int kfrag = lq * KS + 8 * lh * 2, vfrag = lq * VS + 4 * lh * 2;

    load_tile(0, Tk < 64);
    store_tile(std::integral_constant<int, 0>{});
    // NSUB = 1: the tile's second 32-key sub-tile lies entirely beyond Tk (cross-attention: 77 keys = 64 + 13) and is skipped --
    // its QK^T and PV MFMAs, its 16 exponentials per lane, its fragment reads
    auto tile = [&](int k0, auto curc, auto nsubc) {
        constexpr int cur = decltype(curc)::value;
        constexpr int NSUB = decltype(nsubc)::value;
        const bool more = k0 + 64 < Tk;
        if (more) load_tile(k0 + 64, k0 + 128 > Tk);      // global loads fly while this tile is computed
        __syncthreads();                   // buffer `cur` is complete; buffer cur^1 is no longer read by anyone
        const char* Ks = smem + cur * (KBYTES + VBYTES);
        const char* Vs = Ks + KBYTES;

        // ---- S^T = K Q^T for two 32-key sub-tiles.  ALL K fragments of the tile are requested first, then the MFMAs run
        // behind counted lgkmcnt waits: a read-wait-MFMA chain per fragment left the wave parked on LDS latency ~13 times per
        // tile (the ISA of the previous version: ds_read_b128, s_waitcnt lgkmcnt(0), v_mfma ... six times over).
        h8 kf[2][NKS];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int s = 0; s < NKS; ++s) kf[sub][s] = *(const h8*)(Ks + kfrag + sub * 32 * KS + 16 * s * 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sacc[2];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][s], qf[s], sacc[sub], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the V^T fragments of the whole tile are requested NOW (into the registers the K fragments just left): they land
        // while the softmax below keeps the VALU busy, so the PV MFMAs start without an LDS round trip
        h4 v0f[2][2][DT], v1f[2][2][DT];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const char* vp = Vs + vfrag + t * 32 * VS + (sub * 32 + 16 * s) * 2;
                    v0f[sub][s][t] = *(const h4*)vp;
                    v1f[sub][s][t] = *(const h4*)(vp + 16);
                }
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 64 > Tk) {   // ragged last tile: keys >= Tk get -inf
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
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
            float mx = sacc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
            if constexpr (NSUB == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
            }
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
                for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[sub][r] += delta;   // this tile was taken against the old offset
                if (lh == PH) qf[PS][0] = (half_t)(-m_hat);
            }
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[sub][r] = __builtin_amdgcn_exp2f(sacc[sub][r]);
        } else {
        float mx = sacc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
        if constexpr (NSUB == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
        }
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
        for (int sub = 0; sub < NSUB; ++sub)
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
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[sub][8 * s + j];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const h4 v0 = v0f[sub][s][t], v1 = v1f[sub][s][t];
                    const h8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[t], 0, 0, 0);
                }
            }
        }
        if (more) store_tile(std::integral_constant<int, cur ^ 1>{});     // safe: everyone passed this iteration's barrier after reading cur^1
    };
    using one_t = std::integral_constant<int, 1>;
    using two_t = std::integral_constant<int, 2>;
    for (int k0 = 0; k0 < Tk; k0 += 128) {
        if (k0 + 32 >= Tk) tile(k0, std::integral_constant<int, 0>{}, one_t{});
        else tile(k0, std::integral_constant<int, 0>{}, two_t{});
        if (k0 + 64 < Tk) {
            if (k0 + 96 >= Tk) tile(k0 + 64, std::integral_constant<int, 1>{}, one_t{});
            else tile(k0 + 64, std::integral_constant<int, 1>{}, two_t{});
        }
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


// =====================================================================================================================
// Cross-attention (64 < Tk <= 96: the 77 text tokens).  The kernel above gives every 128 queries their own workgroup, which
// stages K / V^T tile by tile behind barriers: with two tiles of keys that is three dependent memory round trips and two barriers
// for 50 MFMAs -- latency-bound at 2.2 TB/s of Q + O traffic (72 us at T = 4096, d = 40).  Here a workgroup stages ALL keys of its
// (batch, head) once (K [96][DP], V^T [d][96] with the ones row), passes ONE barrier, and then each of its four waves walks CPW
// chunks of 32 queries on its own: S^T = K Q^T for the three 32-key sub-tiles in one go, a plain (single-pass) softmax, O^T = V^T P^T;
// the next chunk's Q rows are requested before the current chunk is computed.  Heads run fastest over workgroups (see above).
template <int D>
__global__ __launch_bounds__(256) void attn_cross_kernel(const half_t* __restrict__ Q, int ldq,
                                                         const half_t* __restrict__ K, int ldk,
                                                         const half_t* __restrict__ Vt, int ldvt,
                                                         half_t* __restrict__ O, int ldo,
                                                         int H, int T, int Tk, float sl2e, int cpw) {
    constexpr int NS = 3, KEYS = NS * 32;
    constexpr int DP = (D + 15) / 16 * 16, NKS = DP / 16, DT = (D + 31) / 32;
    constexpr bool ONES = (DT * 32 > D);
    constexpr bool FOLD = (DP > D) && (D % 8 == 0);      // same Q scaling rule as attn_kernel (the spare slot itself stays 0 here)
    constexpr int KS = DP * 2 + 16;            // K row stride (bytes): odd multiple of 16
    constexpr int VS = KEYS * 2 + 72;          // V^T row stride: 66 dwords = 2 mod 32 -> b64 reads conflict-free like the 34 above
    constexpr int DC = D / 8;
    // Q rows enter and O rows leave through a wave-private [32][D] LDS tile, so that global memory sees 16-byte pieces of whole
    // 2 D-byte row slices (a lane per query row means 64 different lines per instruction, 8 bytes each on the way out)
    constexpr bool RELAY = D <= 80;
    constexpr int RS = D * 2 + 8;              // 22 / 42 dwords per row: 8-byte accesses of 16 consecutive rows hit 16 banks
    constexpr int PCS = 32 * (D / 8), KP = (PCS + 63) / 64;      // 16-byte pieces of a 32-row chunk; per lane
    __shared__ __attribute__((aligned(16))) char smem[KEYS * KS + DT * 32 * VS + (RELAY ? 4 * 32 * RS : 0)];
    char* Ks = smem;
    char* Vs = smem + KEYS * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int qpw = 128 * cpw;                               // queries per workgroup
    const int ncb = (T + qpw - 1) / qpw;
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qd = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
    }
    const int head = logical % H, rest = logical / H;
    const int cblk = rest % ncb, b = rest / ncb;

    const half_t* Kb = K + (size_t)b * Tk * ldk + head * D;
    const half_t* Vb = Vt + ((size_t)b * H + head) * D * ldvt;
    // ---- stage K (rows past Tk repeat the last valid one: their scores are masked) and V^T (rows are padded to 64-key multiples):
    // every global load of the workgroup is issued before the first LDS store -- ONE memory round trip (as loops of load / store
    // pairs the staging was 5-6 dependent round trips, most of a workgroup's life)
    constexpr int KPC = KEYS * (DP / 8), VPC = DT * 32 * (KEYS / 8);
    constexpr int KPT = (KPC + 255) / 256, VPT = (VPC + 255) / 256;
    h8 kst[KPT], vst[VPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = tid + 256 * u, key = i / (DP / 8), c = i - key * (DP / 8);
        kst[u] = (h8)(half_t)0;
        if (i < KPC && c < DC) kst[u] = *(const h8*)(Kb + (size_t)min(key, Tk - 1) * ldk + c * 8);
    }
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid + 256 * u, r = i / (KEYS / 8), c = i - r * (KEYS / 8);
        vst[u] = (r == D) ? (h8)(half_t)1 : (h8)(half_t)0;
        if (i < VPC && r < D) vst[u] = *(const h8*)(Vb + (size_t)r * ldvt + c * 8);
    }
    char* scr = smem + KEYS * KS + DT * 32 * VS + (threadIdx.x >> 6) * 32 * RS;      // this wave's tile
    // the chunk's Q rows: per-lane fragments straight from global memory, or (RELAY) 16-byte pieces in row order
    constexpr int NQ = RELAY ? KP : NKS;
    auto load_q = [&](int q0, h8 (&qr)[NQ]) {      // q0: first query of the chunk (wave-uniform)
        if constexpr (RELAY) {
#pragma unroll
            for (int u = 0; u < KP; ++u) {
                const int c = lane + 64 * u, row = c / (D / 8), ck = c - row * (D / 8);
                qr[u] = (h8)(half_t)0;
                if (c < PCS && q0 + row < T) qr[u] = *(const h8*)(Q + ((size_t)b * T + q0 + row) * ldq + head * D + ck * 8);
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < NKS; ++s2) {
                const int c = 16 * s2 + 8 * lh;
                qr[s2] = (h8)(half_t)0;
                if (c < D && q0 + lq < T) qr[s2] = *(const h8*)(Q + ((size_t)b * T + q0 + lq) * ldq + head * D + c);
            }
        }
    };
    const int q_first = cblk * qpw + wave * 32;              // chunk i of this wave: queries q_first + i * 128 + [0, 32)
    h8 qn[NQ];
    load_q(q_first, qn);
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = tid + 256 * u, key = i / (DP / 8), c = i - key * (DP / 8);
        if (i < KPC) *(h8*)(Ks + key * KS + c * 16) = kst[u];
    }
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid + 256 * u, r = i / (KEYS / 8), c = i - r * (KEYS / 8);
        if (i < VPC) {
            const h8 v = vst[u];
            const h4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
            *(h4*)(Vs + r * VS + c * 16) = lo;
            *(h4*)(Vs + r * VS + c * 16 + 8) = hi;
        }
    }
    __syncthreads();

    const int kfrag = lq * KS + 8 * lh * 2, vfrag = lq * VS + 4 * lh * 2;
    for (int i = 0; i < cpw; ++i) {
        const int q0 = __builtin_amdgcn_readfirstlane(q_first + i * 128);
        if (q0 >= T) break;                                  // wave-uniform: this chunk starts past the end
        const int q = q0 + lq;
        h8 qf[NKS];
        if constexpr (RELAY) {
#pragma unroll
            for (int u = 0; u < KP; ++u) {
                const int c = lane + 64 * u, row = c / (D / 8), ck = c - row * (D / 8);
                if (c < PCS) {
                    const h4 lo = {qn[u][0], qn[u][1], qn[u][2], qn[u][3]}, hi = {qn[u][4], qn[u][5], qn[u][6], qn[u][7]};
                    *(h4*)(scr + row * RS + ck * 16) = lo;
                    *(h4*)(scr + row * RS + ck * 16 + 8) = hi;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < NKS; ++s2) {
                const int c = 16 * s2 + 8 * lh;
                qf[s2] = (h8)(half_t)0;
                if (c < D) {
                    const h4 lo = *(const h4*)(scr + lq * RS + c * 2), hi = *(const h4*)(scr + lq * RS + c * 2 + 8);
                    qf[s2] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragments read before the tile is reused for O
        } else {
#pragma unroll
            for (int s2 = 0; s2 < NKS; ++s2) qf[s2] = qn[s2];
        }
        if (i + 1 < cpw) load_q(q0 + 128, qn);
        if constexpr (FOLD) {
#pragma unroll
            for (int s2 = 0; s2 < NKS; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[s2][j] = (half_t)((float)qf[s2][j] * sl2e);
        }
        // ---- S^T = K Q^T, all keys
        f32x16 sacc[NS];
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < NKS; ++s2) {
                const h8 kf = *(const h8*)(Ks + kfrag + sub * 32 * KS + 16 * s2 * 2);
                sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s2], sacc[sub], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {           // only the last sub-tile can hold keys >= Tk (64 < Tk <= 96)
            const int key = (NS - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= Tk) sacc[NS - 1][r] = -INFINITY;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int sub = 0; sub < NS; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[sub][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float sc = FOLD ? 1.0f : sl2e;
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < NS; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f((sacc[sub][r] - mx) * sc);
                sacc[sub][r] = pv;
                if constexpr (!ONES) psum += pv;
            }
        // ---- O^T = V^T P^T (row D of O^T: the sum of the fp16 probabilities)
        f32x16 oacc[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
#pragma unroll
        for (int sub = 0; sub < NS; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[sub][8 * s2 + j];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const char* vp = Vs + vfrag + t * 32 * VS + (sub * 32 + 16 * s2) * 2;
                    const h4 v0 = *(const h4*)vp, v1 = *(const h4*)(vp + 16);
                    const h8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[t], 0, 0, 0);
                }
            }
        float l_tot;
        if constexpr (ONES) {
            constexpr int rr = D % 32;
            constexpr int reg = (rr & 3) + 4 * (rr >> 3);
            constexpr int owner_half = (rr >> 2) & 1;
            const float mine = oacc[D / 32][reg];
            const float other = __shfl_xor(mine, 32);
            l_tot = (lh == owner_half) ? mine : other;
        } else {
            l_tot = psum + __shfl_xor(psum, 32);
        }
        const float inv = 1.0f / l_tot;
        if constexpr (RELAY) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dd = t * 32 + 8 * g + 4 * lh;
                    if (dd < D) {
                        h4 pk = {(half_t)(oacc[t][4 * g] * inv), (half_t)(oacc[t][4 * g + 1] * inv),
                                 (half_t)(oacc[t][4 * g + 2] * inv), (half_t)(oacc[t][4 * g + 3] * inv)};
                        *(h4*)(scr + lq * RS + dd * 2) = pk;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < KP; ++u) {
                const int c = lane + 64 * u, row = c / (D / 8), ck = c - row * (D / 8);
                if (c < PCS && q0 + row < T) {
                    const h4 lo = *(const h4*)(scr + row * RS + ck * 16), hi = *(const h4*)(scr + row * RS + ck * 16 + 8);
                    *(h8*)(O + ((size_t)b * T + q0 + row) * ldo + head * D + ck * 8) = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile is free for the next chunk's Q
        } else if (q < T) {
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
}

// =====================================================================================================================
// Ping-pong variant for long self-attention (round 2).  PMC on the kernel above (T = Tk = 4096, d = 40): per SIMD the matrix
// pipe is busy 40 % of the time and the vector ALU 52 % -- and the two hardly ever at the same moment: three free-running
// waves per SIMD drift into the same phase, all in their softmax, then all in their MFMAs.  Here a workgroup has EIGHT waves,
// two per SIMD: group A (waves 0-3) and group B (waves 4-7) own 128 queries each and run the SAME program one barrier apart,
// so that on every SIMD one wave is in its MFMA slot {P V of tile k, Q K^T of tile k+1} while its partner is in its VALU slot
// {softmax of tile k, LDS writes of the next K / V^T tile}; two raw s_barriers per tile keep them there.
//   slot:      0        1        2        3        4   ...
//   group A:  QK(0)    X(0)     Y(0)     X(1)     Y(1)          X(k) = softmax(k) + stage tile, Y(k) = PV(k) + QK(k+1)
//   group B:   -       QK(0)    X(0)     Y(0)     X(1)
// K / V^T live in LDS as PAIRS p(j) = (K tile j+1, V^T tile j), double-buffered (buffer j & 1; K tile 0 is "p(-1)"): p(j) is
// read in slots 2j+2 (A) and 2j+3 (B), each wave writes its share of it in its own X phase -- A in X(j) (slot 2j+1), B in
// X(j-1) (slot 2j) -- which is after the last read of p(j-2) (slot 2j-1) and before the first read of p(j).  Every wave
// executes the same number of barriers (B one extra at the start, A one extra at the end).
// (measured and rejected around this kernel, and the software-pipelined variants of rounds 2 and 3: DESIGN.md section 4.2 / 4.3)
template <int D>
__global__ __launch_bounds__(512) void attn_pp_kernel(const half_t* __restrict__ Q, int ldq,
                                                      const half_t* __restrict__ K, int ldk,
                                                      const half_t* __restrict__ Vt, int ldvt,
                                                      half_t* __restrict__ O, int ldo,
                                                      int H, int T, int Tk, float sl2e) {
    constexpr int DP = (D + 15) / 16 * 16, NKS = DP / 16, DT = (D + 31) / 32;
    constexpr bool ONES = (DT * 32 > D);
    constexpr bool FOLD = (DP > D) && (D % 8 == 0);
    constexpr int PS = D / 16, PH = (D % 16) / 8;
    constexpr int KS = DP * 2 + 16, VS = 64 * 2 + 8, DC = D / 8;
    constexpr int NCH = 64 * DC;                    // 16-byte chunks of a K tile = of a V^T tile (D * 8)
    constexpr int CH = (NCH + 511) / 512;           // chunks per thread per tile
    constexpr int KBYTES = 64 * KS, VBYTES = DT * 32 * VS, PAIR = KBYTES + VBYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * PAIR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                      // 0 = A, 1 = B
    const int lq = lane & 31, lh = lane >> 5;
    const int nqb = (T + 255) / 256;
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qd = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
    }
    const int qblk = logical % nqb, bh = logical / nqb;
    const int b = bh / H, head = bh - b * H;
    const int q = qblk * 256 + wave * 32 + lq;
    const int nt = (Tk + 63) / 64;

    // pad regions of both buffers, written once (visible after the prologue barrier)
    for (int buf = 0; buf < 2; ++buf) {
        char* Ksb = smem + buf * PAIR;
        char* Vsb = Ksb + KBYTES;
        if constexpr (DP > D) {
            for (int i = tid; i < 64 * (DP - D) / 8; i += 512) {
                const int key = i / ((DP - D) / 8), c = i % ((DP - D) / 8);
                h8 pad = (h8)(half_t)0;
                if (FOLD && c == 0) pad[0] = (half_t)1;
                *(h8*)(Ksb + key * KS + (D + c * 8) * 2) = pad;
            }
        }
        for (int i = tid; i < (DT * 32 - D) * 16; i += 512) {
            const int r = D + i / 16, c = i % 16;
            *(h4*)(Vsb + r * VS + c * 8) = (r == D) ? (h4)(half_t)1 : (h4)(half_t)0;
        }
    }

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
    float m_run = FOLD ? 0.f : -INFINITY, l_run = 0.f;

    const half_t* Kb = K + (size_t)b * Tk * ldk + head * D;
    const half_t* Vb = Vt + ((size_t)b * H + head) * D * ldvt;

    // per-thread, loop-invariant parts of the tile copy; chunk validity is wave-uniform (NCH is a multiple of 64)
    int k_goff[CH], k_loff[CH], k_key[CH], v_goff[CH], v_loff[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int i = tid + u * 512;
        const int key = i / DC, c = i - key * DC;
        k_key[u] = key; k_goff[u] = key * ldk + c * 8; k_loff[u] = key * KS + c * 16;
        const int r = i >> 3, cv = i & 7;
        v_goff[u] = r * ldvt + cv * 8; v_loff[u] = r * VS + cv * 16;
    }
    h8 kreg[CH], vreg[CH];
    // K tile kt and V^T tile vtile -> registers (either may be absent: kt >= nt / vtile < 0)
    auto load_pair = [&](int kt, int vtile) {
        if (kt < nt) {
            const int k0 = kt * 64;
            const half_t* kb = Kb + (size_t)k0 * ldk;
            const bool ragged = k0 + 64 > Tk;
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if ((u * 8 + wave) * 64 < NCH) {
                    int off = k_goff[u];
                    if (ragged) off += (min(k_key[u], Tk - 1 - k0) - k_key[u]) * ldk;
                    kreg[u] = *(const h8*)(kb + off);
                }
        }
        if (vtile >= 0 && vtile < nt) {
            const half_t* vb = Vb + vtile * 64;
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if ((u * 8 + wave) * 64 < NCH) vreg[u] = *(const h8*)(vb + v_goff[u]);
        }
    };
    auto store_pair = [&](int kt, int vtile, int buf) {
        char* Ksb = smem + buf * PAIR;
        char* Vsb = Ksb + KBYTES;
        if (kt < nt) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if ((u * 8 + wave) * 64 < NCH) *(h8*)(Ksb + k_loff[u]) = kreg[u];
        }
        if (vtile >= 0 && vtile < nt) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if ((u * 8 + wave) * 64 < NCH) {
                    const h8 v = vreg[u];
                    h4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
                    *(h4*)(Vsb + v_loff[u]) = lo;
                    *(h4*)(Vsb + v_loff[u] + 8) = hi;
                }
        }
    };
    auto barrier = [&]() { __syncthreads(); };    // workgroup-scope fences + s_barrier: LDS traffic drained, global prefetches stay in flight
    const int kfrag = lq * KS + 8 * lh * 2, vfrag = lq * VS + 4 * lh * 2;

    f32x16 sacc[2];
    h8 pf[2][2];                       // fp16 probabilities of the tile: the B operand of P V
    auto qk = [&](int buf) {           // S^T = K Q^T of the K tile in `buf`
        const char* Ks = smem + buf * PAIR;
        h8 kf[2][NKS];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < NKS; ++s) kf[sub][s] = *(const h8*)(Ks + kfrag + sub * 32 * KS + 16 * s * 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][s], qf[s], sacc[sub], 0, 0, 0);
        }
    };
    auto pv = [&](int buf) {           // O^T += V^T P^T with the V^T tile in `buf`
        const char* Vs = smem + buf * PAIR + KBYTES;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const char* vp = Vs + vfrag + t * 32 * VS + (sub * 32 + 16 * s) * 2;
                    const h4 v0 = *(const h4*)vp, v1 = *(const h4*)(vp + 16);
                    const h8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][s], oacc[t], 0, 0, 0);
                }
    };
    auto softmax = [&](int k0, bool first) {      // sacc -> pf (online softmax in the log2 domain, deferred rescale)
        if (k0 + 64 > Tk) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= Tk) sacc[sub][r] = -INFINITY;
                }
        }
        float psum = 0.f;
        if constexpr (FOLD) {
            float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (first || !__all(mx <= ATT_THR)) {
                const float m_new = first ? mx : m_run + fmaxf(mx, 0.f);
                const float m_hat = (float)(half_t)m_new;
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
                    for (int r = 0; r < 16; ++r) sacc[sub][r] += delta;
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
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[sub][s][j] = (half_t)sacc[sub][8 * s + j];
    };

    // ---- prologue: K tile 0 ("pair -1") into buffer 1; group B also writes its shares of pair 0 = (K tile 1, V^T tile 0)
    load_pair(0, -1);
    store_pair(0, -1, 1);
    if (grp == 1) {
        load_pair(1, 0);
        store_pair(1, 0, 0);
    }
    load_pair(1 + grp, grp);           // the pair this wave writes in X(0): A pair 0, B pair 1
    barrier();
    if (grp == 1) barrier();           // group B runs one slot behind group A
    qk(1);                             // Q K^T of tile 0
    for (int k = 0; k < nt; ++k) {
        barrier();
        // ---- X(k): VALU slot
        softmax(k * 64, k == 0);
        {
            const int j = k + grp;                         // pair written in this phase (pair j = K tile j+1, V^T tile j)
            if (j < nt) store_pair(j + 1, j, j & 1);
            if (j + 1 < nt) load_pair(j + 2, j + 1);       // ... and the one after it starts its trip from HBM / L2
        }
        barrier();
        // ---- Y(k): matrix slot
        pv(k & 1);
        if (k + 1 < nt) qk(k & 1);                         // K tile k+1 travels in pair k
    }
    if (grp == 0) barrier();

    float l_tot;
    if constexpr (ONES) {
        constexpr int rr = D % 32;
        constexpr int reg = (rr & 3) + 4 * (rr >> 3);
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

// =====================================================================================================================
// Two-strand kernel for long self-attention (round 4).  What bounds the kernels above at d = 40 is the SUM of a wave's matrix and
// vector-ALU time (vector instructions of ANOTHER wave hardly enter a wave's MFMA shadows; instructions of the
// SAME wave do), and 1.4x the algorithmic MFMAs.  Here one wave owns 64 queries as two independent 32-query STRANDS A and B, and its
// instruction stream is written so that every MFMA of one strand is followed by softmax work of the other:
//   * S^T = K Q^T stays on v_mfma_f32_32x32x16_f16 (keys on the rows, d = 40 -> 48 with the scale / running-max fold in slot 40),
//     but O^T += V^T P^T runs on v_mfma_f32_16x16x32_f16: three 16-row tiles cover d = 40 (+ the ones row) instead of two 32-row
//     tiles, 192 instead of 256 matrix cycles per 32 x 64 block.  The fp16 probabilities reach the B-operand layout of the 16-wide
//     shape by FOUR v_permlane16_swap per 32 keys: lane n keeps the key groups of query n, lane n + 16 hands over its own in
//     exchange -- and the K rows of a tile sit in LDS in the order that makes every lane group's eight keys CONTIGUOUS in V^T, so a
//     V^T fragment is one ds_read_b128;
//   * K / V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a ring of 64-key slots, two tiles ahead, one raw s_barrier
//     per tile: no staging registers, no ds_write, no second barrier.  Both images are ROW-major like their sources (a wave
//     instruction fetches whole rows), conflict-free for the fragment reads: K rows of 80 bytes (20 banks: sixteen consecutive rows
//     tile the 64 banks), V^T rows of 128 bytes with their 16-byte pieces XOR-swizzled on the SOURCE side by (row >> 1) & 7;
//     the constant pieces (K[:, 40] = 1, the ones row of V^T, zero padding) live in 48 bytes of LDS that the pad lanes read instead;
//   * NW = 4 waves per workgroup (256 queries), two workgroups per CU: the two waves of a SIMD belong to different workgroups and
//     drift apart instead of meeting at the same barrier; NW = 8: one workgroup of 512 queries per CU, half the L2 -> LDS bytes.
// Needs T % (64 NW) == 0 and Tk % 64 == 0 (the self-attention shapes).
__device__ __forceinline__ void att_dma16(unsigned voff, const void* sbase, unsigned lds_wave_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave_addr) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void att_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ABL: ablation switches for tools/bench_attention.py (FGDM_ATTN_ABL; results are then meaningless): 1 = no exponentials,
// 2 = no maximum / offset decision, 4 = no LDS-DMA after the first two tiles, 16 = no V^T P^T MFMAs, 32 = no K Q^T MFMAs
template <int D, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 2) void attn_dq_kernel(const half_t* __restrict__ Q, int ldq,
                                                         const half_t* __restrict__ K, int ldk,
                                                         const half_t* __restrict__ Vt, int ldvt,
                                                         half_t* __restrict__ O, int ldo,
                                                         int H, int T, int Tk, float sl2e) {
    constexpr int DP = (D + 16) / 16 * 16, NKS = DP / 16;    // contraction length of K Q^T incl. the fold slot; k-steps of 16
    constexpr int DT = DP / 16;                              // 16-row tiles of O^T; row D carries the softmax denominator
    constexpr int DC = D / 8;                                // 16-byte pieces per K row
    constexpr int PS = D / 16, PH = (D % 16) / 8;            // fragment / lane half of contraction slot D
    constexpr int KROW = DC * 16, KBYTES = 64 * KROW, VBYTES = D * 128, SLOT = KBYTES + VBYTES;
    constexpr int NSLOT = 3;
    constexpr int NDMA = SLOT / 1024, NI = (NDMA + NW - 1) / NW;   // 1 KiB LDS-DMA wave-instructions per tile; per wave
    static_assert(KBYTES % 1024 == 0 && VBYTES % 1024 == 0 && NDMA >= NW, "whole wave-instructions per image");
    static_assert((KROW / 4) % 8 == 4, "K row stride: 4 mod 8 dwords so that sixteen consecutive rows tile the banks");
    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * SLOT + 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, h = lane >> 5;                // S^T layout: query on lane & 31, key half on lane >> 5
    const int n16 = lane & 15, g = lane >> 4;                // O^T layout: query on lane & 15, row quad on lane >> 4
    const int nqb = T / (NW * 64);
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qd = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
    }
    const int qblk = logical % nqb, bh = logical / nqb;
    const int b = bh / H, head = bh - b * H;
    const int q0w = qblk * (NW * 64) + wave * 64;                  // first query of this wave; strand X: q0w + 32 X + [0, 32)
    const int nt = Tk >> 6;

    // constant pieces: [1 0 0 0 0 0 0 0] (K[:, D] of the fold), eight ones (row D of V^T), zeros
    char* const cst = smem + NSLOT * SLOT;
    if (tid < 3) {
        h8 v = (h8)(half_t)0;
        if (tid == 0) v[0] = (half_t)1;
        if (tid == 1) v = (h8)(half_t)1;
        *(h8*)(cst + tid * 16) = v;
    }

    // ---- LDS-DMA plan of this wave: wave-instruction j = wave + 4 u of a tile (a surplus one repeats the wave's previous piece:
    // same bytes to the same address).  K piece L = 64 j + lane: LDS row L / DC (row order below), 16-byte piece L % DC;
    // V^T piece L: row L / 8, LDS position L % 8 holds source piece (L % 8) ^ ((row >> 1) & 7).
    // K row order inside a 32-key sub-tile: LDS row rho = 8 a + 4 b + c holds key 16 b + 4 a + c, so that the keys whose
    // probabilities end up in lane group g of the 16-wide B operand (after the permlane16 swaps) are keys 8 g .. 8 g + 7.
    const half_t* Kb = K + (size_t)b * Tk * ldk + head * D;
    const half_t* Vb = Vt + ((size_t)b * H + head) * D * ldvt;
    unsigned voff[NI], ldst[NI];
    bool isk[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        int j = wave + NW * u;
        if (j >= NDMA) j -= NW;
        isk[u] = j < DC * 64 * 16 / 1024;
        ldst[u] = (unsigned)j * 1024u;
        if (isk[u]) {
            const int L = 64 * j + lane, rp = L / DC, c = L - rp * DC;
            const int rho = rp & 31, key = (rp & 32) + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
            voff[u] = (unsigned)(key * ldk + c * 8) * 2u;
        } else {
            const int L = 64 * (j - KBYTES / 1024) + lane, d = L >> 3, x = L & 7;
            voff[u] = (unsigned)(d * ldvt + ((x ^ ((d >> 1) & 7)) * 8)) * 2u;
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    auto issue = [&](int t, auto slotc) {                    // tile t -> slot t % NSLOT
        const char* kb = (const char*)(Kb + (size_t)t * 64 * ldk);
        const char* vb = (const char*)(Vb + t * 64);
        const unsigned base = lds0 + (unsigned)decltype(slotc)::value * SLOT;
#pragma unroll
        for (int u = 0; u < NI; ++u) att_dma16(voff[u], isk[u] ? kb : vb, base + ldst[u]);
    };
    issue(0, std::integral_constant<int, 0>{});
    if (nt > 1) issue(1, std::integral_constant<int, 1>{});

    // ---- Q^T fragments (B operand of K Q^T), pre-scaled; slot D will carry -max
    h8 qf[2][NKS];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int c = 16 * s + 8 * h;
            qf[X][s] = (h8)(half_t)0;
            if (c < D) qf[X][s] = *(const h8*)(Q + ((size_t)b * T + q0w + 32 * X + n32) * ldq + head * D + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[X][s][e] = (half_t)((float)qf[X][s][e] * sl2e);
        }
    f32x4 oacc[2][DT][2];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) oacc[X][t][qg] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {0.f, 0.f};

    // ---- fragment addresses inside a slot.  K: row (32 sub + n32), piece 2 s + h -> immediate offsets 2560 sub + 32 s;
    // pieces >= DC are the constants.  V^T: row 16 t + n16, source piece 4 sub + g at position (4 sub + g) ^ ((n16 >> 1) & 7).
    const int k_lane = n32 * KROW + h * 16;
    const int v_lane0 = KBYTES + n16 * 128 + ((g ^ ((n16 >> 1) & 7)) * 16);
    const int v_lane1 = KBYTES + n16 * 128 + (((4 + g) ^ ((n16 >> 1) & 7)) * 16);
    const char* const c_one = cst, *const c_ones = cst + 16, *const c_zero = cst + 32;

    f32x16 sacc[2][2];
    h8 kf[2][NKS], vf[DT][2], pb[2][2][2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define ATT_SB() __builtin_amdgcn_sched_barrier(0)
    auto rd_k = [&](const char* sl, int sub, int s) {
        const int ci0 = 2 * s;                               // piece index of lane half 0; half 1 reads ci0 + 1
        const char* p = sl + k_lane + sub * 32 * KROW + s * 32;
        if (ci0 + 1 >= DC) {                                 // some lanes read a constant piece
            const char* cp0 = ci0 < DC ? p : (ci0 == DC ? c_one : c_zero);
            const char* cp1 = ci0 + 1 == DC ? c_one : c_zero;
            p = h ? cp1 : cp0;
        }
        kf[sub][s] = *(const h8*)p;
    };
    auto rd_v = [&](const char* sl, int t, int sub) {
        const char* p = sl + (sub ? v_lane1 : v_lane0) + t * 16 * 128;
        if (16 * t + 15 >= D) {                              // rows >= D of the last tile: the ones row, then zeros
            const int d = 16 * t + n16;
            p = d < D ? p : (d == D ? c_ones : c_zero);
        }
        vf[t][sub] = *(const h8*)p;
    };
    auto qk = [&](int X, int sub, int s) {
        if constexpr ((ABL & 32) != 0) { if (s == 0) asm volatile("" : "=v"(sacc[X][sub])); return; }
        sacc[X][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][s], qf[X][s], s == 0 ? zero16 : sacc[X][sub], 0, 0, 0);
    };
    auto pv = [&](int X, int sub, int t, int qg) {
        if constexpr ((ABL & 16) != 0) { asm volatile("" : "+v"(oacc[X][t][qg]) : "v"(vf[t][sub]), "v"(pb[X][sub][qg])); return; }
        oacc[X][t][qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[t][sub], pb[X][sub][qg], oacc[X][t][qg], 0, 0, 0);
    };
    auto max_part = [&](int X, int sub, float mx) {          // 8 v_max3 over one sub-tile's 16 scores of this lane
        if constexpr ((ABL & 2) != 0) return mx;
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[X][sub][r]), sacc[X][sub][r + 1]);
        return mx;
    };
    auto max_cross = [&](float mx) {                         // the other key half of the same query: lane ^ 32
        // inline asm, not __builtin_amdgcn_permlane32_swap: hipcc folds fmaxf(r[0], r[1]) of the builtin's two results into r[0]
        // (visible in the IR; round 3 met the same fold) and the cross-half maximum silently disappears.  s_nop 1 = the two wait
        // states between a vector-ALU write of an operand and the swap that reads it
        float lo = mx, hi = mx;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        return fmaxf(lo, hi);
    };
    // the offset moves only when some query of the strand outgrew it by more than 2^ATT_THR (or on the first tile)
    auto decide = [&](int X, float mx, bool first) {
        if constexpr ((ABL & 2) != 0) return;
        if (first || !__all(mx <= ATT_THR)) {
            const float m_new = first ? mx : m_run[X] + fmaxf(mx, 0.f);
            const float m_hat = (float)(half_t)m_new;        // exactly what Q[q][D] can hold
            const float delta = m_run[X] - m_hat;
            m_run[X] = m_hat;
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(delta);
#pragma unroll
                for (int qg = 0; qg < 2; ++qg) {             // O^T keeps query 16 qg + n16 on this lane: fetch ITS factor
                    const float aq = __shfl(alpha, 16 * qg + n16);
#pragma unroll
                    for (int t = 0; t < DT; ++t) oacc[X][t][qg] *= aq;
                }
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[X][sub][r] += delta;   // this tile was taken against the old offset
            if (h == PH) qf[X][PS][0] = (half_t)(-m_hat);
        }
    };
    auto expo = [&](int X, int sub, int r0, int r1) {
#pragma unroll
        for (int r = r0; r < r1; ++r) sacc[X][sub][r] = (ABL & 1) ? sacc[X][sub][r] : __builtin_amdgcn_exp2f(sacc[X][sub][r]);
    };
    // fp16 probabilities of one 32-key sub-tile -> the two B operands (queries 0-15 / 16-31 of the strand) of the 16-wide MFMA
    auto pack = [&](int X, int sub) {
        unsigned pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const h2 t2 = {(half_t)sacc[X][sub][2 * i], (half_t)sacc[X][sub][2 * i + 1]};
            pk[i] = __builtin_bit_cast(unsigned, t2);
        }
        u32x4 lo, hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto r2 = __builtin_amdgcn_permlane16_swap(pk[i], pk[i + 4], false, false);
            lo[i] = r2[0]; hi[i] = r2[1];
        }
        pb[X][sub][0] = __builtin_bit_cast(h8, lo);
        pb[X][sub][1] = __builtin_bit_cast(h8, hi);
    };

    // one tile; the ring slot is a compile-time constant (the loop below is unrolled over the slots), so every fragment address is
    // a loop-invariant register plus an immediate
    auto tile = [&](int kt, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        // my pieces of tile kt landed (tile kt + 1 may fly), then everybody's: one barrier per tile; behind it nobody reads the
        // slot that tile kt + 2 goes to (the one of tile kt - 1) any more
        if (!(ABL & 4) && kt + 1 < nt) att_wait_vmcnt<NI>(); else att_wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 4) && kt + 2 < nt) issue(kt + 2, std::integral_constant<int, (slot + 2) % NSLOT>{});
        const char* sl = smem + slot * SLOT;
        const bool first = kt == 0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < NKS; ++s) rd_k(sl, sub, s);
        ATT_SB();
        // (b) K Q^T of strand A; the V^T fragments of the tile are requested between its MFMAs
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                qk(0, sub, s);
                if (sub * NKS + s < DT * 2) rd_v(sl, (sub * NKS + s) % DT, (sub * NKS + s) / DT);
                ATT_SB();
            }
        // (c) K Q^T of strand B  |  strand A: maximum, decision, exp and pack of its first 32 keys
        float mx = sacc[0][0][0];                   // (sub-tile 0 first: its MFMAs retired three MFMAs ago)
        qk(1, 0, 0); mx = max_part(0, 0, mx); ATT_SB();
        qk(1, 0, 1); mx = max_part(0, 1, mx); ATT_SB();
        qk(1, 0, 2 % NKS); mx = max_cross(mx); ATT_SB();
#pragma unroll
        for (int s = 3; s < NKS; ++s) { qk(1, 0, s); ATT_SB(); }
        decide(0, mx, first);
        ATT_SB();
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            qk(1, 1, s);
            expo(0, 0, 16 * s / NKS, 16 * (s + 1) / NKS);
            ATT_SB();
        }
        pack(0, 0);
        ATT_SB();
        // (d) V^T P^T of strand A, first 32 keys  |  strand A: exp of its second 32 keys; strand B: maximum
        float mxb = sacc[1][0][0];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) {
                const int i = t * 2 + qg, n = DT * 2;
                pv(0, 0, t, qg);
                expo(0, 1, 16 * i / n, 16 * (i + 1) / n);
                if (i == 1) mxb = max_part(1, 0, mxb);
                if (i == 3) mxb = max_part(1, 1, mxb);
                if (i == 4) mxb = max_cross(mxb);
                ATT_SB();
            }
        pack(0, 1);
        ATT_SB();
        decide(1, mxb, first);
        ATT_SB();
        // (e) V^T P^T of strand A, second 32 keys  |  strand B: exp and pack of its first 32 keys
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) {
                const int i = t * 2 + qg, n = DT * 2;
                pv(0, 1, t, qg);
                expo(1, 0, 16 * i / n, 16 * (i + 1) / n);
                ATT_SB();
            }
        pack(1, 0);
        ATT_SB();
        // (f) V^T P^T of strand B, first 32 keys  |  strand B: exp and pack of its second 32 keys
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) {
                const int i = t * 2 + qg, n = DT * 2;
                pv(1, 0, t, qg);
                expo(1, 1, 16 * i / n, 16 * (i + 1) / n);
                ATT_SB();
            }
        pack(1, 1);
        ATT_SB();
        // (g) V^T P^T of strand B, second 32 keys
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) pv(1, 1, t, qg);
        ATT_SB();
    };
    for (int kt = 0; kt < nt; kt += NSLOT) {
        tile(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nt) tile(kt + 1, std::integral_constant<int, 1>{});
        if (kt + 2 < nt) tile(kt + 2, std::integral_constant<int, 2>{});
    }
#undef ATT_SB

    // ---- O = O^T / l: row D of O^T (tile D / 16, row D % 16: lane group (D % 16) / 4, register D % 4) is the denominator
    constexpr int LT = D / 16, LG = (D % 16) / 4, LR = D % 4;
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) {
            const float l = __shfl(oacc[X][LT][qg][LR], 16 * LG + n16);
            const float inv = 1.0f / l;
            half_t* op = O + ((size_t)b * T + q0w + 32 * X + 16 * qg + n16) * ldo + head * D;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int dd = 16 * t + 4 * g;
                if (dd < D) {
                    const h4 o4 = {(half_t)(oacc[X][t][qg][0] * inv), (half_t)(oacc[X][t][qg][1] * inv),
                                   (half_t)(oacc[X][t][qg][2] * inv), (half_t)(oacc[X][t][qg][3] * inv)};
                    *(h4*)(op + dd) = o4;
                }
            }
        }
}

// The same two-strand structure with V^T P^T back on v_mfma_f32_32x32x16_f16 (round 4, after the ablations of the kernel above:
// its time is the SUM of its vector-ALU issue cycles and 8 issue cycles per MFMA -- the matrix pipe's own time hides in the
// shadows -- so what counts is the NUMBER of MFMAs and of vector instructions, not the matrix cycles): 8 MFMAs per 32 x 64 block
// instead of 12, no permlane swaps (the S^T accumulators, converted, ARE the B operand), O^T as two 32-row tiles (d = 40 -> 64,
// rows 41.. zero).  K rows sit in LDS in the order that makes the eight keys of a lane's B-operand slice contiguous in V^T, so a
// V^T fragment is again one ds_read_b128.  The constant rows of V^T (ones row, zero rows) are part of the LDS image (written once
// per slot; the DMA fills rows 0 .. D - 1 only), so only the K fragment of the fold slot needs a per-lane address.
template <int D, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, D <= 40 ? 2 : 1) void attn_dq32_kernel(const half_t* __restrict__ Q, int ldq,
                                                           const half_t* __restrict__ K, int ldk,
                                                           const half_t* __restrict__ Vt, int ldvt,
                                                           half_t* __restrict__ O, int ldo,
                                                           int H, int T, int Tk, float sl2e) {
    constexpr int DP = (D + 16) / 16 * 16, NKS = DP / 16;    // contraction length of K Q^T incl. the fold slot; k-steps of 16
    constexpr int DT = (D + 32) / 32;                        // 32-row tiles of O^T; row D carries the softmax denominator
    constexpr int DC = D / 8;                                // 16-byte pieces per K row
    constexpr int PS = D / 16, PH = (D % 16) / 8;            // fragment / lane half of contraction slot D
    constexpr int KROW = DC * 16, KBYTES = 64 * KROW, VDATA = D * 128, VBYTES = DT * 32 * 128, SLOT = KBYTES + VBYTES;
    constexpr int NSLOT = 3;
    constexpr int NDMA = (KBYTES + VDATA) / 1024, NI = (NDMA + NW - 1) / NW;   // 1 KiB LDS-DMA wave-instructions per tile; per wave
    static_assert(KBYTES % 1024 == 0 && VDATA % 1024 == 0 && NDMA >= NW, "whole wave-instructions per image");
    // K row stride: 4 mod 8 dwords makes sixteen consecutive rows tile the banks (d = 40: 80-byte rows).  d = 80 (160-byte rows: rows
    // r and r + 8 would meet in the same banks): the 16-byte pieces of rows with (r >> 3) & 1 are ROTATED by ROT = 5 positions,
    // half a row, on the source side -- sixteen consecutive rows then tile the banks again
    constexpr int ROT = (KROW / 4) % 8 == 4 ? 0 : DC / 2;
    static_assert(ROT == 0 || ((KROW / 4) % 16 == 8 && DC % 2 == 0), "K row stride");
    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * SLOT + 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, h = lane >> 5;                // query (S^T, O^T) / row (fragments) on lane & 31, half on lane >> 5
    const int nqb = T / (NW * 64);
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qd = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
    }
    const int qblk = logical % nqb, bh = logical / nqb;
    const int b = bh / H, head = bh - b * H;
    const int q0w = qblk * (NW * 64) + wave * 64;            // first query of this wave; strand X: q0w + 32 X + [0, 32)
    const int nt = Tk >> 6;

    // constant parts of the LDS image: rows D .. 32 DT - 1 of every slot's V^T (row D = ones), and [1 0 0 0 0 0 0 0] / zeros for
    // the fold slot of K
    char* const cst = smem + NSLOT * SLOT;
    if (tid < 2) {
        h8 v = (h8)(half_t)0;
        if (tid == 0) v[0] = (half_t)1;
        *(h8*)(cst + tid * 16) = v;
    }
    for (int i = tid; i < NSLOT * (DT * 32 - D) * 8; i += NW * 64) {
        const int sl_i = i / ((DT * 32 - D) * 8), rem = i - sl_i * ((DT * 32 - D) * 8), row = D + rem / 8, pc = rem & 7;
        *(h8*)(smem + sl_i * SLOT + KBYTES + row * 128 + pc * 16) = row == D ? (h8)(half_t)1 : (h8)(half_t)0;
    }

    // ---- LDS-DMA plan of this wave (see the kernel above).  K row order inside a 32-key sub-tile: LDS row rho = 8 a + 4 b + c
    // holds key 16 (a >> 1) + 8 b + 4 (a & 1) + c: lane half b of the B operand of k-step s2 = a >> 1 holds the probabilities of
    // LDS rows {8 (2 s2) + 4 b + c, 8 (2 s2 + 1) + 4 b + c}, i.e. of keys 16 s2 + 8 b + [0, 8)
    const half_t* Kb = K + (size_t)b * Tk * ldk + head * D;
    const half_t* Vb = Vt + ((size_t)b * H + head) * D * ldvt;
    unsigned voff[NI], ldst[NI];
    bool isk[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        int j = wave + NW * u;
        if (j >= NDMA) j -= NW;
        isk[u] = j < KBYTES / 1024;
        ldst[u] = (unsigned)j * 1024u;
        if (isk[u]) {
            const int L = 64 * j + lane, rp = L / DC, c = L - rp * DC;
            const int rho = rp & 31, a = rho >> 3;
            const int key = (rp & 32) + 16 * (a >> 1) + 8 * ((rho >> 2) & 1) + 4 * (a & 1) + (rho & 3);
            const int cs = (c + DC - ROT * ((rho >> 3) & 1)) % DC;           // LDS position c of this row holds source piece cs
            voff[u] = (unsigned)(key * ldk + cs * 8) * 2u;
        } else {
            const int L = 64 * (j - KBYTES / 1024) + lane, d = L >> 3, x = L & 7;
            voff[u] = (unsigned)(d * ldvt + ((x ^ ((d >> 1) & 7)) * 8)) * 2u;
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    auto issue = [&](int t, auto slotc) {                    // tile t -> slot t % NSLOT
        const char* kb = (const char*)(Kb + (size_t)t * 64 * ldk);
        const char* vb = (const char*)(Vb + t * 64);
        const unsigned base = lds0 + (unsigned)decltype(slotc)::value * SLOT;
#pragma unroll
        for (int u = 0; u < NI; ++u) att_dma16(voff[u], isk[u] ? kb : vb, base + ldst[u]);
    };
    issue(0, std::integral_constant<int, 0>{});
    if (nt > 1) issue(1, std::integral_constant<int, 1>{});

    // ---- Q^T fragments (B operand of K Q^T), pre-scaled; slot D will carry -max
    h8 qf[2][NKS];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int c = 16 * s + 8 * h;
            qf[X][s] = (h8)(half_t)0;
            if (c < D) qf[X][s] = *(const h8*)(Q + ((size_t)b * T + q0w + 32 * X + n32) * ldq + head * D + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[X][s][e] = (half_t)((float)qf[X][s][e] * sl2e);
        }
    f32x16 oacc[2][DT];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[X][t][r] = 0.f;
    float m_run[2] = {0.f, 0.f};

    // ---- fragment addresses inside a slot.  K: row (32 sub + n32), piece 2 s + h -> immediate offsets 32 KROW sub + 32 s;
    // pieces >= DC are the constants.  V^T: row 32 t + n32, source piece 4 sub + 2 s2 + h at position (that) ^ ((n32 >> 1) & 7).
    int k_lane[NKS];                                         // byte offset of piece 2 s + h in this lane's row (rotated rows: see ROT)
#pragma unroll
    for (int s2 = 0; s2 < NKS; ++s2) k_lane[s2] = n32 * KROW + ((2 * s2 + h + ROT * ((n32 >> 3) & 1)) % DC) * 16;
    int v_lane[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) v_lane[sub][s2] = KBYTES + n32 * 128 + (((4 * sub + 2 * s2 + h) ^ ((n32 >> 1) & 7)) * 16);
    const char* const c_one = cst, *const c_zero = cst + 16;

    f32x16 sacc[2][2];
    h8 kf[2][NKS], vf[DT][2][2], pf[2][2][2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define ATT_SB() __builtin_amdgcn_sched_barrier(0)
    auto rd_k = [&](const char* sl, int sub, int s) {
        const int ci0 = 2 * s;                               // piece index of lane half 0; half 1 reads ci0 + 1
        const char* p = sl + k_lane[s] + sub * 32 * KROW;
        if (ci0 + 1 >= DC) {                                 // some lanes read a constant piece
            const char* cp0 = ci0 < DC ? p : (ci0 == DC ? c_one : c_zero);
            const char* cp1 = ci0 + 1 == DC ? c_one : c_zero;
            p = h ? cp1 : cp0;
        }
        kf[sub][s] = *(const h8*)p;
    };
    auto rd_v = [&](const char* sl, int t, int sub, int s2) { vf[t][sub][s2] = *(const h8*)(sl + v_lane[sub][s2] + t * 32 * 128); };
    auto qk = [&](int X, int sub, int s) {
        if constexpr ((ABL & 32) != 0) { if (s == 0) asm volatile("" : "=v"(sacc[X][sub])); return; }
        sacc[X][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][s], qf[X][s], s == 0 ? zero16 : sacc[X][sub], 0, 0, 0);
    };
    auto pv = [&](int X, int sub, int t, int s2) {
        if constexpr ((ABL & 16) != 0) { asm volatile("" : "+v"(oacc[X][t]) : "v"(vf[t][sub][s2]), "v"(pf[X][sub][s2])); return; }
        oacc[X][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t][sub][s2], pf[X][sub][s2], oacc[X][t], 0, 0, 0);
    };
    auto max_part = [&](int X, int sub, float mx) {          // 8 v_max3 over one sub-tile's 16 scores of this lane
        if constexpr ((ABL & 2) != 0) return mx;
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[X][sub][r]), sacc[X][sub][r + 1]);
        return mx;
    };
    auto max_cross = [&](float mx) {                         // the other key half of the same query: lane ^ 32 (asm: see above)
        float lo = mx, hi = mx;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        return fmaxf(lo, hi);
    };
    auto decide = [&](int X, float mx, bool first) {
        if constexpr ((ABL & 2) != 0) return;
        if (first || !__all(mx <= ATT_THR)) {
            const float m_new = first ? mx : m_run[X] + fmaxf(mx, 0.f);
            const float m_hat = (float)(half_t)m_new;        // exactly what Q[q][D] can hold
            const float delta = m_run[X] - m_hat;
            m_run[X] = m_hat;
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(delta);
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[X][t][r] *= alpha;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[X][sub][r] += delta;   // this tile was taken against the old offset
            if (h == PH) qf[X][PS][0] = (half_t)(-m_hat);
        }
    };
    auto expo = [&](int X, int sub, int r0, int r1) {
#pragma unroll
        for (int r = r0; r < r1; ++r) sacc[X][sub][r] = (ABL & 1) ? sacc[X][sub][r] : __builtin_amdgcn_exp2f(sacc[X][sub][r]);
    };
    auto pack = [&](int X, int sub) {                        // fp16 probabilities: registers 8 s2 .. 8 s2 + 7 are k-step s2's B operand
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[X][sub][s2][e] = (half_t)sacc[X][sub][8 * s2 + e];
    };
    constexpr int NPV = DT * 2;                              // V^T P^T MFMAs per 32-key sub-tile and strand
    auto tile = [&](int kt, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        if (!(ABL & 4) && kt + 1 < nt) att_wait_vmcnt<NI>(); else att_wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 4) && kt + 2 < nt) issue(kt + 2, std::integral_constant<int, (slot + 2) % NSLOT>{});
        const char* sl = smem + slot * SLOT;
        const bool first = kt == 0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < NKS; ++s) rd_k(sl, sub, s);
        ATT_SB();
        // (b) K Q^T of strand A; the V^T fragments of the first 32 keys are requested between its MFMAs
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const int i = sub * NKS + s;
                qk(0, sub, s);
                if (i < NPV) rd_v(sl, i / 2, 0, i & 1);
                ATT_SB();
            }
        // (c) K Q^T of strand B  |  strand A: maximum, decision, exp and pack of its first 32 keys
        float mx = sacc[0][0][0];                            // (sub-tile 0 first: its MFMAs retired three MFMAs ago)
        qk(1, 0, 0); mx = max_part(0, 0, mx); ATT_SB();
        qk(1, 0, 1); mx = max_part(0, 1, mx); ATT_SB();
        qk(1, 0, 2 % NKS); mx = max_cross(mx); ATT_SB();
#pragma unroll
        for (int s = 3; s < NKS; ++s) { qk(1, 0, s); ATT_SB(); }
        decide(0, mx, first);
        ATT_SB();
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            qk(1, 1, s);
            expo(0, 0, 16 * s / NKS, 16 * (s + 1) / NKS);
            if (s < NPV) rd_v(sl, s / 2, 1, s & 1);          // ... and the V^T fragments of the second 32 keys
            ATT_SB();
        }
#pragma unroll
        for (int i = NKS; i < NPV; ++i) rd_v(sl, i / 2, 1, i & 1);
        pack(0, 0);
        ATT_SB();
        // (d) V^T P^T of strand A, first 32 keys  |  strand A: exp of its second 32 keys; strand B: maximum
        float mxb = sacc[1][0][0];
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            pv(0, 0, i / 2, i & 1);
            expo(0, 1, 16 * i / NPV, 16 * (i + 1) / NPV);
            if (i == NPV - 3) mxb = max_part(1, 0, mxb);
            if (i == NPV - 2) mxb = max_part(1, 1, mxb);
            if (i == NPV - 1) mxb = max_cross(mxb);
            ATT_SB();
        }
        pack(0, 1);
        ATT_SB();
        decide(1, mxb, first);
        ATT_SB();
        // (e) V^T P^T of strand A, second 32 keys  |  strand B: exp and pack of its first 32 keys
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            pv(0, 1, i / 2, i & 1);
            expo(1, 0, 16 * i / NPV, 16 * (i + 1) / NPV);
            ATT_SB();
        }
        pack(1, 0);
        ATT_SB();
        // (f) V^T P^T of strand B, first 32 keys  |  strand B: exp and pack of its second 32 keys
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            pv(1, 0, i / 2, i & 1);
            expo(1, 1, 16 * i / NPV, 16 * (i + 1) / NPV);
            ATT_SB();
        }
        pack(1, 1);
        ATT_SB();
        // (g) V^T P^T of strand B, second 32 keys
#pragma unroll
        for (int i = 0; i < NPV; ++i) pv(1, 1, i / 2, i & 1);
        ATT_SB();
    };
    for (int kt = 0; kt < nt; kt += NSLOT) {
        tile(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nt) tile(kt + 1, std::integral_constant<int, 1>{});
        if (kt + 2 < nt) tile(kt + 2, std::integral_constant<int, 2>{});
    }
#undef ATT_SB

    // ---- O = O^T / l: row D of O^T lives in tile D / 32, register ((D % 32) & 3) + 4 ((D % 32) >> 3) of the half with 4 h == (D % 32) & 4
    constexpr int rr = D % 32, reg = (rr & 3) + 4 * (rr >> 3), owner_half = (rr >> 2) & 1;
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const float mine = oacc[X][D / 32][reg];
        const float other = __shfl_xor(mine, 32);
        const float inv = 1.0f / (h == owner_half ? mine : other);
        half_t* op = O + ((size_t)b * T + q0w + 32 * X + n32) * ldo + head * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = t * 32 + 8 * g4 + 4 * h;
                if (dd < D) {
                    const h4 o4 = {(half_t)(oacc[X][t][4 * g4] * inv), (half_t)(oacc[X][t][4 * g4 + 1] * inv),
                                   (half_t)(oacc[X][t][4 * g4 + 2] * inv), (half_t)(oacc[X][t][4 * g4 + 3] * inv)};
                    *(h4*)(op + dd) = o4;
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
    // long self-attention: the eight-wave ping-pong kernel (256 queries per workgroup); FGDM_ATTN_PP=0 switches it off (A/B)
    static const bool pp_on = !(getenv("FGDM_ATTN_PP") && atoi(getenv("FGDM_ATTN_PP")) == 0);
    // ... and the two-strand kernel where its shape conditions hold (FGDM_ATTN_DQ: 0 = off, 1 = plain, 2 = rotated tail)
    // ... and the two-strand kernels where their shape conditions hold.  FGDM_ATTN_DQ: 0 = off, 1 = 16-wide V^T P^T (the faster one
    // on random operands: 756 vs 733 TF/s at B32 T4096), 3 = 32-wide V^T P^T (default: the faster one inside the network, where the
    // activations toggle less and the chip holds its clock: attention family 357 vs 362 ms per sampling pass, 421 before)
    static const int dq = getenv("FGDM_ATTN_DQ") ? atoi(getenv("FGDM_ATTN_DQ")) : 3;
    if (dq > 0 && d == 40 && T % 256 == 0 && Tk % 64 == 0 && Tk >= 128 && (size_t)64 * ldk * 2 < (1u << 31) &&
        (size_t)d * ldvt * 2 < (1u << 31)) {
        const dim3 gridq((T / 256) * H * B), blockq(256);
        static const int abl = getenv("FGDM_ATTN_ABL") ? atoi(getenv("FGDM_ATTN_ABL")) : 0;     // tools/bench_attention.py only
        if (abl) {
            switch (abl + (dq == 3 ? 1000 : 0)) {
#define ATT_ABL_CASE(v) case v: FGDM_LAUNCH((attn_dq_kernel<40, 4, v>), gridq, blockq, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break; \
                        case 1000 + v: FGDM_LAUNCH((attn_dq32_kernel<40, 4, v>), gridq, blockq, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
                ATT_ABL_CASE(1) ATT_ABL_CASE(2) ATT_ABL_CASE(3) ATT_ABL_CASE(4) ATT_ABL_CASE(16) ATT_ABL_CASE(32) ATT_ABL_CASE(48) ATT_ABL_CASE(51) ATT_ABL_CASE(7)
#undef ATT_ABL_CASE
                default: return FGDM_ERR_ARG;
            }
            return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
        }
        if (dq == 1) FGDM_LAUNCH((attn_dq_kernel<40, 4>), gridq, blockq, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e);
        else FGDM_LAUNCH((attn_dq32_kernel<40, 4>), gridq, blockq, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e);
        return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
    }
    // d = 80: the same kernel at one wave per SIMD (its two strands need ~280 registers); FGDM_ATTN_DQ80=0: the ping-pong kernel
    static const int dq80 = getenv("FGDM_ATTN_DQ80") ? atoi(getenv("FGDM_ATTN_DQ80")) : 1;
    if (dq80 > 0 && d == 80 && T % 256 == 0 && Tk % 64 == 0 && Tk >= 128 && (size_t)64 * ldk * 2 < (1u << 31) &&
        (size_t)d * ldvt * 2 < (1u << 31)) {
        const dim3 gridq((T / 256) * H * B), blockq(256);
        FGDM_LAUNCH((attn_dq32_kernel<80, 4>), gridq, blockq, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e);
        return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
    }
    if (pp_on && T >= 256 && Tk >= 256 && (d == 40 || d == 80)) {
        const dim3 grid2(((T + 255) / 256) * H * B), block2(512);
        if (d == 40) FGDM_LAUNCH(attn_pp_kernel<40>, grid2, block2, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e);
        else FGDM_LAUNCH(attn_pp_kernel<80>, grid2, block2, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e);
        return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
    }
    // the text tokens: all keys staged once per workgroup, several query chunks per wave; FGDM_ATTN_CROSS=0 switches it off (A/B)
    static const int cross = getenv("FGDM_ATTN_CROSS") ? atoi(getenv("FGDM_ATTN_CROSS")) : 4;
    if (cross > 0 && Tk > 64 && Tk <= 96 && ldvt >= 96 && T >= 128) {
        const int cpw = std::min(cross, T / 128);
        const dim3 gridc(((T + 128 * cpw - 1) / (128 * cpw)) * H * B), blockc(256);
        switch (d) {
            case 40: FGDM_LAUNCH(attn_cross_kernel<40>, gridc, blockc, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e, cpw); break;
            case 80: FGDM_LAUNCH(attn_cross_kernel<80>, gridc, blockc, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e, cpw); break;
            case 160: FGDM_LAUNCH(attn_cross_kernel<160>, gridc, blockc, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e, cpw); break;
            default: return FGDM_ERR_ARG;
        }
        return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
    }
    const dim3 grid(((T + 127) / 128) * H * B), block(256);
    switch (d) {
        case 40: FGDM_LAUNCH(attn_kernel<40>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        case 80: FGDM_LAUNCH(attn_kernel<80>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        case 160: FGDM_LAUNCH(attn_kernel<160>, grid, block, 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, H, T, Tk, sl2e); break;
        default: return FGDM_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}
