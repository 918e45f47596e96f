// Pipelined implicit-GEMM kernel for the large layers (second-generation of igemm.hip; same IgemmArgs contract).
//
// What changed against igemm_kernel, and why (measured on MI355X, profiles/r01_*):
//   * one 512-thread workgroup per CU owns a 256x320 / 256x256 / 128x320 output tile (N = 320 k for every SD
//     layer, so no column waste) -> 2x the flops per byte staged into LDS;
//   * BK = 32, FOUR-stage LDS ring filled by global_load_lds_dwordx4, waited with a COUNTED s_waitcnt vmcnt so
//     two stages stay in flight across the (single, raw) s_barrier of each K-step -- the old kernel drained to
//     vmcnt(0) every step and ran latency-bound at 16 % of the MFMA peak;
//   * accumulators are kept TRANSPOSED (rows = output channels, lane = pixel), so a lane owns 4 consecutive
//     channels per register quad: the tile is bounced through wave-private LDS as fp16 and leaves as 16-byte
//     row-contiguous stores with a 16-byte residual read (the old 2-byte stores cost ~135 us per 84 MB tensor).
#include "common.h"
#include <algorithm>
#include <type_traits>

#pragma clang diagnostic ignored "-Winline-asm"
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

namespace {

// global -> LDS DMA, 16 bytes per lane, issued through inline asm.  hipcc's wait-count pass treats the builtin form
// (__builtin_amdgcn_global_load_lds) as a FLAT access to both address spaces ("pending flat"), after which EVERY
// lgkmcnt / vmcnt wait it inserts is a full drain to 0 -- with loads permanently in flight that serialised each
// stage's 14 fragment reads against its MFMAs.  The asm form is invisible to the pass: completion is tracked by the
// hand-counted s_waitcnt vmcnt(N) in the K loop.  M0 = wave-uniform LDS byte address; saved / restored around the op.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_wave_addr)
                 : "memory", "m0");
}
// the same for the pipelined K loop, whose instruction stream is what bounds it: M0 is written and not restored, which the "m0"
// clobber tells the compiler (its own M0 users -- indexed register moves, sendmsg, the LDS-DMA builtin -- re-initialise M0 behind
// such a statement instead of assuming it survived; LDS instructions take no M0 on gfx9+), and the weight pieces use the
// SGPR-base + 32-bit VGPR offset form.  (-Winline-asm: M0 is a reserved register; naming it as a clobber is the point.)
__device__ __forceinline__ void glds16_m0(const void* g, unsigned lds_wave_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_addr) : "memory", "m0");
}
__device__ __forceinline__ void glds16_sv(unsigned voff, const void* sbase, unsigned lds_wave_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(LDS_AS const void*)p; }
// rstd (acc - mean u): one fma and one multiply as inline asm -- with -ffp-contract=fast the backend fuses a multiply into the
// bias add that follows whatever the source says; the 2-stage kernel (igemm.hip) rounds in exactly this sequence, and WHICH
// kernel evaluates a layer must not change a bit of its output
__device__ __forceinline__ float ln_scale(float acc, float mean, float rstd, float u) {
    float t, w;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(-mean), "v"(u), "v"(acc));
    asm("v_mul_f32 %0, %1, %2" : "=v"(w) : "v"(rstd), "v"(t));
    return w;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int BK = 32;          // halfs per K-step: 64-byte LDS rows, 4 x 16-byte chunks
constexpr int ROWB = BK * 2;

constexpr int RV_MAX = 6;       // per-sample emb rows staged in LDS per tile (more samples per tile: global loads)

// PIPE = 2 (round 4): the 3x3 taps from an LDS-resident HALO tile.  Pixels a halo buffer holds (a multiple of 16 = one 1 KiB
// LDS-DMA piece): (BM / W + 2) rows of (W + 2) pixels for W = 16 / 32 / 64 -- 324 / 340 / 396 for a 256-row tile, 180 / 204 / 264
// for a 128-row tile -- and the LDS image of the K loop: two halo buffers, then the four-stage ring of W stages.
template <int BM> constexpr int halo_px() { return BM == 256 ? 400 : 272; }
template <int BM, int BN, int STAGES, int PIPE> constexpr int ring_bytes() {
    return PIPE == 2 ? 2 * halo_px<BM>() * ROWB + STAGES * BN * ROWB : STAGES * (BM + BN) * ROWB;
}

// MS = MFMA tile edge: 16 -> v_mfma_f32_16x16x32_f16 (one instruction per BK = 32 step and 16x16 tile), 32 ->
// v_mfma_f32_32x32x16_f16 (two k-substeps per 32x32 tile).  Same flops, same LDS bytes, same accumulator registers; the
// chip holds a higher clock under the 16x16x32 shape (this kernel is power-limited: all-zero operands run 1.3x faster
// than random ones), measured +3..9 % on the conv shapes (tools/bench_igemm.py cfg 4 vs 7).
// LN: the consumer side of a folded LayerNorm (IgemmArgs::ln_stats) is its own instantiation, so every other layer pays nothing
// for it (as a run-time branch inside the register stage it cost the 256x320 tile 208 bytes of scratch per lane)
// PIPE: 0 = every K-step reads its 14 fragments, then runs its 40 MFMAs (all eight waves in the same phase); 1 = software-pipelined
// K loop (see there).  ABL: the ablation switches of IgemmArgs::debug exist only in the instantiations tools/bench_igemm.py asks for.
template <int BM, int BN, int WM, int WN, int STAGES, bool CONV, bool GEGLU, bool SPLIT, int MS, bool LN, int PATH, int PIPE, bool ABL>
__device__ __forceinline__ void igemm2_body(const IgemmArgs& a) {
    constexpr int NW = WM * WN, T = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / MS, NI = TN / MS;
    constexpr int KS = MS == 32 ? 2 : 1;              // MFMA k-substeps per BK = 32 stage
    constexpr int AR = MS == 32 ? 16 : 4;             // accumulator registers per tile
    typedef float acc_t __attribute__((ext_vector_type(AR)));
    // XOR swizzle of the four 16-byte chunks of a 64-byte LDS row, chosen per MFMA shape so that every 16-lane group
    // of a ds_read_b128 fragment read covers all 64 banks (lane groups: MI355X_MICROARCH.md, LDS)
    auto swz_of = [](int r) { return MS == 32 ? ((r >> 2) & 3) : (((r >> 3) & 1) << 1); };
    constexpr int A_INSTR = BM * 4 / 64, B_INSTR = BN * 4 / 64;     // wave-instructions (1 KiB each) per stage
    static_assert(A_INSTR % NW == 0, "A tile must split evenly over the waves");
    constexpr int LA = A_INSTR / NW, LB = (B_INSTR + NW - 1) / NW;  // A / B load slots per wave per stage
    constexpr int LPS = LA + LB;                                    // loads per stage per wave (uniform)
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int A_BYTES = BM * ROWB;
    constexpr int RING_BYTES = ring_bytes<BM, BN, STAGES, PIPE>();     // after the ring: bias[BN] then emb rows [RV_MAX][BN], fp32
    static_assert(TM % 32 == 0 && TN % MS == 0, "wave tile: 32-pixel epilogue passes, MFMA-sized channel tiles");
    static_assert(!GEGLU || TN % 64 == 0, "GEGLU: whole 64-row [32 value | 32 gate] groups per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int ntn = (a.N + BN - 1) / BN;
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nsplit = SPLIT ? a.splitk : 1;          // split-K lives in its own instantiation (register budget)
    // split-K (the 8x8 level: M = 2048, weights 30-60 MB, activations 5 MB): ROW tiles fastest, so that the workgroups of one
    // XCD (a run of consecutive logical ids) share few (column tile, K slice) pairs and that slice of W stays in the XCD's L2
    // (measured: 1280->1280 94 -> 93 us, 2560->1280 182 -> 171 us)
    int split = 0, m0, n0;
    if (SPLIT) {
        const int ntm = (a.M + BM - 1) / BM;
        const int row = logical % ntm, pair = logical / ntm;
        split = pair % nsplit; n0 = (pair / nsplit) * BN; m0 = row * BM;
    } else {
        m0 = (logical / ntn) * BM; n0 = (logical % ntn) * BN;
    }

    const int nk_all = a.K >> 5;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int k_begin = split * per;                       // this workgroup's K-step range [k_begin, k_begin + nk)
    const int nk = max(0, min(per, nk_all - k_begin));

    // ---- this wave's load slots.  A slot i covers wave-instruction (wave + i*NW) of the A tile (16 rows x 64 B);
    // B slot likewise; a surplus B slot repeats the last instruction (same bytes to the same LDS address: harmless)
    int a_pix[LA], a_yx[LA], a_off[LA];      // pixel base, (oy,ox), source chunk offset (halfs); pix = -1: row >= M
    int b_off[LB], b_lds[LB];
    const int sy = (CONV && a.mode == IG_CONV3_S2) ? 2 : 1;
    const int up = (CONV && a.mode == IG_CONV3_UP2) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int q = (wave + i * NW) * 64 + lane, r = q >> 2, s = q & 3;
        a_off[i] = (s ^ swz_of(r)) << 3;                            // source-side swizzle (LDS image stays linear)
        a_pix[i] = -1; a_yx[i] = 0;
        const int m = m0 + r;
        if (PIPE == 0 && m < a.M) {
            if (!CONV) {
                a_pix[i] = m;
            } else {
                const int hw = a.Ho * a.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                a_pix[i] = b * a.H * a.W;
                a_yx[i] = ((oy * sy) << 16) | (ox * sy);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int idx = wave + i * NW;
        if (idx >= B_INSTR) idx = B_INSTR - 1;
        const int q = idx * 64 + lane, r = q >> 2, s = q & 3;
        b_off[i] = (n0 + r) * a.K + ((s ^ swz_of(r)) << 3);
        b_lds[i] = A_BYTES + idx * 1024;
    }
    const unsigned Hu = (unsigned)(a.H << up), Wu = (unsigned)(a.W << up);
    // ... and the form the pipelined loop uses (the launcher keeps tensors >= 4 GB on the loop above): byte
    // offset of the slot's centre pixel in either source, and one validity bit per tap -- a piece's address is then
    // (source + tap / channel displacement: scalar) + offset, or the zero page
    unsigned a_b0[LA], a_b1[LA], a_msk[LA], b_b[LB];
    if constexpr (PIPE == 2) {
#pragma unroll
        for (int i = 0; i < LB; ++i) b_b[i] = (unsigned)b_off[i] * 2u;
    }
    if constexpr (PIPE == 1) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            a_b0[i] = a_b1[i] = a_msk[i] = 0;
            const int q = (wave + i * NW) * 64 + lane, r = q >> 2;
            const int m = m0 + r;
            if (m < a.M) {
                unsigned cen = (unsigned)m;
                a_msk[i] = 1;
                if (CONV) {
                    const int hw = a.Ho * a.Wo;
                    const int b = m / hw, rem = m - b * hw;
                    const int oy = (rem / a.Wo) * sy, ox = (rem - (rem / a.Wo) * a.Wo) * sy;
                    // nearest-2x upsampled input (up = 1): (oy, ox) are coordinates of the upsampled image, the centre is the
                    // source pixel (oy >> 1, ox >> 1), and a tap's source displacement depends on the parity of oy / ox:
                    // bits 16 / 17 of the mask say "oy even" / "ox even"
                    cen = (unsigned)(b * a.H * a.W + (oy >> up) * a.W + (ox >> up));
                    unsigned msk = 0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int uy = oy + t / 3 - 1, ux = ox + t % 3 - 1;
                        if ((unsigned)uy < Hu && (unsigned)ux < Wu) msk |= 1u << t;
                    }
                    if (up) msk |= ((oy & 1) ? 0u : 1u << 16) | ((ox & 1) ? 0u : 1u << 17);
                    a_msk[i] = msk;
                }
                a_b0[i] = (cen * (unsigned)a.C0 + (unsigned)a_off[i]) * 2u;
                a_b1[i] = (cen * (unsigned)a.C1 + (unsigned)a_off[i]) * 2u;
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) b_b[i] = (unsigned)b_off[i] * 2u;
    }

    // K-step order (CONV): 64-channel chunk outermost, then the 9 taps, then the two 32-channel halves -- so the nine
    // shifted re-reads of a (pixel, channel-chunk) happen in consecutive steps and hit L1/L2 instead of travelling
    // to the Infinity Cache (the LDS fill path, ~70 GB/s per CU from L2 vs ~33 from MALL, is what bounds this
    // kernel).  `cc` = channel offset of the step, `tap` = 0..8.  Weights are packed in the same order.
    // piece p of a stage: p < LA = this wave's A slot p, else its B slot p - LA (one 1 KiB LDS-DMA wave-instruction each)
    auto stage_piece = [&](int kt, int tap, int cc, int buf, int p) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(smem) + buf * STAGE_BYTES);
        if (p < LA) {
            const int i = p;
            const half_t* src = a.A0;
            int Cs = a.C0, co = cc;
            if (co >= a.C0) { src = a.A1; Cs = a.C1; co -= a.C0; }
            const int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
            const half_t* ptr = a.zero;
            if (!CONV) {
                if (a_pix[i] >= 0) ptr = src + (size_t)a_pix[i] * Cs + co + a_off[i];
            } else {
                // one formula for stride 1 / stride 2 / conv over the nearest-2x-upsampled image
                const int uy = (a_yx[i] >> 16) + ky, ux = (a_yx[i] & 0xffff) + kx;
                if (a_pix[i] >= 0 && (unsigned)uy < Hu && (unsigned)ux < Wu)
                    ptr = src + (size_t)(a_pix[i] + (uy >> up) * a.W + (ux >> up)) * Cs + co + a_off[i];
            }
            if constexpr (ABL) { if (a.debug & 8) ptr = a.zero; }     // ablation: same instruction stream, no memory footprint
            glds16(ptr, base + (wave + i * NW) * 1024);
        } else {
            const int i = p - LA;
            const half_t* ptr = a.Wt + (size_t)b_off[i] + ((size_t)(k_begin + kt) << 5);
            if constexpr (ABL) { if (a.debug & 8) ptr = a.zero; }
            glds16(ptr, base + b_lds[i]);
        }
    };
    auto stage = [&](int kt, int tap, int cc, int buf) {
#pragma unroll
        for (int p = 0; p < LPS; ++p) stage_piece(kt, tap, cc, buf, p);
    };

    // accumulators, TRANSPOSED: acc[nj][mi] = W-tile(nj) x X-tile(mi)^T ; row = channel, lane column = pixel
    acc_t acc[NI][MI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < AR; ++r) acc[j][i][r] = 0.f;

    // fragment lane mapping: row of the MS-row tile, 16-byte k-chunk; lq = the lane's 4-channel block inside a tile
    const int lrow = lane & (MS - 1), lh = lane / MS, swz = swz_of(lrow);
    const int lq = 4 * lh;
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = lrow * ROWB + ((((MS == 32 ? ks * 2 : 0) + lh) ^ swz) << 4);

    // ---- prologue: STAGES-1 stages in flight
    int tap = 0, cc = k_begin << 5;      // tap index, channel offset of the next stage to issue
    if (CONV) { const int q = k_begin >> 1; tap = q % 9; cc = (q / 9) * 64 + (k_begin & 1) * 32; }
    auto advance = [&]() {
        if (!CONV) { cc += 32; return; }
        if (cc & 32) {                       // second half of the 64-chunk done: next tap (or next chunk)
            if (++tap == 9) { tap = 0; cc += 32; } else { cc -= 32; }
        } else {
            cc += 32;
        }
    };
    // the pipelined loop's form of a stage: (tap, cc) -> ONE scalar displacement; a piece = base + precomputed offset, or zeros
    struct StageSc { const char* sbase; const char* wbase; unsigned lbase; bool s1; int tap; int dyE, dyO, dxE, dxO; };
    auto stage_scalars = [&](int kn, int tap, int cc) {
        StageSc sc;
        sc.s1 = cc >= a.C0;                                   // second source of a virtual concat
        const half_t* src = sc.s1 ? a.A1 : a.A0;
        const int Cs = sc.s1 ? a.C1 : a.C0;
        int disp = sc.s1 ? cc - a.C0 : cc;                    // halfs: channel offset (+ the tap's pixel displacement)
        const int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
        sc.dyE = sc.dyO = sc.dxE = sc.dxO = 0;
        if (CONV) {
            if (!up) {
                disp += (ky * a.W + kx) * Cs;
            } else {    // source displacement in bytes by parity: even rows reach up with ky = -1, odd rows reach down with ky = +1
                sc.dyE = ky < 0 ? -a.W * Cs * 2 : 0; sc.dyO = ky > 0 ? a.W * Cs * 2 : 0;
                sc.dxE = kx < 0 ? -Cs * 2 : 0;       sc.dxO = kx > 0 ? Cs * 2 : 0;
            }
        }
        sc.sbase = (const char*)src + (ptrdiff_t)disp * 2;
        sc.wbase = (const char*)a.Wt + ((size_t)(k_begin + kn) << 6);
        sc.lbase = __builtin_amdgcn_readfirstlane(lds_addr(smem) + (kn & 3) * STAGE_BYTES);
        sc.tap = tap;
        return sc;
    };
    auto lean_piece = [&](const StageSc& sc, int p) {
        bool zsrc = false;
        if constexpr (ABL) zsrc = (a.debug & 8) != 0;
        if (p < LA) {
            unsigned off = sc.s1 ? a_b1[p] : a_b0[p];
            if (CONV && up) off += (unsigned)(((a_msk[p] >> 16) & 1 ? sc.dyE : sc.dyO) + ((a_msk[p] >> 17) & 1 ? sc.dxE : sc.dxO));
            const char* ptr = ((a_msk[p] >> sc.tap) & 1) ? sc.sbase + off : (const char*)a.zero;
            if constexpr (ABL) { if (zsrc) ptr = (const char*)a.zero; }
            glds16_m0(ptr, sc.lbase + (wave + p * NW) * 1024);
        } else if (ABL && zsrc) {
            glds16_m0(a.zero, sc.lbase + b_lds[p - LA]);
        } else {
            glds16_sv(b_b[p - LA], sc.wbase, sc.lbase + b_lds[p - LA]);
        }
    };
    auto advance_bf = [&]() {       // the same walk as advance(), branch-free
        if (!CONV) { cc += 32; return; }
        const bool second = (cc & 32) != 0, wrap = second && tap == 8;
        cc = second ? (wrap ? cc + 32 : cc - 32) : cc + 32;
        tap = second ? (wrap ? 0 : tap + 1) : tap;
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        if (PIPE != 2 && s < nk) {
            if constexpr (PIPE == 0) { stage(s, tap, cc, s); advance(); }
            else if constexpr (PIPE == 1) {
                const StageSc sc = stage_scalars(s, tap, cc);
#pragma unroll
                for (int pp = 0; pp < LPS; ++pp) lean_piece(sc, pp);
                advance_bf();
            }
        }
    }

    // ---- epilogue operands (bias, per-sample emb rows) into LDS now, so the epilogue never waits on global loads
    float* bias_l = (float*)(smem + RING_BYTES);
    float* rv_l = bias_l + BN;
    float* mr_l = rv_l + RV_MAX * BN;        // [BM][2] (mean, rstd) of the A rows: LayerNorm folded into this GEMM
    float* u_l = mr_l + 2 * BM;              // [BN]    sum_k W'[n][k]
    const int rps = a.rows_per_sample;
    const int smp0 = m0 / rps;
    const int last_row = min(m0 + BM, a.M) - 1;
    const int nsmp = a.rowvec ? last_row / rps - smp0 + 1 : 0;
    const bool rv_in_lds = nsmp <= RV_MAX;
    for (int c = tid; c < BN; c += T) bias_l[c] = a.bias ? a.bias[n0 + c] : 0.f;
    if constexpr (LN) {
        for (int c = tid; c < BN; c += T) u_l[c] = a.ln_u[n0 + c];
        for (int r = tid; r < BM; r += T) {      // partial sums in fixed slot order; E[x^2] - mean^2 in double
            float mean = 0.f, rstd = 0.f;
            if (m0 + r < a.M) {
                const float* sp = a.ln_stats + (size_t)(m0 + r) * a.ln_slots * 2;
                float sm = 0.f, sq = 0.f;       // fp32 in slot order: the same additions as the 2-stage kernel's epilogue
                for (int p = 0; p < a.ln_slots; ++p) { sm += sp[2 * p]; sq += sp[2 * p + 1]; }
                const double mu = (double)sm / (double)a.C0;
                double var = (double)sq / (double)a.C0 - mu * mu;
                if (var < 0.0) var = 0.0;
                mean = (float)mu;
                rstd = (float)(1.0 / sqrt(var + (double)a.ln_eps));
            }
            mr_l[2 * r] = mean; mr_l[2 * r + 1] = rstd;
        }
    }
    if (a.rowvec && rv_in_lds)
        for (int c = tid; c < nsmp * BN; c += T) {
            const int sidx = c / BN, ch = c - sidx * BN;
            rv_l[c] = a.rowvec[(size_t)(smp0 + sidx) * a.rv_stride + n0 + ch];
        }

    // ---- K loop: one stage (BK = 32) and ONE barrier per iteration; STAGES-1 stages of loads in flight.
    // (what was tried on the way and rejected: DESIGN.md section 4.1 / 4.2)
    if constexpr (PIPE == 0) {
        auto wait_stage = [&](int k_needed) {     // stage k_needed landed; younger stages of this wave may stay in flight
            const int younger = min(nk - 1 - k_needed, STAGES - 2);
            if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * LPS>();
            else if (STAGES >= 3 && younger == 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
        };
        for (int kt = 0; kt < nk; ++kt) {
            wait_stage(kt);
            // my fragment reads of step kt-1 are done (WAR below).  The BUILTIN form (0xC07F = lgkmcnt(0) only) is modelled by
            // hipcc's wait-count pass, so the ds_reads that follow get COUNTED lgkmcnt waits.
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            // every wave's part of stage kt is in LDS, and nobody still reads buffer (kt-1) % STAGES: refill it
            bool refill = kt + STAGES - 1 < nk;
            if constexpr (ABL) refill = refill && !(a.debug & 1);
            if (refill) { stage(kt + STAGES - 1, tap, cc, (kt + STAGES - 1) % STAGES); advance(); }
            const char* As = smem + (kt % STAGES) * STAGE_BYTES + (wm * TM) * ROWB;
            const char* Bs = smem + (kt % STAGES) * STAGE_BYTES + A_BYTES + (wn * TN) * ROWB;
            h8 xf[KS][MI], wf[KS][NI];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {     // read order = MFMA consumption order
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[ks][i] = *(const h8*)(As + i * MS * ROWB + koff[ks]);
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[ks][j] = *(const h8*)(Bs + j * MS * ROWB + koff[ks]);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep all reads ahead of the MFMA cluster
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        if constexpr (MS == 32) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], xf[ks][i], acc[j][i], 0, 0, 0);
                        else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[ks][i], acc[j][i], 0, 0, 0);
                    }
        }
    } else if constexpr (PIPE == 2) {
        // The software-pipelined loop below with the nine taps of a 3x3 convolution served from an LDS-resident HALO tile (round 4;
        // VERDICT r3 item 2).  The loop below fetches every 32-channel activation chunk of the tile NINE times from L2 (one K-step
        // per tap: 16 KB of activations + 20 KB of weights per step); here the chunk's (BM / W + 2) x (W + 2) halo pixels are
        // fetched ONCE (25 KB for 4 x 64 pixels) into one of two halo buffers, its nine K-steps take their X fragments from that
        // image at the tap's displacement, and only the weight stages stream through the four-stage ring: 205 instead of 324 KB
        // from L2 to LDS per chunk (-37 %), 31 instead of 45 LDS-DMA instructions per wave.
        //   * halo image: pixel hp = (r + 1) (W + 2) + x + 1 (r = -1 .. BM / W, x = -1 .. W) at byte 64 hp, its four 16-byte
        //     k-pieces XOR-swizzled on the SOURCE side by (hp >> 2) & 3: a fragment read (16 consecutive pixels at any tap
        //     displacement, one k-piece) then covers all 64 banks whatever its first pixel is; pixels outside the image come from
        //     the zero page, so the taps need no masks at all;
        //   * K order: 64-channel chunk, 32-channel half, tap -- the weights stay packed as they are ((chunk, tap, half) order: a
        //     stage's weight piece is addressed, not moved), only the ORDER of the fp32 sums differs from the other loops; the
        //     launcher takes this loop for BOTH row-tile heights of a layer or for neither (the decision is geometry only), so a
        //     sample's bits do not depend on the batch it is evaluated in;
        //   * the halo of the next 32 channels is requested in the first step of the current one and is eight steps old when its
        //     first fragments are read (behind that step's barrier); the buffer it overwrites was last read two steps before.
        static_assert(CONV && MS == 16 && STAGES == 4 && !ABL, "halo loop: 3x3 stride-1 convolution on the 16x16x32 MFMA");
        constexpr int RW = NI % 5 == 0 ? 5 : 4, WD = 3;
        static_assert(NI % RW == 0 && WD < RW && NI >= RW, "W ring");
        constexpr int HB = halo_px<BM>() * ROWB, HPCS = halo_px<BM>() / 16;   // bytes / 1 KiB pieces of a halo buffer
        constexpr int LH = (HPCS + NW - 1) / NW;                               // halo pieces per wave
        constexpr int WST = BN * ROWB;                                        // bytes of a weight stage
        // a halo buffer holds NSL slots of (RS + 2) x (W + 2) pixels: one slot = this tile's band of RS = BM / W rows of ONE image
        // (H W >= BM), or NSL = BM / (H W) WHOLE images of RS = H rows each (the 8 x 8 level: four images per 256-row tile)
        const int Wd = a.W, Pp = Wd + 2, hwp = a.H * a.W;
        const bool whole = hwp < BM;
        const int RS = whole ? a.H : BM / Wd, HP1 = (RS + 2) * Pp, NSL = whole ? BM / hwp : 1, NPX = NSL * HP1;
        const int bimg = m0 / hwp, y0 = whole ? 0 : (m0 - bimg * hwp) / Wd;
        const int nsub = nk / 9, sub0 = k_begin / 9;                          // 32-channel sub-chunks of this workgroup (split-K: its first one)
        // this wave's halo pieces: piece j = wave + u NW covers halo pixels [16 j, 16 j + 16), lane -> (pixel, k-piece)
        // (source pixel and byte offset of the k-piece, not one finished offset per source: selecting between two per-lane arrays by
        // the run-time source made hipcc put them into scratch memory in the grouped instantiation)
        unsigned h_pix[LH], h_pc[LH];
        bool h_ok[LH];
#pragma unroll
        for (int u = 0; u < LH; ++u) {
            const int q = (wave + u * NW) * 64 + lane, hp = q >> 2, sp = q & 3;
            const int img = hp / HP1, rem = hp - img * HP1, hr = rem / Pp, hx = rem - hr * Pp, y = y0 + hr - 1, x = hx - 1;
            h_ok[u] = hp < NPX && (unsigned)x < (unsigned)Wd && (unsigned)y < (unsigned)a.H && bimg + img < a.B;
            h_pix[u] = h_ok[u] ? (unsigned)((bimg + img) * hwp + y * Wd + x) : 0u;
            h_pc[u] = (unsigned)((sp ^ ((hp >> 2) & 3)) << 4);
        }
        const unsigned lds_h = __builtin_amdgcn_readfirstlane(lds_addr(smem)), lds_w = lds_h + 2 * HB;
        auto issue_halo = [&](int sub) {                                      // halo of sub-chunk `sub` -> buffer sub & 1
            const int cch = (sub0 + sub) << 5;
            const bool s1 = cch >= a.C0;
            const char* sb = s1 ? (const char*)(a.A1 + (cch - a.C0)) : (const char*)(a.A0 + cch);
            const unsigned rowb = (unsigned)(s1 ? a.C1 : a.C0) * 2u;      // bytes per pixel of that source
            const unsigned lb = lds_h + (unsigned)(sub & 1) * HB;
#pragma unroll
            for (int u = 0; u < LH; ++u)
                if (wave + u * NW < HPCS) {
                    const char* ptr = h_ok[u] ? sb + (h_pix[u] * rowb + h_pc[u]) : (const char*)a.zero;
                    glds16_m0(ptr, lb + (unsigned)(wave + u * NW) * 1024u);
                }
        };
        // weight stage of step (sub, tap): the packed order is ((chunk64 * 9 + tap) * 2 + half) * 32
        auto w_src = [&](int sub, int tp) { const int sa = sub0 + sub; return (const char*)a.Wt + ((size_t)(((sa >> 1) * 9 + tp) * 64 + (sa & 1) * 32) << 1); };
        auto issue_w = [&](int kn, int sub, int tp, int p) { glds16_sv(b_b[p], w_src(sub, tp), lds_w + (unsigned)(kn & 3) * WST + (unsigned)(b_lds[p] - A_BYTES)); };
        // fragment reads.  X: this lane's pixel of fragment i sits at halo pixel hpc[i] (+ the tap's displacement)
        int hpc[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int pl = wm * TM + i * MS + lrow, img = pl / (RS * Wd), rem = pl - img * RS * Wd, r = rem / Wd, x = rem - r * Wd;
            hpc[i] = img * HP1 + (r + 1) * Pp + x + 1;
        }
        auto rdXh = [&](int sub, int tp, int i) {
            const int hp = hpc[i] + (tp / 3 - 1) * Pp + (tp - (tp / 3) * 3 - 1);
            return *(const h8*)(smem + (sub & 1) * HB + hp * ROWB + ((lh ^ ((hp >> 2) & 3)) << 4));
        };
        auto rdWh = [&](int kt, int j) { return *(const h8*)(smem + 2 * HB + (kt & 3) * WST + (wn * TN + j * MS) * ROWB + koff[0]); };
        // ---- prologue: the first halo, three weight stages
        int subI = 0, tapI = 0;                   // (sub, tap) of the next stage to issue
        auto next_st = [](int& sub, int& tp) { const bool wrap = tp == 8; tp = wrap ? 0 : tp + 1; sub += wrap ? 1 : 0; };
        issue_halo(0);
#pragma unroll
        for (int sg = 0; sg < STAGES - 1; ++sg)
            if (sg < nk) {
#pragma unroll
                for (int p2 = 0; p2 < LB; ++p2) issue_w(sg, subI, tapI, p2);
                next_st(subI, tapI);
            }
        h8 xa[MI], xb[MI], wr[RW];
        {
            const int younger = min(nk - 1, STAGES - 2);
            if (younger >= 2) wait_vmcnt<2 * LB>(); else if (younger == 1) wait_vmcnt<LB>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = rdXh(0, 0, i);
#pragma unroll
            for (int d = 0; d < WD; ++d) wr[d] = rdWh(0, d);
        }
        int subK = 0, tapK = 0, subR = 0, tapR = 1;        // (sub, tap) of the step being computed / of the step whose X is read
        auto step = [&](auto RF, int kt, h8 (&cur)[MI], h8 (&nxt)[MI]) {
            constexpr bool REFILL = decltype(RF)::value;
            // my pieces of weight stage kt + 1 landed (the pieces of one younger stage may fly; a halo piece issued since is waited
            // for as well: it is older than that stage)
            if (REFILL || kt + 2 < nk) wait_vmcnt<LB>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            if (tapK == 0 && subK + 1 < nsub) issue_halo(subK + 1);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (REFILL) {
#pragma unroll
                    for (int p2 = 0; p2 < LB; ++p2)
                        if (p2 * NI / LB == j) issue_w(kt + STAGES - 1, subI, tapI, p2);
                }
                { const int jn = j + WD; wr[jn % RW] = jn < NI ? rdWh(kt, jn) : rdWh(kt + 1, jn - NI); }
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    if (i * NI / MI == j) nxt[i] = rdXh(subR, tapR, i);
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[j % RW], cur[i], acc[j][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (REFILL) next_st(subI, tapI);
            next_st(subK, tapK);
            next_st(subR, tapR);
        };
        const int nr = max(nk - (STAGES - 1), 0);        // steps that refill
        int kt = 0;
        for (; kt + 1 < nr; kt += 2) {
            step(std::true_type{}, kt, xa, xb);
            step(std::true_type{}, kt + 1, xb, xa);
        }
        if (kt < nr) {
            step(std::true_type{}, kt, xa, xb);
            ++kt;
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = xb[i];
        }
        if (kt < nk) step(std::false_type{}, kt, xa, xb);
        if (kt + 1 < nk) step(std::false_type{}, kt + 1, xb, xa);
        if (kt + 2 < nk) step(std::false_type{}, kt + 2, xa, xb);
        wait_vmcnt<0>();
    } else {
        // Software-pipelined K loop.  In the loop above all eight waves sit in the same phase: after the barrier they all issue
        // LDS-DMAs, then all read fragments (the LDS array saturated, the matrix pipes idle), then all run MFMAs (the LDS idle);
        // a K-step took ~2600 cycles against 1280 of MFMA work per SIMD.  Here a wave's stream is its MFMAs with everything else
        // between them: step kt is NI groups of MI MFMAs (one W fragment against the MI X fragments), and group j also
        //   * reads W fragment j + WD of the step into a register ring (the first WD of step kt + 1 at the end),
        //   * reads its share of the MI X fragments of step kt + 1 into the other half of a double buffer,
        //   * issues its share of the LPS LDS-DMA pieces of stage kt + STAGES - 1,
        // so after the barrier a wave's first MFMA has its operands in registers, and the DMA requests reach the fill path
        // (L2 -> LDS, ~70 GB/s per CU: as long as the MFMAs of a 256x320 step at the clock this kernel holds) evenly spread.
        // Reads of stage kt + 1 happen during step kt, so the barrier of step kt is the one behind which stage kt + 1 has
        // landed: one stage fewer in flight across it than in the loop above.  WAR: buffer (kt - 1) % STAGES is refilled
        // during step kt; its last reads fed MFMAs of step kt - 1, all issued before the barrier of step kt.
        static_assert(MS == 16 && STAGES == 4, "pipelined loop: 16x16x32 MFMA, four-stage ring");
        constexpr int RW = NI % 5 == 0 ? 5 : 4;        // W ring slots; NI % RW == 0 keeps the slot of (step, j) static
        constexpr int WD = 3;                          // ... and W fragment j + WD is requested in group j
        static_assert(NI % RW == 0 && WD < RW && NI >= RW, "W ring");
        auto rdX = [&](int kt, int i) { return *(const h8*)(smem + (kt & 3) * STAGE_BYTES + (wm * TM + i * MS) * ROWB + koff[0]); };
        auto rdW = [&](int kt, int j) { return *(const h8*)(smem + (kt & 3) * STAGE_BYTES + A_BYTES + (wn * TN + j * MS) * ROWB + koff[0]); };
        h8 xa[MI], xb[MI], wr[RW];
        {   // stage 0 landed (everybody's part of it) -> first fragments
            const int younger = min(nk - 1, STAGES - 2);
            if (younger >= 2) wait_vmcnt<2 * LPS>(); else if (younger == 1) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = rdX(0, i);
#pragma unroll
            for (int d = 0; d < WD; ++d) wr[d] = rdW(0, d);
        }
        // What bounds this loop is its instruction stream next to the MFMAs (MI355X under load: the scalar and vector ALU work of
        // the LDS-DMA addressing, ~90 SALU + 50 VALU per wave and step in the first version, cost a third of the step), so:
        // per-slot byte offsets and tap-validity masks are precomputed, a stage's displacement is ONE scalar, the tap / channel
        // walk is branch-free, the weight pieces take the SGPR-base form, M0 is not saved, and the steps that refill (all but
        // the last STAGES - 1) are their own instantiation of the step, so nothing in them is conditional.
        const StageSc sc_abl = stage_scalars(0, 4, 0);
        auto step = [&](auto RF, int kt, h8 (&cur)[MI], h8 (&nxt)[MI]) {
            constexpr bool REFILL = decltype(RF)::value;
            // my pieces of stage kt + 1 landed (kt + 2 may fly; it exists in every refilling step)
            if (REFILL || kt + 2 < nk) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            // the fragments requested at the end of the last step: waited for HERE (the builtin form: hipcc's wait-count pass
            // then knows that nothing is pending and counts the waits of this step exactly; across the loop edge it assumed
            // lgkmcnt(0) in front of the first MFMA, i.e. behind the reads group 0 has just issued)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            StageSc sc = stage_scalars(kt + STAGES - 1, tap, cc);
            if constexpr (ABL) {    // ablation 16: what the tap / channel walk and its scalar arithmetic cost (addresses then repeat)
                if (a.debug & 16) { sc = sc_abl; sc.lbase = __builtin_amdgcn_readfirstlane(lds_addr(smem) + ((kt + STAGES - 1) & 3) * STAGE_BYTES); }
            }
            bool refill = REFILL;
            if constexpr (ABL) refill = refill && !(a.debug & 1);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (REFILL) {
#pragma unroll
                    for (int p = 0; p < LPS; ++p)
                        if (p * NI / LPS == j && (!ABL || refill)) lean_piece(sc, p);
                }
                { const int jn = j + WD; wr[jn % RW] = jn < NI ? rdW(kt, jn) : rdW(kt + 1, jn - NI); }
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    if (i * NI / MI == j) nxt[i] = rdX(kt + 1, i);
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[j % RW], cur[i], acc[j][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (REFILL) { if (!(ABL && (a.debug & 16))) advance_bf(); }
        };
        const int nr = max(nk - (STAGES - 1), 0);        // steps that refill
        int kt = 0;
        for (; kt + 1 < nr; kt += 2) {
            step(std::true_type{}, kt, xa, xb);
            step(std::true_type{}, kt + 1, xb, xa);
        }
        if (kt < nr) {                                   // odd count: one more, then the fragment double buffer changes sides
            step(std::true_type{}, kt, xa, xb);
            ++kt;
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = xb[i];
        }
        if (kt < nk) step(std::false_type{}, kt, xa, xb);
        if (kt + 1 < nk) step(std::false_type{}, kt + 1, xb, xa);
        if (kt + 2 < nk) step(std::false_type{}, kt + 2, xa, xb);
        wait_vmcnt<0>();        // (nothing is in flight here; keeps the ring provably idle for the epilogue's staging)
    }

    if constexpr (SPLIT) {  // split-K: raw fp32 partial tile -> ws[split][row][col]; igemm_splitk_reduce finishes the job
        float* wsp = a.ws + (size_t)split * a.M * a.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = m0 + wm * TM + i * MS + lrow;
            if (row >= a.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < AR / 4; ++g) {
                    const int col = n0 + wn * TN + j * MS + (MS == 32 ? 8 * g : 0) + lq;
                    f32x4 pk = {acc[j][i][g * 4], acc[j][i][g * 4 + 1], acc[j][i][g * 4 + 2], acc[j][i][g * 4 + 3]};
                    *(f32x4*)(wsp + (size_t)row * a.N + col) = pk;
                }
        }
        return;
    } else {
    if (ABL && (a.debug & 4)) {      // ablation: keep the accumulators alive but skip the whole epilogue
        float sink = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) sink += acc[j][i][0] + acc[j][i][AR - 1];
        if (sink == 12345.678f) ((float*)a.out)[0] = sink;
        return;
    }
    // ---------------------------------------------------------------- epilogue
    // MS = 32: acc[j][i][r]: channel = n0 + wn*TN + j*32 + (r&3) + 8*(r>>2) + 4*lh ; pixel = m0 + wm*TM + i*32 + lrow
    // MS = 16: acc[j][i][r]: channel = n0 + wn*TN + j*16 + 4*lh + r            ; pixel = m0 + wm*TM + i*16 + lrow
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // all waves are out of the K loop: the ring is free for staging
    if constexpr (LN) {
        // LayerNorm folded into this GEMM: acc <- rstd_m (acc - mean_m u_n), IN PLACE and before anything else (the bias, which
        // carries beta W, is added by the ordinary epilogue below): one column quad of u and one row's (mean, rstd) live at a time,
        // so the fix-up costs no registers next to the accumulators.  fma + mul as inline asm: see ln_scale.
        const float* uw = u_l + wn * TN;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < AR / 4; ++g) {
                const f32x4 uq = *(const f32x4*)(uw + j * MS + (MS == 32 ? 8 * g : 0) + lq);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int trow = wm * TM + i * MS + lrow;            // row inside the tile
                    const float mean = mr_l[2 * trow], rstd = mr_l[2 * trow + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][i][g * 4 + e] = ln_scale(acc[j][i][g * 4 + e], mean, rstd, uq[e]);
                }
            }
    }
    constexpr int OUT_TN = GEGLU ? TN / 2 : TN;          // output channels this wave produces
    constexpr int PITCH = OUT_TN * 2 + 8;                // wave-private staging tile: 32 pixels x OUT_TN fp16
    constexpr int CPR = OUT_TN / 8;                      // 16-byte chunks per pixel row
    constexpr int WB_IT = (32 * CPR + 63) / 64;          // writeback iterations per lane
    static_assert(!GEGLU || NI % 2 == 0, "GEGLU needs value/gate tile pairs");
    // PATH 1 of the LayerNorm consumers also serves a TRANSPOSED second destination (V^T of the stacked q|k|v projection):
    // that tile is staged [channel][pixel] and leaves as 16-byte runs along the token axis (the general path's 2-byte stores
    // were the slowest part of that GEMM); the host guarantees rows_per_sample % 32 == 0, so a 32-pixel pass has ONE sample
    constexpr bool TR = PATH == 1 && LN && !GEGLU;
    constexpr int TPITCH = 32 * 2 + 16;                  // [channel][32 pixels] fp16
    constexpr int CST_BYTES = TR && OUT_TN * TPITCH > 32 * PITCH ? OUT_TN * TPITCH : 32 * PITCH;
    static_assert(NW * CST_BYTES <= RING_BYTES, "epilogue staging must fit in the ring");
    char* cst = smem + wave * CST_BYTES;
    // destination of this tile (block-uniform): the second one for packed columns >= split_n
    const bool second = a.out2 && n0 >= a.split_n;
    void* const outp = second ? a.out2 : a.out;
    const int okind = second ? a.out_kind2 : a.out_kind, ldo = second ? a.ld_out2 : a.ld_out;
    const int ncol0 = second ? a.split_n : 0, nend = a.out2 ? (second ? a.N : a.split_n) : a.N;
    const bool tr = TR && okind == OUT_F16_T;
    const int ocol0 = GEGLU ? ((n0 + wn * TN) >> 1) : (n0 + wn * TN - ncol0);   // first output column of this wave
    const int nvalid = GEGLU ? a.N / 2 : nend - ncol0;                          // valid output columns of the destination
    const float* bw = bias_l + wn * TN;                                  // this wave's slice of the staged bias

    constexpr int PT = 32 / MS;                          // MFMA pixel tiles per 32-pixel staging pass
    constexpr int NG = AR / 4;                           // 4-channel register quads per accumulator tile
    // The passes below are straight-line code 40 register quads long.  PATH 0 is the general form: output kind, activation and
    // the source of the emb row are block-uniform RUN-TIME branches inside every quad (1450 branches and 78 KB of code in the
    // 256x320 kernel, the hot path threading through all of it).  PATH 1 / 2 are the same code with the common answers compiled
    // in -- fp16 row-major output, no activation (or GEGLU), no emb row (1) / emb rows staged in LDS (2) -- chosen by the host
    // (epilogue_path); they differ from PATH 0 in nothing but the branches.
#pragma unroll
    for (int ip = 0; ip < TM / 32; ++ip) {
#pragma unroll
      for (int hf = 0; hf < PT; ++hf) {
        const int i = ip * PT + hf;
        const int prow = hf * MS + lrow;                  // this lane's pixel inside the 32-pixel pass
        const int row = m0 + wm * TM + ip * 32 + prow;    // ... and in the whole problem
        const int bsmp = (a.rowvec && row < a.M) ? row / rps : smp0;
        const float* rw = rv_l + (bsmp - smp0) * BN + wn * TN;
        // ---- registers -> (bias, emb, activation, scale) -> fp16 -> LDS [pixel][channel]
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (GEGLU && (MS == 32 ? (j & 1) : (j & 2))) continue;      // gate tiles are consumed with their value tile
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int wc = j * MS + (MS == 32 ? 8 * g : 0) + lq;      // packed channel inside the wave tile
                const f32x4 bq = *(const f32x4*)(bw + wc);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][i][g * 4 + e] + bq[e];
                if (PATH == 2 || (PATH == 0 && a.rowvec)) {
                    if (PATH == 2 || rv_in_lds) {
                        const f32x4 rq = *(const f32x4*)(rw + wc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rq[e];
                    } else if (row < a.M) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += a.rowvec[(size_t)bsmp * a.rv_stride + n0 + wn * TN + wc + e];
                    }
                }
                if (PATH == 0 && a.act == ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                } else if (PATH == 0 && a.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr (GEGLU) {      // packed rows: 64-row groups [32 value | 32 gate]
                    constexpr int GJ = MS == 32 ? 1 : 2;                 // gate tile = value tile + GJ
                    const f32x4 gq = *(const f32x4*)(bw + wc + 32);
                    // two gates per call: the polynomial issues as packed fp32 (same operations, same bits as gelu_f)
                    const f32x2 g01 = gelu2_f(f32x2{acc[j + GJ][i][g * 4] + gq[0], acc[j + GJ][i][g * 4 + 1] + gq[1]});
                    const f32x2 g23 = gelu2_f(f32x2{acc[j + GJ][i][g * 4 + 2] + gq[2], acc[j + GJ][i][g * 4 + 3] + gq[3]});
                    v[0] *= g01[0]; v[1] *= g01[1]; v[2] *= g23[0]; v[3] *= g23[1];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= a.scale;
                // output channel inside the wave tile (GEGLU: value channels only, 32 per 64-row group)
                const int oc = !GEGLU ? wc : (MS == 32 ? (j >> 1) * 32 + 8 * g + lq : (j >> 2) * 32 + (j & 1) * 16 + lq);
                if (TR && tr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(half_t*)(cst + (oc + e) * TPITCH + prow * 2) = (half_t)v[e];
                } else if (PATH != 0 || okind == OUT_F16) {
                    h4 pk = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *(h4*)(cst + prow * PITCH + oc * 2) = pk;
                } else if (row < a.M && ocol0 + oc < nvalid) {
                    // direct paths (rare outputs): lane = pixel, 4 consecutive channels
                    const int ch = ocol0 + oc;
                    if (okind == OUT_F32) {
                        f32x4 pk = {v[0], v[1], v[2], v[3]};
                        *(f32x4*)((float*)outp + (size_t)row * ldo + ch) = pk;
                    } else {
                        const int b = row / rps, t = row - b * rps;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const size_t off = ((size_t)b * nvalid + ch + e) * ldo + t;
                            if (okind == OUT_F16_T) ((half_t*)outp)[off] = (half_t)v[e];
                            else ((float*)outp)[off] = v[e];
                        }
                    }
                }
            }
        }
      }
        if (TR && tr) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int row0 = m0 + wm * TM + ip * 32;              // 32 tokens of one sample: 64-byte runs per channel
            const int b = row0 / rps, t0 = row0 - b * rps;
            constexpr int TT = OUT_TN * 4;                        // (channel, 8-token chunk) tasks
            if (row0 < a.M) {
#pragma unroll
                for (int it = 0; it < (TT + 63) / 64; ++it) {
                    const int task = lane + it * 64, c = task >> 2, q = task & 3;
                    if (task < TT && ocol0 + c < nvalid) {
                        const h8 val = *(const h8*)(cst + c * TPITCH + q * 16);
                        *(h8*)((half_t*)outp + ((size_t)b * nvalid + ocol0 + c) * ldo + t0 + q * 8) = val;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the next pass overwrites
        } else if (PATH != 0 || okind == OUT_F16) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private tile: no barrier needed
            // ---- LDS -> (+ residual) -> 16-byte row-contiguous global stores; all loads first, then all stores
            h8 v[WB_IT], rr[WB_IT];
            size_t goff[WB_IT];
            bool ok[WB_IT];
#pragma unroll
            for (int it = 0; it < WB_IT; ++it) {
                const int c = lane + it * 64;
                const int pr = c / CPR, ck = c - pr * CPR;
                const int grow = m0 + wm * TM + ip * 32 + pr;
                const int gcol = ocol0 + ck * 8;
                ok[it] = c < 32 * CPR && grow < a.M && gcol < nvalid;
                goff[it] = (size_t)grow * ldo + gcol;
                if (ok[it]) {
                    const char* sp = cst + pr * PITCH + ck * 16;
                    const h4 lo = *(const h4*)sp, hi = *(const h4*)(sp + 8);
                    v[it] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (a.resid) rr[it] = *(const h8*)(a.resid + (size_t)grow * a.ld_res + gcol);
                }
            }
            bool want_stats = false;
            // 160-column wave tiles own a whole slot; the 64 x 160 tile's two 80-column waves of a row meet in LDS below (HALF)
            constexpr bool HALF = OUT_TN == 80 && WN == 2 && TM == 32 && !GEGLU;
            if constexpr (OUT_TN == 160 || HALF) want_stats = a.stats_out != nullptr;
            float* tab = (float*)cst;                            // [32][CPR][2], over the (by then dead) staging tile
            if (want_stats) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every staging read has landed
#pragma unroll
            for (int it = 0; it < WB_IT; ++it) {
                if (ok[it]) {
                    // fp16 + fp16 rounded once: the packed fp16 add IS the fp32 add followed by a rounding (24 >= 2 * 11 + 2
                    // significand bits: the double rounding is innocuous), at 4 instructions per 8 channels instead of 32
                    if (a.resid) v[it] = v[it] + rr[it];
                    *(h8*)((half_t*)outp + goff[it]) = v[it];
                }
                if (want_stats) {
                    // LayerNorm partial sums of the rows just written, from the STORED fp16 values: this lane's 8 channels
                    // -> (row, chunk) of a wave-private LDS table, right here so that v[it] dies with its store
                    const int c = lane + it * 64;
                    if (c < 32 * CPR) {
                        float sm = 0.f, sq = 0.f;
                        if (ok[it]) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float f = (float)v[it][e]; sm += f; sq += f * f; }
                        }
                        tab[2 * c] = sm; tab[2 * c + 1] = sq;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the next pass overwrites
            if constexpr (HALF) {
                if (want_stats) {        // (block-uniform)
                    // the slot's 20 chunks in the SAME order of additions as everywhere else (row_stats_kernel): the left wave adds
                    // its ten, then the right wave's ten from that wave's table
                    __builtin_amdgcn_s_barrier();
                    if (wn == 0 && lane < 32) {
                        const float* tab1 = (const float*)(smem + (wave + 1) * CST_BYTES);
                        const int grow = m0 + wm * TM + ip * 32 + lane;
                        float sm = 0.f, sq = 0.f;
#pragma unroll
                        for (int k = 0; k < CPR; ++k) { sm += tab[2 * (lane * CPR + k)]; sq += tab[2 * (lane * CPR + k) + 1]; }
#pragma unroll
                        for (int k = 0; k < CPR; ++k) { sm += tab1[2 * (lane * CPR + k)]; sq += tab1[2 * (lane * CPR + k) + 1]; }
                        if (grow < a.M) {
                            float* dst = a.stats_out + ((size_t)grow * (a.N / 160) + n0 / 160) * 2;
                            dst[0] = sm; dst[1] = sq;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else
            if (want_stats) {
                // one lane per row adds the row's CPR chunks in fixed order -> slot (n0 + wn * TN) / 160 of stats_out
                if (lane < 32) {
                    const int grow = m0 + wm * TM + ip * 32 + lane;
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int k = 0; k < CPR; ++k) { sm += tab[2 * (lane * CPR + k)]; sq += tab[2 * (lane * CPR + k) + 1]; }
                    if (grow < a.M) {
                        float* dst = a.stats_out + ((size_t)grow * (a.N / 160) + (n0 + wn * TN) / 160) * 2;
                        dst[0] = sm; dst[1] = sq;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    }   // !SPLIT
}

template <int BM, int BN, int WM, int WN, int STAGES, bool CONV, bool GEGLU, bool SPLIT, int MS, bool LN, int PATH, int PIPE, bool ABL>
__global__ __launch_bounds__(WM * WN * 64) void igemm2_kernel(const IgemmArgs a) {
    igemm2_body<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, PATH, PIPE, ABL>(a);
}
// Up to FGDM_MAX_GROUP problems of the same instantiation and grid in ONE launch (twin layers of the UNet encoder and the
// ControlNets: common.h, "deferred launches"): blockIdx.y selects the argument set, everything else is the kernel above.
struct IgemmArgsG { IgemmArgs g[FGDM_MAX_GROUP]; };
template <int BM, int BN, int WM, int WN, int STAGES, bool CONV, bool GEGLU, bool SPLIT, int MS, bool LN, int PATH, int PIPE, bool ABL>
__global__ __launch_bounds__(WM * WN * 64) void igemm2_pair_kernel(const IgemmArgsG p) {
    igemm2_body<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, PATH, PIPE, ABL>(p.g[blockIdx.y]);
}

// =====================================================================================================================
// Persistent GEGLU projection (round 4).  ff.net.0.proj (ldm/modules/attention.py:37-64) has K = C = 320 / 640 /
// 1280: ten to forty K-steps per 256 x 256 tile and then an epilogue as long as the loop itself.  With one workgroup per CU the
// tiles of a CU run strictly one after the other (tools/bench_igemm.py --batch: the time is linear in tiles per CU), so every tile
// pays the latency of its first LDS-DMA stages in full.  Here 256 workgroups walk the tiles: behind a tile's K loop the first two
// stages of the NEXT tile are requested (ring slots 0 and 1; the epilogue stages through slots 2 and 3), so they travel while the
// current tile's bias / LayerNorm fix-up / GELU / pack / store runs; the third stage follows behind the epilogue.  The K loop and
// the epilogue are the ones of igemm2_kernel<256, 256, 4, 2, 4, false, true, false, 16, LN, 1, 1, false> statement for statement
// (same K order, same MFMA, same ln_scale / gelu2_f / roundings): the output does not change by a bit.
// Only what that layer needs: LINEAR, one source, fp16 row-major output, no emb row, no residual, no second destination.
template <bool LN>
__global__ __launch_bounds__(512) void igemm2_geglu_persist_kernel(const IgemmArgs a, int ntiles, int contiguous) {
    constexpr int BM = 256, BN = 256, WM = 4, WN = 2, NW = 8, T = 512, STAGES = 4, MS = 16;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / MS, NI = TN / MS, AR = 4;
    typedef float acc_t __attribute__((ext_vector_type(AR)));
    constexpr int LA = BM * 4 / 64 / NW, LB = BN * 4 / 64 / NW, LPS = LA + LB;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB, A_BYTES = BM * ROWB, RING_BYTES = STAGES * STAGE_BYTES;
    constexpr int OPS_FLOATS = BN + 2 * BM + BN;          // bias, (mean, rstd) per row, u  -- two sets (this tile's / the next one's)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto swz_of = [](int r) { return ((r >> 3) & 1) << 1; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = a.N / BN;
    const int nk = a.K >> 5;
    // tiles of this workgroup: XCD x = blockIdx & 7 owns a contiguous range of logical tiles (as in igemm2_kernel), and each of its
    // gridDim / 8 workgroups a contiguous RUN of that range: consecutive tiles of a run are column tiles of the same 256 rows, so the
    // rows' LayerNorm (mean, rstd) -- double-precision arithmetic per row -- are formed once per row block, not once per tile
    const int xcd = blockIdx.x & 7, G8 = gridDim.x >> 3, tq = ntiles >> 3, tr = ntiles & 7;
    const int x_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, x_count = tq + (xcd < tr ? 1 : 0);
    // (contiguous = 0: round-robin instead -- at any moment the workgroups of an XCD then sit on ~3 row blocks and share their X
    // tiles through the L2, where contiguous runs keep 32 different X tiles per XCD in flight)
    const int per = (x_count + G8 - 1) / G8, jw = blockIdx.x >> 3;
    const int t_first = contiguous ? x_first + jw * per : x_first + jw;
    const int t_count = contiguous ? min(per, x_count - jw * per) : (x_count - jw + G8 - 1) / G8;
    const int t_step = contiguous ? 1 : G8;
    int local = 0;
    if (t_count <= 0) return;

    const int lrow = lane & (MS - 1), lh = lane / MS, lq = 4 * lh;
    const int koff = lrow * ROWB + ((lh ^ swz_of(lrow)) << 4);
    unsigned a_b0[LA], a_ok[LA], b_b[LB];
    int b_lds[LB];
    auto setup = [&](int m0x, int n0x) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int q = (wave + i * NW) * 64 + lane, r = q >> 2, sp = q & 3, m = m0x + r;
            a_ok[i] = m < a.M ? 1u : 0u;
            a_b0[i] = a_ok[i] ? ((unsigned)m * (unsigned)a.C0 + (unsigned)((sp ^ swz_of(r)) << 3)) * 2u : 0u;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int idx = wave + i * NW, q = idx * 64 + lane, r = q >> 2, sp = q & 3;
            b_b[i] = ((unsigned)(n0x + r) * (unsigned)a.K + (unsigned)((sp ^ swz_of(r)) << 3)) * 2u;
            b_lds[i] = A_BYTES + idx * 1024;
        }
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue_stage = [&](int kn) {                         // K-step kn of the tile set up last -> ring slot kn & 3
        const char* sbase = (const char*)a.A0 + ((size_t)kn << 6);
        const char* wbase = (const char*)a.Wt + ((size_t)kn << 6);
        const unsigned lbase = lds0 + (unsigned)(kn & 3) * STAGE_BYTES;
#pragma unroll
        for (int p2 = 0; p2 < LA; ++p2) glds16_m0(a_ok[p2] ? sbase + a_b0[p2] : (const char*)a.zero, lbase + (unsigned)(wave + p2 * NW) * 1024u);
#pragma unroll
        for (int p2 = 0; p2 < LB; ++p2) glds16_sv(b_b[p2], wbase, lbase + (unsigned)b_lds[p2]);
    };
    float* const ops = (float*)(smem + RING_BYTES);
    auto stage_ops = [&](int m0x, int n0x, int set, bool same_rows) {      // epilogue operands of a tile -> LDS set `set`
        float* bias_l = ops + set * OPS_FLOATS, *mr_l = bias_l + BN, *u_l = mr_l + 2 * BM;
        for (int c = tid; c < BN; c += T) bias_l[c] = a.bias ? a.bias[n0x + c] : 0.f;
        if constexpr (LN) {
            for (int c = tid; c < BN; c += T) u_l[c] = a.ln_u[n0x + c];
            if (same_rows) {             // the previous tile of this run had the same rows: copy its (mean, rstd) from the other set
                const float* prev = ops + (set ^ 1) * OPS_FLOATS + BN;
                for (int r = tid; r < 2 * BM; r += T) mr_l[r] = prev[r];
            } else
            for (int r = tid; r < BM; r += T) {      // partial sums in fixed slot order; E[x^2] - mean^2 in double (as igemm2_kernel)
                float mean = 0.f, rstd = 0.f;
                if (m0x + r < a.M) {
                    const float* sp = a.ln_stats + (size_t)(m0x + r) * a.ln_slots * 2;
                    float sm = 0.f, sq = 0.f;
                    for (int p2 = 0; p2 < a.ln_slots; ++p2) { sm += sp[2 * p2]; sq += sp[2 * p2 + 1]; }
                    const double mu = (double)sm / (double)a.C0;
                    double var = (double)sq / (double)a.C0 - mu * mu;
                    if (var < 0.0) var = 0.0;
                    mean = (float)mu;
                    rstd = (float)(1.0 / sqrt(var + (double)a.ln_eps));
                }
                mr_l[2 * r] = mean; mr_l[2 * r + 1] = rstd;
            }
        }
    };

    int logical = t_first;
    int m0 = (logical / ntn) * BM, n0 = (logical % ntn) * BN;
    setup(m0, n0);
#pragma unroll
    for (int sg = 0; sg < STAGES - 1; ++sg)
        if (sg < nk) issue_stage(sg);
    stage_ops(m0, n0, 0, false);
    int set = 0;
    bool first = true;
    for (;;) {
        acc_t acc[NI][MI];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < AR; ++r) acc[j][i][r] = 0.f;
        // ---- K loop: igemm2_kernel's software-pipelined loop (PIPE = 1), LINEAR
        constexpr int RW = 4, WD = 3;
        auto rdX = [&](int kt, int i) { return *(const h8*)(smem + (kt & 3) * STAGE_BYTES + (wm * TM + i * MS) * ROWB + koff); };
        auto rdW = [&](int kt, int j) { return *(const h8*)(smem + (kt & 3) * STAGE_BYTES + A_BYTES + (wn * TN + j * MS) * ROWB + koff); };
        h8 xa[MI], xb[MI], wr[RW];
        {
            // stage 0 landed.  First tile: stages 1, 2 may fly.  Later tiles: stages 0, 1 were requested before the previous
            // epilogue (whose stores are older than stage 2 as well): everything but stage 2 is waited for
            if (first) {
                const int younger = min(nk - 1, STAGES - 2);
                if (younger >= 2) wait_vmcnt<2 * LPS>(); else if (younger == 1) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            } else {
                if (nk > 2) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = rdX(0, i);
#pragma unroll
            for (int d = 0; d < WD; ++d) wr[d] = rdW(0, d);
        }
        auto step = [&](auto RF, int kt, h8 (&cur)[MI], h8 (&nxt)[MI]) {
            constexpr bool REFILL = decltype(RF)::value;
            if (REFILL || kt + 2 < nk) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            const char* sbase = (const char*)a.A0 + ((size_t)(kt + STAGES - 1) << 6);
            const char* wbase = (const char*)a.Wt + ((size_t)(kt + STAGES - 1) << 6);
            const unsigned lbase = lds0 + (unsigned)((kt + STAGES - 1) & 3) * STAGE_BYTES;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (REFILL) {
#pragma unroll
                    for (int p2 = 0; p2 < LPS; ++p2)
                        if (p2 * NI / LPS == j) {
                            if (p2 < LA) glds16_m0(a_ok[p2] ? sbase + a_b0[p2] : (const char*)a.zero, lbase + (unsigned)(wave + p2 * NW) * 1024u);
                            else glds16_sv(b_b[p2 - LA], wbase, lbase + (unsigned)b_lds[p2 - LA]);
                        }
                }
                { const int jn = j + WD; wr[jn % RW] = jn < NI ? rdW(kt, jn) : rdW(kt + 1, jn - NI); }
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    if (i * NI / MI == j) nxt[i] = rdX(kt + 1, i);
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[j % RW], cur[i], acc[j][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        {
            const int nr = max(nk - (STAGES - 1), 0);
            int kt = 0;
            for (; kt + 1 < nr; kt += 2) {
                step(std::true_type{}, kt, xa, xb);
                step(std::true_type{}, kt + 1, xb, xa);
            }
            if (kt < nr) {
                step(std::true_type{}, kt, xa, xb);
                ++kt;
#pragma unroll
                for (int i = 0; i < MI; ++i) xa[i] = xb[i];
            }
            if (kt < nk) step(std::false_type{}, kt, xa, xb);
            if (kt + 1 < nk) step(std::false_type{}, kt + 1, xb, xa);
            if (kt + 2 < nk) step(std::false_type{}, kt + 2, xa, xb);
            wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // all waves are out of the K loop: the ring is free
        // ---- the next tile's first two stages start their trip now (ring slots 0, 1; the epilogue stages through slots 2, 3)
        local += 1;
        const bool more = local < t_count;
        const int m0c = m0, n0c = n0;
        if (more) {
            logical = t_first + local * t_step;
            m0 = (logical / ntn) * BM; n0 = (logical % ntn) * BN;
            setup(m0, n0);
            if (nk > 0) issue_stage(0);
            if (nk > 1) issue_stage(1);
        }
        // ---- epilogue of the tile just finished: igemm2_kernel's GEGLU PATH 1, statement for statement
        const float* bias_l = ops + set * OPS_FLOATS, *mr_l = bias_l + BN, *u_l = mr_l + 2 * BM;
        if constexpr (LN) {
            const float* uw = u_l + wn * TN;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const f32x4 uq = *(const f32x4*)(uw + j * MS + lq);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int trow = wm * TM + i * MS + lrow;
                    const float mean = mr_l[2 * trow], rstd = mr_l[2 * trow + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][i][e] = ln_scale(acc[j][i][e], mean, rstd, uq[e]);
                }
            }
        }
        constexpr int OUT_TN = TN / 2, PITCH = OUT_TN * 2 + 8, CPR = OUT_TN / 8, WB_IT = (32 * CPR + 63) / 64, CST_BYTES = 32 * PITCH;
        static_assert(NW * CST_BYTES <= 2 * STAGE_BYTES, "epilogue staging must fit ring slots 2 and 3");
        char* cst = smem + 2 * STAGE_BYTES + wave * CST_BYTES;
        const int ocol0 = (n0c + wn * TN) >> 1, nvalid = a.N / 2;
        const float* bw = bias_l + wn * TN;
#pragma unroll
        for (int ip = 0; ip < TM / 32; ++ip) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int i = ip * 2 + hf;
                const int prow = hf * MS + lrow;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (j & 2) continue;                      // gate tiles are consumed with their value tile
                    const int wc = j * MS + lq;
                    const f32x4 bq = *(const f32x4*)(bw + wc);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][e] + bq[e];
                    const f32x4 gq = *(const f32x4*)(bw + wc + 32);
                    const f32x2 g01 = gelu2_f(f32x2{acc[j + 2][i][0] + gq[0], acc[j + 2][i][1] + gq[1]});
                    const f32x2 g23 = gelu2_f(f32x2{acc[j + 2][i][2] + gq[2], acc[j + 2][i][3] + gq[3]});
                    v[0] *= g01[0]; v[1] *= g01[1]; v[2] *= g23[0]; v[3] *= g23[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= a.scale;
                    const int oc = (j >> 2) * 32 + (j & 1) * 16 + lq;
                    const h4 pk = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *(h4*)(cst + prow * PITCH + oc * 2) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private tile: no barrier needed
#pragma unroll
            for (int it = 0; it < WB_IT; ++it) {
                const int c = lane + it * 64, pr = c / CPR, ck = c - pr * CPR;
                const int grow = m0c + wm * TM + ip * 32 + pr, gcol = ocol0 + ck * 8;
                if (c < 32 * CPR && grow < a.M && gcol < nvalid) {
                    const char* sp = cst + pr * PITCH + ck * 16;
                    const h4 lo = *(const h4*)sp, hi = *(const h4*)(sp + 8);
                    *(h8*)((half_t*)a.out + (size_t)grow * a.ld_out + gcol) = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the next pass overwrites
        }
        if (!more) break;
        // ---- the next tile's epilogue operands (other LDS set), then -- behind a barrier: every wave is out of the staging
        // slots -- its third stage
        set ^= 1;
        stage_ops(m0, n0, set, m0 == m0c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nk > 2) issue_stage(2);
        first = false;
    }
}

template <bool LN>
int launch_geglu_persist(const IgemmArgs& a, int ntiles, int contiguous, hipStream_t s) {
    constexpr int smem = 4 * (256 + 256) * ROWB + 2 * (256 + 2 * 256 + 256) * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    auto k = igemm2_geglu_persist_kernel<LN>;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    FGDM_LAUNCH(k, dim3(256), dim3(512), smem, s, a, ntiles, contiguous);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

template <int BM, int BN, int WM, int WN, int STAGES, bool CONV, bool GEGLU, bool SPLIT, int MS, bool LN, int PATH, int PIPE, bool ABL>
int launch2p(const IgemmArgs& a, hipStream_t s) {
    constexpr int ring = ring_bytes<BM, BN, STAGES, PIPE>();
    constexpr int smem = ring + (1 + RV_MAX) * BN * 4 + (2 * BM + BN) * 4;      // + staged bias, emb rows, LayerNorm (mean, rstd), u
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    auto k = igemm2_kernel<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, PATH, PIPE, ABL>;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const unsigned gx = (unsigned)(ntm * ntn * (SPLIT ? a.splitk : 1));
    if (fgdm_recording()) {
        auto single = [=](hipStream_t rs) -> int {
            hipLaunchKernelGGL(k, dim3(gx), dim3(WM * WN * 64), smem, rs, a);
            return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
        };
        // the hot instantiations (pipelined loop, straight-line epilogue) have a twin that takes two argument sets
        if constexpr (PIPE != 0 && PATH != 0 && !SPLIT && !ABL && MS == 16) {
            static bool pair_attr_set = false;
            auto kp = igemm2_pair_kernel<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, PATH, PIPE, ABL>;
            if (!pair_attr_set) {
                HIP_TRY(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                pair_attr_set = true;
            }
            IgemmGroupFn pf = [](const IgemmArgs* const* av, int n, unsigned grid_x, hipStream_t rs) -> int {
                if (n < 2 || n > FGDM_MAX_GROUP) return FGDM_ERR_ARG;
                IgemmArgsG pg;
                for (int i = 0; i < FGDM_MAX_GROUP; ++i) pg.g[i] = *av[i < n ? i : 0];
                hipLaunchKernelGGL((igemm2_pair_kernel<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, PATH, PIPE, ABL>), dim3(grid_x, n),
                                   dim3(WM * WN * 64), smem, rs, pg);
                return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
            };
            fgdm_record(single, (const void*)kp, pf, &a, gx);
        } else {
            fgdm_record(single);
        }
        return FGDM_OK;
    }
    hipLaunchKernelGGL(k, dim3(gx), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}
// Which epilogue PATH serves `a` (see igemm2_kernel): 1 / 2 need the fp16 row-major output without an activation (GEGLU is
// its own instantiation); 2 also needs every tile's emb rows to fit the LDS staging area (a BM-row tile touches at most
// (BM - 1) / rows_per_sample + 2 samples)
template <int BM, bool GEGLU, bool LN>
int epilogue_path(const IgemmArgs& a) {
    if (a.out_kind != OUT_F16 || !(a.act == ACT_NONE || GEGLU)) return 0;
    if (a.out2 && a.out_kind2 != OUT_F16) {     // transposed second destination: staged by PATH 1 of the LayerNorm consumers
        const bool staged = LN && a.out_kind2 == OUT_F16_T && a.rows_per_sample % 32 == 0 && a.M % 32 == 0 && a.ld_out2 % 8 == 0 &&
                            ((size_t)a.out2 & 15) == 0;
        if (!staged || a.rowvec) return 0;
    }
    if (!a.rowvec) return 1;
    return (BM - 1) / a.rows_per_sample + 2 <= RV_MAX ? 2 : 0;
}
// FAST: also build the specialised epilogues (the hot tile configurations); otherwise PATH 0 only.  The pipelined K loop exists
// for PATH 1 / 2 (every layer of the sampling loop but a handful); the general epilogue keeps the phase-locked loop (its register
// stage and the pipelined loop's fragment ring do not fit 256 VGPRs together).  a.debug != 0 (only tools/bench_igemm.py sets
// it) takes the ablation instantiation: PATH 1 of the FAST non-LayerNorm tiles.
template <int BM, int BN, int WM, int WN, int STAGES, bool CONV, bool GEGLU, bool SPLIT, int MS, bool LN, bool FAST, int PIPE>
int launch2(const IgemmArgs& a, hipStream_t s) {
    static const bool fast_on = !(getenv("FGDM_IGEMM_EPI_PATHS") && atoi(getenv("FGDM_IGEMM_EPI_PATHS")) == 0);   // A/B knob
    if constexpr (FAST && !SPLIT && !LN) {
        if (a.debug)
            return epilogue_path<BM, GEGLU, LN>(a) == 1 ? launch2p<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, 1, (PIPE == 2 ? 1 : PIPE), true>(a, s)
                                                        : FGDM_ERR_ARG;
    }
    if (a.debug) return FGDM_ERR_ARG;
    // split-K tiles leave as raw fp32 partial sums (no register stage at all): either K loop fits
    if constexpr (SPLIT) return launch2p<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, 0, PIPE, false>(a, s);
    if constexpr (FAST && !SPLIT) {
        switch (fast_on ? epilogue_path<BM, GEGLU, LN>(a) : 0) {
            case 1: return launch2p<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, 1, PIPE, false>(a, s);
            case 2: if constexpr (!GEGLU && !LN) return launch2p<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, 2, PIPE, false>(a, s);
            default: break;
        }
    }
    return launch2p<BM, BN, WM, WN, STAGES, CONV, GEGLU, SPLIT, MS, LN, 0, 0, false>(a, s);
}
template <int BM, int BN, bool GEGLU, int MS, int PIPE, int WM, int WN>
int launch2m(const IgemmArgs& a, hipStream_t s) {
    constexpr int PL = PIPE == 2 ? 1 : PIPE;        // the halo loop is a convolution's
    return a.mode == IG_LINEAR ? launch2<BM, BN, WM, WN, 4, false, GEGLU, false, MS, false, MS == 16, PL>(a, s)
                               : launch2<BM, BN, WM, WN, 4, true, GEGLU, false, MS, false, MS == 16, (GEGLU ? PL : PIPE)>(a, s);
}
// consumer of a folded LayerNorm: always a LINEAR GEMM
template <int BM, int BN, bool GEGLU, int MS, int PIPE, int WM, int WN>
int launch2ln(const IgemmArgs& a, hipStream_t s) {
    return launch2<BM, BN, WM, WN, 4, false, GEGLU, false, MS, true, MS == 16, (PIPE == 2 ? 1 : PIPE)>(a, s);
}
// one tile configuration (waves as WM x WN, four-stage ring), any mode
template <int BM, int BN, int MS, int PIPE, int WM = 4, int WN = 2>
int launch_tile(const IgemmArgs& a, hipStream_t s) {
    const bool g = a.act == ACT_GEGLU;
    if (a.ln_stats) {
        if constexpr (BN == 256) { if (g) return launch2ln<BM, BN, true, MS, PIPE, WM, WN>(a, s); }
        if constexpr (BN == 128) return FGDM_ERR_ARG;
        else return launch2ln<BM, BN, false, MS, PIPE, WM, WN>(a, s);
    }
    if constexpr (BN == 256) { if (g) return launch2m<BM, BN, true, MS, PIPE, WM, WN>(a, s); }
    return launch2m<BM, BN, false, MS, PIPE, WM, WN>(a, s);
}

// out[m][n] = ((sum_s ws[s][m][n]) + bias + emb -> act) * scale + resid, fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmArgs a) {
    const size_t total4 = (size_t)a.M * a.N / 4, plane = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t e0 = i * 4;
        const int row = (int)(e0 / a.N), col = (int)(e0 - (size_t)row * a.N);
        f32x4 v = *(const f32x4*)(a.ws + e0);
        for (int s = 1; s < a.splitk; ++s) {
            const f32x4 u = *(const f32x4*)(a.ws + (size_t)s * plane + e0);
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        h4 r = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
        if (a.resid) r = *(const h4*)(a.resid + (size_t)row * a.ld_res + col);
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e] + (a.bias ? a.bias[col + e] : 0.f);
            if (a.rowvec) x += a.rowvec[(size_t)(row / a.rows_per_sample) * a.rv_stride + col + e];
            if (a.act == ACT_SILU) x = silu_f(x);
            else if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
            else if (a.act == ACT_QGELU) x = silu_f(x, 1.702f);
            const half_t y = (half_t)(x * a.scale);                     // same two roundings as the fused epilogue
            o[e] = a.resid ? (half_t)((float)y + (float)r[e]) : y;
        }
        *(h4*)((half_t*)a.out + (size_t)row * a.ld_out + col) = o;
    }
}

}  // namespace

// Split K for the long-contraction layers of the coarse latent levels (8x8 positions per sample: M = 64 B rows cannot fill
// the chip with 128-row tiles; 16x16: not with 256-row tiles).  The decision depends ONLY on the per-sample geometry and K -- never on
// the batch size -- so a sample's result stays bit-identical whatever batch (or rank shard) it is evaluated in.
static thread_local bool g_twin_layers = false;
void igemm_set_twin_layers(bool on) { g_twin_layers = on; }

int igemm_splitk_factor(const IgemmArgs& a) {
    if (a.force_cfg || a.act == ACT_GEGLU || a.out_kind != OUT_F16 || a.N % 320 || (a.K & 31)) return 1;
    const int nk = a.K >> 5;
    static const bool fat = !(getenv("FGDM_SPLITK_FAT") && atoi(getenv("FGDM_SPLITK_FAT")) == 0);          // A/B knob
    // the decoder's convolutions of the 16x16 level (they have no ControlNet twin to share a launch with): two ways, so that at
    // B = 32 they fill the chip with 256 x 320 tiles (2560->1280: 576 -> 479 us with the reduction pass; three or four ways 551 /
    // 509; 1280->1280: 301 -> ~255).  The K = 11520 layers WITH a twin stay whole: as twin launches they are on fat tiles already
    // (220 us each).  "Has a twin" is a property of the layer in this engine (igemm_set_twin_layers), not of how it is launched
    if (fat && a.rows_per_sample > 64 && a.rows_per_sample <= 256 && (nk >= 540 || (nk >= 360 && !g_twin_layers))) return 2;
    if (a.rows_per_sample > 64 || nk < 96) return 1;
    // the 3x3 convolutions of that level (K >= 11520) split eight ways: at B = 32 that is what lets them run on 256 x 320 tiles
    // (igemm_launch), two thirds of the L2 -> LDS bytes of the 128-row tiles (same box, B = 32: 1280->1280 88 -> 84 us,
    // 2560->1280 164 -> 137 us; the K = 5120 linear loses with either change and stays at four)
    return (fat && nk >= 360) ? 8 : 4;
}
int igemm_splitk_reduce(const IgemmArgs& a, hipStream_t s) {
    const size_t total4 = (size_t)a.M * a.N / 4;
    const int grid = (int)std::min<size_t>(2048, (total4 + 255) / 256);
    FGDM_LAUNCH(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

// The halo loop (PIPE = 2) serves stride-1 3x3 convolutions whose row tiles -- of BOTH tile heights, 256 and 128 rows -- are whole rows
// of one image (W = 16 / 32 / 64, H W a multiple of 256) or whole 8 x 8 images: a property of the layer's geometry, never of the batch
static bool halo_geometry(const IgemmArgs& a) {
    if (a.mode != IG_CONV3 || a.Ho != a.H || a.Wo != a.W || (a.C0 & 63) || (a.C1 & 63) || (a.K % 288) || a.debug) return false;
    if ((a.W == 16 || a.W == 32 || a.W == 64) && (a.H * a.W) % 256 == 0) return true;
    return a.W == 8 && a.H == 8;
}

// cfg & 15: 0 = 256x320   1 = 256x256 (GEGLU-capable)   2 = 128x320   6 = 256x128   (8 waves, 4-stage ring, 16x16x32 MFMA)
//           7 = 64x160, four waves, two workgroups per CU: the linears of the 8x8 level (M = 64 rows per sample), whose grids
//               cannot fill the chip with taller tiles
//           3..5 = tiles 0..2 on the 32x32x16 MFMA (kept for A/B measurements)
// cfg >> 4: K loop: 0 = the process default (FGDM_IGEMM_PIPE, default 1), 1 = the phase-locked loop, 2 = the software-pipelined one
int igemm2_launch(const IgemmArgs& a, int cfg, hipStream_t s) {
    if ((a.K & 31) || (a.C0 & 31) || (a.C1 & 31)) return FGDM_ERR_ARG;
    static const int pipe_default = getenv("FGDM_IGEMM_PIPE") ? atoi(getenv("FGDM_IGEMM_PIPE")) : 1;
    const int tile = cfg & 15, psel = cfg >> 4;
    // the pipelined loop addresses with 32-bit byte offsets: operands of 4 GB or more stay on the phase-locked loop
    const size_t px = a.mode == IG_LINEAR ? (size_t)a.M : (size_t)a.B * a.H * a.W;
    const bool small = px * (size_t)std::max(a.C0, a.C1) * 2 < (1ull << 32) && (size_t)(a.N + 320) * a.K * 2 < (1ull << 32);
    const bool pipe = (psel == 0 ? pipe_default != 0 : psel == 2) && small;
    const int bn = (tile == 1 || tile == 4) ? 256 : tile == 6 ? 128 : tile == 7 ? 160 : 320;
    if (a.N % bn) return FGDM_ERR_ARG;          // weight rows beyond N are not padded to this tile
    if (a.act == ACT_GEGLU && bn != 256) return FGDM_ERR_ARG;
    if (a.splitk > 1) {
        if ((tile != 2 && tile != 0) || a.ln_stats) return FGDM_ERR_ARG;
        // the halo loop for split K too (round 4): every K slice must be whole 32-channel sub-chunks (nine steps each)
        static const bool halo_sk = !(getenv("FGDM_IGEMM_HALO") && atoi(getenv("FGDM_IGEMM_HALO")) == 0);
        const int nk_all = a.K >> 5;
        if (halo_sk && pipe && halo_geometry(a) && nk_all % a.splitk == 0 && (nk_all / a.splitk) % 9 == 0)
            return tile == 0 ? launch2<256, 320, 4, 2, 4, true, false, true, 16, false, false, 2>(a, s)
                             : launch2<128, 320, 4, 2, 4, true, false, true, 16, false, false, 2>(a, s);
        if (tile == 0 && pipe)       // (the phase-locked loop has no 256 x 320 split-K instantiation: 128 x 320 below)
            return a.mode == IG_LINEAR ? launch2<256, 320, 4, 2, 4, false, false, true, 16, false, false, 1>(a, s)
                                       : launch2<256, 320, 4, 2, 4, true, false, true, 16, false, false, 1>(a, s);
        if (pipe)
            return a.mode == IG_LINEAR ? launch2<128, 320, 4, 2, 4, false, false, true, 16, false, false, 1>(a, s)
                                       : launch2<128, 320, 4, 2, 4, true, false, true, 16, false, false, 1>(a, s);
        return a.mode == IG_LINEAR ? launch2<128, 320, 4, 2, 4, false, false, true, 16, false, false, 0>(a, s)
                                   : launch2<128, 320, 4, 2, 4, true, false, true, 16, false, false, 0>(a, s);
    }
    if (a.ln_stats && a.mode != IG_LINEAR) return FGDM_ERR_ARG;
    // the persistent GEGLU projection (igemm2_geglu_persist_kernel): 256 workgroups walk the tiles once there are more tiles than
    // CUs.  FGDM_IGEMM_PERSIST=0: A/B knob (the outputs are bit-identical either way)
    static const int persist_mode = getenv("FGDM_IGEMM_PERSIST") ? atoi(getenv("FGDM_IGEMM_PERSIST")) : 1;      // 1 = round-robin tiles, 2 = contiguous runs
    const bool persist_on = persist_mode != 0;
    if (persist_on && tile == 1 && pipe && a.act == ACT_GEGLU && a.mode == IG_LINEAR && !a.C1 && a.out_kind == OUT_F16 && !a.rowvec &&
        !a.resid && !a.out2 && !a.stats_out && !a.debug && a.N % 256 == 0) {
        const int ntiles = ((a.M + 255) / 256) * (a.N / 256);
        if (ntiles > 256) return a.ln_stats ? launch_geglu_persist<true>(a, ntiles, persist_mode == 2, s) : launch_geglu_persist<false>(a, ntiles, persist_mode == 2, s);
    }
    // the halo loop (PIPE = 2) where the geometry allows it (halo_geometry).  FGDM_IGEMM_HALO=0: A/B knob
    static const bool halo_on = !(getenv("FGDM_IGEMM_HALO") && atoi(getenv("FGDM_IGEMM_HALO")) == 0);
    const bool halo = halo_on && pipe && halo_geometry(a);
    switch (tile) {
        case 0: return halo ? launch_tile<256, 320, 16, 2>(a, s) : pipe ? launch_tile<256, 320, 16, 1>(a, s) : launch_tile<256, 320, 16, 0>(a, s);
        case 1: return pipe ? launch_tile<256, 256, 16, 1>(a, s) : launch_tile<256, 256, 16, 0>(a, s);
        case 2: return halo ? launch_tile<128, 320, 16, 2>(a, s) : pipe ? launch_tile<128, 320, 16, 1>(a, s) : launch_tile<128, 320, 16, 0>(a, s);
        case 3: return launch_tile<256, 320, 32, 0>(a, s);
        case 4: return launch_tile<256, 256, 32, 0>(a, s);
        case 5: return launch_tile<128, 320, 32, 0>(a, s);
        case 6: return pipe ? launch_tile<256, 128, 16, 1>(a, s) : launch_tile<256, 128, 16, 0>(a, s);
        case 7: return a.mode != IG_LINEAR ? FGDM_ERR_ARG : pipe ? launch_tile<64, 160, 16, 1, 2, 2>(a, s) : launch_tile<64, 160, 16, 0, 2, 2>(a, s);
        default: return FGDM_ERR_ARG;
    }
}
