// Implicit-GEMM convolution / linear kernel for gfx950 (CDNA4), fp16 MFMA with fp32 accumulate.
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
//   m = output pixel (b, oy, ox) of an NHWC fp16 activation, k = (tap, cin), W packed [N][K] K-contiguous.
//
// Covers every dense contraction of the hot path (SURVEY.md section 2 op inventory): conv3x3 s1 / s2, conv3x3 on a
// nearest-2x-upsampled input (upsample folded into the address calculation), conv1x1 / nn.Linear, and the
// channel concat of the UNet decoder as two base pointers ("virtual concat").  Epilogue fuses bias, the
// ResBlock timestep-embedding add, SiLU / ReLU / GEGLU, the ControlNet scale and the residual add.
//
// Structure: BK = 64 halfs per K-step; global -> LDS with global_load_lds_dwordx4 (16 B / lane, no VGPR
// staging); LDS rows are 128 B, XOR-swizzled on the *source* chunk index so that the linear LDS image is
// conflict-free for ds_read_b128 fragment reads; 2-stage LDS ring, one barrier per K-step; each wave
// owns a (BM/WM)x(BN/WN) tile of v_mfma_f32_32x32x16_f16 accumulators; XCD-aware block->tile remap.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}


// rstd (acc - mean u): one fma and one multiply that must NOT be contracted with the bias add that follows -- the pipelined
// kernel (igemm2.hip ln_fix) rounds in exactly this sequence, and which kernel runs a layer must not change a bit
__device__ __forceinline__ float ln_scale(float acc, float mean, float rstd, float u) {
    float t, w;      // inline asm: opaque to the backend's multiply-add fusion (-ffp-contract=fast)
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(-mean), "v"(u), "v"(acc));
    asm("v_mul_f32 %0, %1, %2" : "=v"(w) : "v"(rstd), "v"(t));
    return w;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(const IgemmArgs a) {
    constexpr int T = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_IT = BM * 8 / T, B_IT = BN * 8 / T;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the thread count");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
    // range of tiles with the N tiles of one M tile adjacent, so A rows and the weight panel stay L2-hot.
    const int ntn = (a.N + BN - 1) / BN;
    const int nb = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (logical / ntn) * BM, n0 = (logical % ntn) * BN;

    const int Ctot = a.C0 + a.C1;
    const int cpt = Ctot >> 6;   // 64-wide chunks per tap
    const int nk = a.K >> 6;

    // ---- per-thread A row descriptors (fixed for the whole K loop)
    int a_pix[A_IT], a_yx[A_IT], a_co[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int q = j * T + tid, r = q >> 3, s = q & 7;
        a_co[j] = ((s ^ ((r >> 1) & 7)) << 3);   // source chunk (halfs) that lands in LDS slot s of row r
        const int m = m0 + r;
        a_pix[j] = -1;
        a_yx[j] = 0;
        if (m < a.M) {
            if (a.mode == IG_LINEAR) {
                a_pix[j] = m;
            } else {
                const int hw = a.Ho * a.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                a_pix[j] = b * a.H * a.W;
                a_yx[j] = (oy << 16) | ox;
            }
        }
    }
    int b_off[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int q = j * T + tid, r = q >> 3, s = q & 7;
        b_off[j] = (n0 + r) * a.K + ((s ^ ((r >> 1) & 7)) << 3);
    }

    auto stage = [&](int kt, int tap, int cc, int buf) {
        char* As = smem + buf * STAGE;
        char* Bs = As + A_BYTES;
        // K-step order for convolutions: 64-channel chunk outermost, the 9 taps inside (see igemm2.hip); `cc` = chunk
        const half_t* src = a.A0;
        int Cs = a.C0, co = cc << 6;
        if (co >= a.C0) { src = a.A1; Cs = a.C1; co -= a.C0; }
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const half_t* p = a.zero;
            if (a_pix[j] >= 0) {
                if (a.mode == IG_LINEAR) {
                    p = src + (size_t)a_pix[j] * Cs + co + a_co[j];
                } else {
                    const int oy = a_yx[j] >> 16, ox = a_yx[j] & 0xffff;
                    int iy, ix;
                    bool ok;
                    if (a.mode == IG_CONV3) {
                        iy = oy + ky - 1; ix = ox + kx - 1;
                        ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    } else if (a.mode == IG_CONV3_S2) {
                        iy = 2 * oy + ky - 1; ix = 2 * ox + kx - 1;
                        ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    } else {   // IG_CONV3_UP2: conv over the virtual 2H x 2W nearest-upsampled image
                        const int uy = oy + ky - 1, ux = ox + kx - 1;
                        ok = (unsigned)uy < (unsigned)(2 * a.H) && (unsigned)ux < (unsigned)(2 * a.W);
                        iy = uy >> 1; ix = ux >> 1;
                    }
                    if (ok) p = src + (size_t)(a_pix[j] + iy * a.W + ix) * Cs + co + a_co[j];
                }
            }
            glds16(p, As + (j * T + wave * 64) * 16);
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            glds16(a.Wt + (size_t)b_off[j] + ((size_t)kt << 6), Bs + (j * T + wave * 64) * 16);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row = 32*tile + (lane & 31); (row >> 1) & 7 depends on the lane only
    const int lrow = lane & 31, lh = lane >> 5, swz = (lrow >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = lrow * 128 + (((ks * 2 + lh) ^ swz) << 4);

    int tap = 0, cc = 0;
    stage(0, 0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile kt has landed for every wave; buffer (kt+1)&1 is no longer being read
        if (kt + 1 < nk) {
            if (a.mode == IG_LINEAR) ++cc;
            else if (++tap == 9) { tap = 0; ++cc; }
            stage(kt + 1, tap, cc, (kt + 1) & 1);
        }
        const char* As = smem + (kt & 1) * STAGE + (wm * TM) * 128;
        const char* Bs = smem + (kt & 1) * STAGE + A_BYTES + (wn * TN) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const h8*)(As + i * 32 * 128 + koff[ks]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *(const h8*)(Bs + j * 32 * 128 + koff[ks]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---------------------------------------------------------------- epilogue
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (a.ln_stats) {
        // LayerNorm folded into this GEMM (see IgemmArgs): acc <- rstd (acc - mean u); the bias (which carries beta W) follows
        // below.  (mean, rstd) of the 16 rows this lane holds per row tile, from the partial sums in fixed slot order.
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float sm[16], sq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sm[r] = 0.f; sq[r] = 0.f; }
            for (int p = 0; p < a.ln_slots; ++p) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < a.M) {
                        const float* sp = a.ln_stats + ((size_t)row * a.ln_slots + p) * 2;
                        sm[r] += sp[0]; sq[r] += sp[1];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double mu = (double)sm[r] / (double)a.C0;
                double var = (double)sq[r] / (double)a.C0 - mu * mu;
                if (var < 0.0) var = 0.0;
                sm[r] = (float)mu;
                sq[r] = (float)(1.0 / sqrt(var + (double)a.ln_eps));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pc = n0 + wn * TN + j * 32 + (lane & 31);
                const float un = pc < a.N ? a.ln_u[pc] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = ln_scale(acc[i][j][r], sm[r], sq[r], un);   // + bias below
            }
        }
    }
    const int rps = a.rows_per_sample;
    const bool geglu = (a.act == ACT_GEGLU);
    // destination of this tile (block-uniform): the second one for packed columns >= split_n (see IgemmArgs::out2)
    const bool second = a.out2 && n0 >= a.split_n;
    void* const outp = second ? a.out2 : a.out;
    const int okind = second ? a.out_kind2 : a.out_kind, ldo = second ? a.ld_out2 : a.ld_out;
    const int ncol0 = second ? a.split_n : 0, nchan = (a.out2 ? (second ? a.N : a.split_n) : a.N) - ncol0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (geglu && (j & 1)) continue;   // gate tiles are consumed together with their value tile
            const int pcol = n0 + wn * TN + j * 32 + lrow;   // packed column of this lane
            if (pcol >= a.N) continue;
            const float bias = a.bias ? a.bias[pcol] : 0.f;
            float gbias = 0.f;
            if (geglu) gbias = a.bias ? a.bias[pcol + 32] : 0.f;
            const int ocol = geglu ? ((pcol >> 6) << 5) + lrow : pcol - ncol0;
            // per-sample emb values for this lane's 16 rows: all loads issued before any store (stores to `out`
            // may alias for the compiler and would otherwise serialise each load behind the previous store)
            float rvv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                rvv[r] = (a.rowvec && row < a.M) ? a.rowvec[(size_t)(row / rps) * a.rv_stride + pcol] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row0 = m0 + wm * TM + i * 32 + 8 * g + 4 * lh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = row0 + e;
                    float x = acc[i][j][g * 4 + e] + bias + rvv[g * 4 + e];
                    if (a.act == ACT_SILU) x = silu_f(x);
                    else if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                    else if (a.act == ACT_QGELU) x = silu_f(x, 1.702f);
                    else if (geglu) {
                        if constexpr (NI >= 2) x = x * gelu_f(acc[i][j | 1][g * 4 + e] + gbias);
                    }
                    v[e] = x * a.scale;
                }
                if (okind == OUT_F16) {
                    half_t* o = (half_t*)outp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = row0 + e;
                        if (row < a.M) {
                            // numerics policy (oracle/precision.py): the GEMM result is rounded to fp16, THEN the fp16
                            // residual is added in fp32 and the sum rounded -- identical in every kernel of the family
                            half_t x = (half_t)v[e];
                            if (a.resid) x = (half_t)((float)x + (float)a.resid[(size_t)row * a.ld_res + ocol]);
                            o[(size_t)row * ldo + ocol] = x;
                        }
                    }
                } else if (okind == OUT_F32) {
                    float* o = (float*)outp;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row0 + e < a.M) o[(size_t)(row0 + e) * ldo + ocol] = v[e];
                } else {
                    // transposed outputs: [b][col][t]; ld_out = elements per (b, col) row
                    const bool vec = ((rps & 3) == 0) && (row0 + 3 < a.M);
                    if (vec) {
                        const int b = row0 / rps, t = row0 - b * rps;
                        const size_t off = ((size_t)b * nchan + ocol) * ldo + t;
                        if (okind == OUT_F16_T) {
                            h4 pk = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            *(h4*)((half_t*)outp + off) = pk;
                        } else {
                            f32x4 pk = {v[0], v[1], v[2], v[3]};
                            *(f32x4*)((float*)outp + off) = pk;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = row0 + e;
                            if (row < a.M) {
                                const int b = row / rps, t = row - b * rps;
                                const size_t off = ((size_t)b * nchan + ocol) * ldo + t;
                                if (okind == OUT_F16_T) ((half_t*)outp)[off] = (half_t)v[e];
                                else ((float*)outp)[off] = v[e];
                            }
                        }
                    }
                }
            }
        }
    }
}

size_t igemm_npad(int n) { return (size_t)((n + 127) / 128) * 128; }

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const IgemmArgs& a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    auto k = igemm_kernel<BM, BN, WM, WN>;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    FGDM_LAUNCH(k, dim3(ntm * ntn), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

// Tile selection.  Large layers whose N is a multiple of 320 (every SD-v1 width) or, for GEGLU, of 256 go to the
// pipelined big-tile kernel (igemm2.hip); everything else (N = 4 output conv, hint block, tiny M) uses the
// 2-stage kernel above with the tile of lowest modelled time.
static int g_force_cfg = 0;
void igemm_set_force_cfg(int cfg) { g_force_cfg = cfg; }
// > 1 while the engine records a walk whose GEMM launches will be fused with their twins of another net (common.h, "deferred
// launches"): the grid a tile choice is judged by is that many times the single problem's
static thread_local int g_pair_mult = 1;
void igemm_set_pair_hint(int mult) { g_pair_mult = mult > 1 ? mult : 1; }

// tile configuration for `a` (0 = the 2-stage kernel picks one of its own tiles; 4 + c = pipelined kernel cfg c: tile c & 15 =
// 0..6, K loop c >> 4, see igemm2_launch)
static int pick_force(const IgemmArgs& a) {
    int force = a.force_cfg ? a.force_cfg : g_force_cfg;
    if (a.act == ACT_QGELU) {                   // the pipelined kernel's epilogue does not carry quick-GELU (text encoder only)
        if (force >= 4) force = 0;
        if (force == 0) return 0;
    }
    static const bool small_tiles = !(getenv("FGDM_IGEMM_SMALL_TILES") && atoi(getenv("FGDM_IGEMM_SMALL_TILES")) == 0);        // A/B knob
    static const bool other_widths = !(getenv("FGDM_IGEMM_OTHER_WIDTHS") && atoi(getenv("FGDM_IGEMM_OTHER_WIDTHS")) == 0);   // A/B knob
    if (force == 0) {
        const bool geglu = a.act == ACT_GEGLU;
        if (!geglu && a.N % 320 == 0) {
            const long b256 = (long)((a.M + 255) / 256) * (a.N / 320);
            const long b128 = (long)((a.M + 127) / 128) * (a.N / 320);
            // 64x160 four-wave tiles (two workgroups per CU) for the linears whose 128-row grid cannot fill the chip: the 8x8
            // level's (M = 64 rows per sample), and (round 4: FGDM_IGEMM_SMALL_M=0 switches it off) those whose 128 x 320 grid --
            // times the number of problems a grouped launch will carry -- stays under three quarters of the CUs while the 64 x 160
            // grid fills every CU twice: the 16x16 level at 8 prompts per GPU (M = 4096: 128 tiles of 128 x 320, 512 of 64 x 160)
            static const bool small_m = !(getenv("FGDM_IGEMM_SMALL_M") && atoi(getenv("FGDM_IGEMM_SMALL_M")) == 0);              // A/B knob
            const bool lin64 = a.mode == IG_LINEAR && !(a.K & 31) && !(a.C0 & 31) && !(a.C1 & 31) && small_tiles;
            const long b64 = (long)((a.M + 63) / 64) * (a.N / 160);
            // ... except the long-K ones among them (the 16x16 level's feed-forward output, K = 5120, at 8 prompts per GPU): 64 x 160 tiles
            // move too many L2 -> LDS bytes per FLOP there; 160 tiles of 256 x 128 run it in 79 us (64 x 160: 95, 128 x 320: 86)
            const long b256n = (long)((a.M + 255) / 256) * (a.N / 128);
            if (b256 * g_pair_mult >= 192) force = 4;
            else if (small_m && lin64 && a.K >= 2560 && !(a.N % 128) && !a.ln_stats && !a.stats_out && !a.out2 && b128 >= 96 && b128 * g_pair_mult < 192 &&
                     b64 >= 512 && b256n >= 144 && b256n <= 256) force = 10;
            else if (b128 >= 96 && !(small_m && lin64 && b128 * g_pair_mult < 192 && b64 >= 512)) force = 6;
            else if (lin64 && b64 >= 128) force = 11;
        } else if (geglu && a.N % 256 == 0 && (long)((a.M + 255) / 256) * (a.N / 256) >= 128) {
            force = 5;
        } else if (!geglu && !a.ln_stats && !(a.K & 31) && !(a.C0 & 31) && !(a.C1 & 31) && other_widths) {
            // widths that are not multiples of 320 (the autoencoder's 128 / 256 / 512, the hint block's 256): 256-row tiles
            // over 256 or 128 columns once the grid fills the chip (at 210 TFLOP/s the 2-stage kernel was the decoder's bound)
            const long rows = (a.M + 255) / 256;
            if (a.N % 256 == 0 && rows * (a.N / 256) >= 192) force = 5;
            else if (a.N == 128 && rows >= 192) force = 10;
        }
    }
    return force;
}

// Will igemm_launch(a) write LayerNorm partial sums into a.stats_out, and how many slots per row?  Only the pipelined
// kernel's 320-column tiles (two 160-column wave tiles each) and its 64 x 160 tile do; otherwise 0 and the caller runs row_stats_launch.
int igemm_stats_slots(const IgemmArgs& a) {
    if (a.out_kind != OUT_F16 || a.act == ACT_GEGLU || a.splitk > 1 || (a.N % 320) || (a.K & 31) || (a.C0 & 31) || (a.C1 & 31)) return 0;
    const int f = pick_force(a);
    if (f < 4) return 0;
    const int tile = (f - 4) & 15;
    // (tile 7, 64 x 160: its two 80-column waves per row meet in LDS and write the same 160-column slots)
    static const bool stats64 = !(getenv("FGDM_IGEMM_STATS64") && atoi(getenv("FGDM_IGEMM_STATS64")) == 0);       // A/B knob (same bits either way)
    return (tile == 0 || tile == 2 || tile == 3 || tile == 5 || (tile == 7 && stats64)) ? a.N / 160 : 0;
}

int igemm_launch(const IgemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K & 63) || (a.C0 & 63) || (a.C1 & 63)) return FGDM_ERR_ARG;
    if (a.act == ACT_GEGLU && (a.N & 63)) return FGDM_ERR_ARG;
    if (a.ln_stats && (a.mode != IG_LINEAR || a.C1 || a.ln_slots <= 0 || !a.ln_u)) return FGDM_ERR_ARG;
    if (a.stats_out && igemm_stats_slots(a) == 0) return FGDM_ERR_ARG;
    if (a.out2 && (a.act == ACT_GEGLU || a.splitk > 1 || a.stats_out || a.split_n <= 0 || a.split_n >= a.N || (a.split_n % 640))) return FGDM_ERR_ARG;
    if (a.splitk > 1) {        // split-K plan made by the caller (igemm_splitk_factor): 128x320 tiles + reduction pass
        if (!a.ws || a.ln_stats) return FGDM_ERR_ARG;
        // 256 x 320 tiles once they fill the chip (eight-way splits at B = 32), else 128 x 320: the tile changes no sum's order
        const bool fat = (long)((a.M + 255) / 256) * (a.N / 320) * a.splitk >= 192 && !(a.K & 31) && !(a.C0 & 31) && !(a.C1 & 31);
        const int rc = igemm2_launch(a, fat ? 0 : 2, s);
        return rc == FGDM_OK ? igemm_splitk_reduce(a, s) : rc;
    }
    int force = pick_force(a);
    // diagnostic knob for A/B runs (tools/ab_bench.sh): automatic choices take the 32x32x16 instantiations
    static const bool mfma32 = getenv("FGDM_IGEMM_MFMA32") && atoi(getenv("FGDM_IGEMM_MFMA32")) != 0;
    if (mfma32 && force >= 4 && force <= 6 && !a.force_cfg && !g_force_cfg) force += 3;
    if (force >= 4) return igemm2_launch(a, force - 4, s);
    struct Cfg { int bm, bn; float eff; int per_cu; };
    static const Cfg cfgs[] = {{128, 128, 1.00f, 2}, {128, 64, 0.80f, 3}, {64, 64, 0.62f, 4}};
    int best = 0;
    float best_cost = 1e30f;
    for (int i = 0; i < 3; ++i) {
        const Cfg& c = cfgs[i];
        const long nblk = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
        const long slots = 256L * c.per_cu;
        const float rounds = (float)((nblk + slots - 1) / slots);
        const long conc = std::min<long>(c.per_cu, (nblk + 255) / 256);
        const float cost = rounds * (float)conc * (float)(c.bm * c.bn) / c.eff;
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    // a grid that cannot fill the chip with 128x128 tiles runs faster on 64x64 tiles (measured: M = 2048 layers)
    if ((long)((a.M + 127) / 128) * ((a.N + 127) / 128) < 256) best = 2;
    if (a.act == ACT_GEGLU) best = 0;   // value/gate pairs must sit in one wave's 64-column tile
    if (force >= 1 && force <= 3) best = force - 1;
    switch (best) {
        case 0: return launch_cfg<128, 128, 2, 2>(a, s);
        case 1: return launch_cfg<128, 64, 2, 2>(a, s);
        default: return launch_cfg<64, 64, 2, 2>(a, s);
    }
}
