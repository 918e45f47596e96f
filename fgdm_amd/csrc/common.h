// Shared declarations for the FG-DM HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// exact-erf GELU (ldm/modules/attention.py:43-44: F.gelu) as x Phi(x) with the lower Gaussian tail t = Phi(-|x|) written as
// exp2 of a degree-5 polynomial in z = min(|x|, 6) (minimax fit of z |t^ - t|: |gelu^ - gelu| <= 8.2e-7 in fp32 over all x, far
// below the fp16 resolution of the output), and x Phi(x) = x / 2 + |x| (1/2 - t): no reciprocal, no compare, 12 issue slots
// per value where the Abramowitz-Stegun erfc form took 21 -- the GEGLU epilogue evaluates this 64 times per lane and tile.
// One definition for both GEMM kernels: which of them runs a layer must not change its output.
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fminf(fabsf(x), 6.0f);
    float p = -4.7132482596e-04f;
    p = __builtin_fmaf(p, z, 7.0652551839e-03f);
    p = __builtin_fmaf(p, z, -5.1762067461e-02f);
    p = __builtin_fmaf(p, z, -4.6008771426e-01f);
    p = __builtin_fmaf(p, z, -1.1507295505e+00f);
    p = __builtin_fmaf(p, z, -4.8959157159e-05f - 1.0f);       // log2 t
    const float t = __builtin_amdgcn_exp2f(p);
    return __builtin_fmaf(fabsf(x), 0.5f - t, 0.5f * x);
}

// Two values per call, the SAME operations in the same order (bit-identical to gelu_f), written on 2-vectors so that the polynomial,
// the final fma and 0.5 x issue as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): 9 issue slots per value instead of 12.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu2_f(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = {fminf(ax[0], 6.0f), fminf(ax[1], 6.0f)};
    auto c = [](float k) { return f32x2{k, k}; };
    f32x2 p = c(-4.7132482596e-04f);
    p = __builtin_elementwise_fma(p, z, c(7.0652551839e-03f));
    p = __builtin_elementwise_fma(p, z, c(-5.1762067461e-02f));
    p = __builtin_elementwise_fma(p, z, c(-4.6008771426e-01f));
    p = __builtin_elementwise_fma(p, z, c(-1.1507295505e+00f));
    p = __builtin_elementwise_fma(p, z, c(-4.8959157159e-05f - 1.0f));
    const f32x2 t = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    return __builtin_elementwise_fma(ax, c(0.5f) - t, c(0.5f) * x);
}

// x sigmoid(k x) (SiLU: k = 1; CLIP's quick-GELU: k = 1.702) with exp2 and the hardware reciprocal (1 ulp): the IEEE division
// `x / (1 + exp(-x))` expands to ten instructions per value and made the GroupNorm+SiLU pass VALU-bound.  Shared by every kernel.
__device__ __forceinline__ float silu_f(float x, float k = 1.0f) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * k * x));
}

#define FGDM_OK 0
#define FGDM_ERR_ARG -1
#define FGDM_ERR_HIP -2
#define FGDM_ERR_STATE -3
#define FGDM_ERR_NOMEM -4

// ---------------------------------------------------------------- implicit GEMM
// out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),  m = (b, oy, ox) output pixel, k = (tap, cin)
enum IgemmMode { IG_LINEAR = 0, IG_CONV3 = 1, IG_CONV3_S2 = 2, IG_CONV3_UP2 = 3 };
enum IgemmAct { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_GEGLU = 3, ACT_QGELU = 4 /* x * sigmoid(1.702 x), CLIP MLP */ };
enum IgemmOut { OUT_F16 = 0, OUT_F32 = 1, OUT_F32_NCHW = 2, OUT_F16_T = 3 };

struct IgemmArgs {
    const half_t* A0;      // NHWC fp16 source 0 [B, H, W, C0]  (LINEAR: [M, C0])
    const half_t* A1;      // optional source 1 (virtual channel concat) [B, H, W, C1]
    const half_t* Wt;      // packed weights [Npad][K], K = taps * (C0 + C1), k = tap * Ctot + c
    const float* bias;     // [N] (packed order) or null
    const float* rowvec;   // per-sample vector added before the activation: rowvec[b * rv_stride + n], or null
    const half_t* resid;   // [M, ld_res] fp16 added after scale, or null (may alias out)
    void* out;
    const half_t* zero;    // >= 256 B of zeros (source for padded / out-of-range rows)
    int C0, C1;
    int B, H, W;           // input spatial dims (UP2: dims before the nearest-2x upsample)
    int Ho, Wo;            // output spatial dims
    int M, N;              // M = B*Ho*Wo ; N = number of valid packed output columns (GEGLU: 2 * N_out)
    int K;                 // taps * (C0 + C1)
    int mode, act, out_kind;
    int ld_out, ld_res;    // row strides (elements) of out / resid
    int rv_stride;
    int rows_per_sample;   // Ho*Wo (LINEAR: tokens per sample) -> sample index b = m / rows_per_sample
    float scale;           // (acc + bias + rowvec -> act) * scale + resid
    int force_cfg;         // 0 = pick automatically; k > 0 = tile configuration k-1 (tuning / benchmarks)
    int splitk;            // > 1: K is split over `splitk` workgroups per tile; fp32 partial tiles go to `ws`
    float* ws;             //      [splitk][M][N] and igemm_splitk_reduce applies the epilogue (deterministic order)
    int debug;             // ablation switches, honoured by the ABL instantiations tools/bench_igemm.py asks for and rejected everywhere
                           // else: 1 = no global->LDS loads in the K loop, 4 = no epilogue, 8 = every load from the zero page,
                           // 16 = no tap / channel walk (pipelined loop; results are then meaningless)
    // ---- LayerNorm folded into a LINEAR GEMM (attention.py:234-240: attn(norm(x)), ff(norm(x))).  A is the RAW row x; the
    // packed weights are W'[n][k] = fp16(gamma[k] W[n][k]), `bias` holds sum_k beta[k] W[n][k] + b[n], `ln_u` holds
    // sum_k W'[n][k]; with (mean, rstd) of row m from the producer's partial sums the epilogue evaluates
    //     rstd_m (acc_mn - mean_m u_n) + bias_n   ==   sum_k ((x_mk - mean_m) rstd_m gamma_k + beta_k) W_nk + b_n
    // so no normalised copy of x is ever written or read.
    const float* ln_stats; // [M][ln_slots][2] partial (sum, sum of squares) over column slices of row m, or null
    const float* ln_u;     // [N] (packed order)
    int ln_slots;
    float ln_eps;          // (K = C0 is the length of the normalised vector)
    // ---- ... and the producer side: emit those partial sums for the rows this GEMM writes (OUT_F16 only), computed from the
    // stored fp16 values (after the residual add), one slot per 160-column slice, fixed order: stats_out[m][N / 160][2]
    float* stats_out;
    // ---- second destination: packed output columns >= split_n go to out2 (own kind / row stride, columns renumbered from 0);
    // tiles never straddle split_n.  attn1's to_q | to_k | to_v as ONE GEMM: q | k row-major for the attention kernel,
    // v transposed (OUT_F16_T) as its PV operand (ldm/modules/attention.py:180-186)
    void* out2;
    int split_n, out_kind2, ld_out2;
};

int igemm_launch(const IgemmArgs& a, hipStream_t s);
void igemm_set_force_cfg(int cfg);   // process-wide override of the tile choice (0 = automatic)
void igemm_set_twin_layers(bool on);  // the layers walked from now on have a same-shape twin in another net of this engine (UNet
                                       // encoder / middle block and the ControlNets): igemm_splitk_factor leaves their K = 11520
                                       // convolutions of the 16x16 level whole, whichever way the twins are launched
void igemm_set_pair_hint(int mult);  // launches recorded from now on will be fused `mult` at a time (grid size for the tile choice)
int igemm2_launch(const IgemmArgs& a, int cfg, hipStream_t s);
int igemm_stats_slots(const IgemmArgs& a);    // > 0: igemm_launch(a) will fill a.stats_out with that many slots per row; 0: it cannot
int row_stats_slots(int C);                   // slots per row the separate pass writes: C / 160 (the epilogue's layout) or 1
int row_stats_launch(const half_t* x, int rows, int C, float* stats /* [rows][row_stats_slots(C)][2] */, hipStream_t s);
int igemm_splitk_factor(const IgemmArgs& a);                    // 1 = no split; else the engine must provide a.ws
int igemm_splitk_reduce(const IgemmArgs& a, hipStream_t s);     // out = epilogue(sum_s ws[s])   // pipelined big-tile kernel (igemm2.hip)
size_t igemm_npad(int n);   // rows the packed weight must provide

// ---------------------------------------------------------------- norms
// GroupNorm(32 groups) over NHWC fp16, optionally over the virtual concat of two sources; fp32 statistics.
void groupnorm_set_group(bool on);     // while the engine records: attach the grouped form to single-pass GroupNorm launches (default on)
int groupnorm_launch(const half_t* x0, int C0, const half_t* x1, int C1, int B, int HW,
                     const float* gamma, const float* beta, float eps, int silu,
                     half_t* out, float* ws /* >= B*32*2*(chunks+1) floats */, hipStream_t s);
size_t groupnorm_ws_floats(int B, int HW);
int layernorm_launch(const half_t* x, int rows, int C, const float* gamma, const float* beta, float eps,
                     half_t* out, hipStream_t s);

// ---------------------------------------------------------------- attention
// O[b, t, h*d + :] = softmax(Q K^T * d^-1/2) V ; Q [B, T, ldq], K [B, Tk, ldk], Vt [B, H*d, ldvt] (keys contiguous)
int attention_launch(const half_t* Q, int ldq, const half_t* K, int ldk, const half_t* Vt, int ldvt,
                     half_t* O, int ldo, int B, int H, int T, int Tk, int d, int q_prescaled, hipStream_t s);

// ---------------------------------------------------------------- elementwise
int nchw_f32_to_nhwc_f16(const float* x, half_t* y, int B, int C, int HW, int Cpad, hipStream_t s);
int f32_to_f16(const float* x, half_t* y, size_t n, hipStream_t s);
int silu_f32_to_f16(const float* x, half_t* y, size_t n, hipStream_t s);
int nhwc_f16_to_nchw_f32(const half_t* x, float* y, int B, int C, int HW, hipStream_t s);
int im2col3x3(const half_t* x, half_t* A, int B, int H, int W, int C, int stride, int Kpad, hipStream_t s);
int avgpool2(const half_t* x, half_t* y, int B, int H, int W, int C, hipStream_t s);
int add_f16(const half_t* a, const half_t* b, half_t* y, size_t n, hipStream_t s);
int add_f16_to_f32(const float* a, const half_t* b, float* y, size_t n, hipStream_t s);   // y = a + float(b): fp32 residual stream
int timestep_embed(const int64_t* t, const float* t_float, half_t* y, int B, int dim, int rows_pad, hipStream_t s);
int vae_prequant(const float* z, const float* wb, float scale, half_t* y, int B, int HW, hipStream_t s);
int softmax_rows(const float* S, half_t* P, int rows, int cols, hipStream_t s);
int image_to_u8(const float* x, uint8_t* y, int B, int C, int HW, int mode, hipStream_t s);
int resize_linear_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s);
int u8_to_hint(const uint8_t* src, float* dst, int B, int HW, int C, hipStream_t s);
int embed_tokens(const int64_t* ids, const float* tok, const float* pos, float* out, int rows, int T, int W, int vocab,
                 hipStream_t s);
// LayerNorm of an fp32 matrix -> fp16 (out16) or fp32 (out32), exactly one of them non-null
int layernorm32_launch(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, half_t* out16,
                       float* out32, hipStream_t s);
int f16_to_f32(const half_t* x, float* y, size_t n, hipStream_t s);
// causal / full attention over short sequences (T <= 128, d = 64) with q|k|v as column blocks of one matrix
int small_attention_launch(const half_t* qkv, int ld, int koff, int voff, half_t* out, int ldo, int B, int heads, int T,
                           int d, int causal, hipStream_t s);
int transpose_pad_keys(const half_t* v, half_t* vt, int B, int Tk, int C, int Tkpad, hipStream_t s);
// x_prev, pred_x0 from eps (with CFG combine when e_uncond != null); all fp32 NCHW
int ddim_step(const float* x, const float* e_cond, const float* e_uncond, float cfg_scale,
              float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, const float* noise,
              float* x_prev, float* pred_x0, float* e_out, size_t n, hipStream_t s);
int plms_combine(const float* e_t, const float* e1, const float* e2, const float* e3, int order,
                 float* e_prime, size_t n, hipStream_t s);
int axpby(const float* a, float ca, const float* b, float cb, float* y, size_t n, hipStream_t s);
int mask_blend(const float* a, const float* b, const float* m, float* y, size_t n, hipStream_t s);
int ancestral_step(const float* x, const float* eps, float sqrt_recip, float sqrt_recipm1, float coef1, float coef2,
                   float std, const float* noise, float* out, size_t n, hipStream_t s);

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return FGDM_ERR_HIP; } while (0)

// ---------------------------------------------------------------- deferred launches (twin-layer grouped launches)
// While the engine RECORDS (engine.hip: apply_model with FGDM_PAIR_LAUNCH), nothing is enqueued: every launch site goes through
// FGDM_LAUNCH, which then stores a closure (arguments by value) in the engine's list instead; the engine replays the lists of the
// UNet encoder and of the ControlNets in lockstep and fuses launches of the same pipelined-GEMM instantiation and grid into ONE
// grouped launch (igemm2.hip: igemm2_group_kernel; blockIdx.y selects the argument set: the UNet plus up to four ControlNets),
// so that the half-empty grids of the 16x16 / 8x8 levels fill the chip.  Each net's launches keep their order, so the results do
// not change by a bit.
#include <functional>
#define FGDM_MAX_GROUP 5
typedef int (*IgemmGroupFn)(const IgemmArgs* const* a, int n, unsigned grid_x, hipStream_t s);
bool fgdm_recording();
void fgdm_record(std::function<int(hipStream_t)> run, const void* pair_key = nullptr, IgemmGroupFn pair = nullptr,
                 const IgemmArgs* ia = nullptr, unsigned grid_x = 0);
// ... and the same for other kernels with twins (round 4, late: the single-pass GroupNorm kernels): a generic group function gets the
// problems' argument blobs (<= FGDM_GROUP_BLOB bytes each, copied at record time); launches are matched by key, grid and a shape hash
#define FGDM_GROUP_BLOB 96
typedef int (*GenericGroupFn)(const void* const* args, int n, unsigned grid_x, hipStream_t s);
void fgdm_record_generic(std::function<int(hipStream_t)> run, const void* key, GenericGroupFn fn, const void* args, size_t nbytes,
                         unsigned grid_x, unsigned long long shape);
#define FGDM_LAUNCH(kernel, grid, block, smem, stream, ...)                                                                  \
    do {                                                                                                                     \
        if (fgdm_recording()) {                                                                                              \
            fgdm_record([=](hipStream_t fgdm_s_) -> int {                                                                    \
                hipLaunchKernelGGL(kernel, grid, block, smem, fgdm_s_, __VA_ARGS__);                                         \
                return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;                                             \
            });                                                                                                              \
        } else {                                                                                                             \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                              \
        }                                                                                                                    \
    } while (0)
