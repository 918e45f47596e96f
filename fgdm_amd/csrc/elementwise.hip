// Small HBM-bound kernels of the hot path: layout/dtype conversion at the API boundary, im2col for the
// few convolutions whose Cin is not a multiple of 64, 2x2 average pooling (FG-DM adapter), sinusoidal
// timestep embedding, and the fused sampler updates (CFG combine + DDIM / PLMS / ancestral step).
#include "common.h"

#define EW_BLOCK 256
static inline int ew_grid(size_t n) {
    size_t g = (n + EW_BLOCK - 1) / EW_BLOCK;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
#define EW_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < (n); i += (size_t)gridDim.x * EW_BLOCK)
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP)

// x [B, C, HW] fp32  ->  y [B, HW, Cpad] fp16 (channels >= C zero)
__global__ void k_nchw_to_nhwc(const float* __restrict__ x, half_t* __restrict__ y, int B, int C, int HW, int Cpad) {
    const size_t n = (size_t)B * HW * Cpad;
    EW_LOOP(i, n) {
        const int c = (int)(i % Cpad);
        const size_t bp = i / Cpad;
        const size_t b = bp / HW, p = bp - b * HW;
        y[i] = c < C ? (half_t)x[(b * C + c) * HW + p] : (half_t)0;
    }
}
int nchw_f32_to_nhwc_f16(const float* x, half_t* y, int B, int C, int HW, int Cpad, hipStream_t s) {
    FGDM_LAUNCH(k_nchw_to_nhwc, dim3(ew_grid((size_t)B * HW * Cpad)), dim3(EW_BLOCK), 0, s, x, y, B, C, HW, Cpad);
    return LAUNCH_OK();
}

__global__ void k_f32_to_f16(const float* __restrict__ x, half_t* __restrict__ y, size_t n) {
    EW_LOOP(i, n) y[i] = (half_t)x[i];
}
int f32_to_f16(const float* x, half_t* y, size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_f32_to_f16, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, x, y, n);
    return LAUNCH_OK();
}
__global__ void k_nhwc_f16_to_nchw_f32(const half_t* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
    const size_t n = (size_t)B * C * HW;
    EW_LOOP(i, n) {
        const int p = (int)(i % HW);
        const size_t bc = i / HW;
        const int c = (int)(bc % C);
        const size_t b = bc / C;
        y[i] = (float)x[(b * HW + p) * C + c];
    }
}
int nhwc_f16_to_nchw_f32(const half_t* x, float* y, int B, int C, int HW, hipStream_t s) {
    FGDM_LAUNCH(k_nhwc_f16_to_nchw_f32, dim3(ew_grid((size_t)B * C * HW)), dim3(EW_BLOCK), 0, s, x, y, B, C, HW);
    return LAUNCH_OK();
}
// y = fp16(SiLU(x)): the `emb_layers` input of a ResBlock (openaimodel.py:238-244) when `emb` is handed in from outside
__global__ void k_silu_f32_to_f16(const float* __restrict__ x, half_t* __restrict__ y, size_t n) {
    EW_LOOP(i, n) { const float v = x[i]; y[i] = (half_t)silu_f(v); }
}
int silu_f32_to_f16(const float* x, half_t* y, size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_silu_f32_to_f16, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, x, y, n);
    return LAUNCH_OK();
}

// A[m][k], k = tap * C + c (k < 9C), zero for k in [9C, Kpad); 3x3 window, pad 1, given stride.
// G = channels per thread-granule (8 when C % 8 == 0, else 4 for C == 4)
template <int G>
__global__ void k_im2col(const half_t* __restrict__ x, half_t* __restrict__ A, int B, int H, int W, int C,
                         int stride, int Ho, int Wo, int Kpad) {
    typedef half_t vec_t __attribute__((ext_vector_type(G)));
    const int KG = Kpad / G;
    const size_t n = (size_t)B * Ho * Wo * KG;
    EW_LOOP(i, n) {
        const int kg = (int)(i % KG);
        const size_t m = i / KG;
        const int k = kg * G;
        vec_t v = (vec_t)(half_t)0;
        if (k < 9 * C) {
            const int tap = k / C, c = k - tap * C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int hw = Ho * Wo;
            const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw);
            const int oy = rem / Wo, ox = rem - oy * Wo;
            const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *(const vec_t*)(x + (((size_t)b * H + iy) * W + ix) * C + c);
        }
        *(vec_t*)(A + m * Kpad + k) = v;
    }
}
int im2col3x3(const half_t* x, half_t* A, int B, int H, int W, int C, int stride, int Kpad, hipStream_t s) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    if (Kpad < 9 * C || (Kpad & 63)) return FGDM_ERR_ARG;
    if ((C & 7) == 0) {
        FGDM_LAUNCH(k_im2col<8>, dim3(ew_grid((size_t)B * Ho * Wo * (Kpad / 8))), dim3(EW_BLOCK), 0, s, x, A, B,
                           H, W, C, stride, Ho, Wo, Kpad);
    } else if ((C & 3) == 0) {
        FGDM_LAUNCH(k_im2col<4>, dim3(ew_grid((size_t)B * Ho * Wo * (Kpad / 4))), dim3(EW_BLOCK), 0, s, x, A, B,
                           H, W, C, stride, Ho, Wo, Kpad);
    } else {
        return FGDM_ERR_ARG;
    }
    return LAUNCH_OK();
}

// AvgPool2d(2) on NHWC fp16 (ldm/modules/encoders/adapter.py:270-273, use_conv=False)
__global__ void k_avgpool2(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, P = C / 8;
    const size_t n = (size_t)B * Ho * Wo * P;
    EW_LOOP(i, n) {
        const int o = (int)(i % P);
        const size_t m = i / P;
        const int hw = Ho * Wo;
        const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const half_t* p = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + o * 8;
        const h8 a = *(const h8*)p, bq = *(const h8*)(p + C), c = *(const h8*)(p + (size_t)W * C),
                 d = *(const h8*)(p + (size_t)W * C + C);
        h8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (half_t)(0.25f * ((float)a[e] + (float)bq[e] + (float)c[e] + (float)d[e]));
        *(h8*)(y + m * C + o * 8) = r;
    }
}
int avgpool2(const half_t* x, half_t* y, int B, int H, int W, int C, hipStream_t s) {
    if ((C & 7) || (H & 1) || (W & 1)) return FGDM_ERR_ARG;
    FGDM_LAUNCH(k_avgpool2, dim3(ew_grid((size_t)B * (H / 2) * (W / 2) * (C / 8))), dim3(EW_BLOCK), 0, s, x, y, B, H, W, C);
    return LAUNCH_OK();
}

__global__ void k_add_f16(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ y, size_t n8) {
    EW_LOOP(i, n8) {
        const h8 u = ((const h8*)a)[i], v = ((const h8*)b)[i];
        h8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (half_t)((float)u[e] + (float)v[e]);
        ((h8*)y)[i] = r;
    }
}
int add_f16(const half_t* a, const half_t* b, half_t* y, size_t n, hipStream_t s) {
    if (n & 7) return FGDM_ERR_ARG;
    FGDM_LAUNCH(k_add_f16, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, s, a, b, y, n / 8);
    return LAUNCH_OK();
}

// y[b] = [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half)   (util.py:160-180); rows >= B are zero
// y = a + float(b): a fp16 branch output added to the fp32 residual stream (CLIP's hidden states under autocast promote)
__global__ void k_add_f16_to_f32(const float* __restrict__ a, const half_t* __restrict__ b, float* __restrict__ y, size_t n) {
    EW_LOOP(i, n) y[i] = a[i] + (float)b[i];
}
int add_f16_to_f32(const float* a, const half_t* b, float* y, size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_add_f16_to_f32, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, a, b, y, n);
    return LAUNCH_OK();
}

__global__ void k_timestep_embed(const int64_t* __restrict__ t, const float* __restrict__ tf, half_t* __restrict__ y,
                                 int B, int dim, int rows_pad) {
    const int half = dim / 2;
    const size_t n = (size_t)rows_pad * dim;
    EW_LOOP(i, n) {
        const int b = (int)(i / dim), j = (int)(i % dim);
        float v = 0.f;
        if (b < B) {
            const int k = j < half ? j : j - half;
            const float f = expf(-9.210340371976184f * (float)k / (float)half);
            const float arg = (tf ? tf[b] : (float)t[b]) * f;      // "These may be fractional" (util.py:165): DPM-Solver
            v = j < half ? cosf(arg) : sinf(arg);
        }
        y[i] = (half_t)v;
    }
}
int timestep_embed(const int64_t* t, const float* tf, half_t* y, int B, int dim, int rows_pad, hipStream_t s) {
    FGDM_LAUNCH(k_timestep_embed, dim3(ew_grid((size_t)rows_pad * dim)), dim3(EW_BLOCK), 0, s, t, tf, y, B, dim, rows_pad);
    return LAUNCH_OK();
}

// v [B, Tk, C] -> vt [B, C, Tkpad] (zero for t >= Tk); used once per sample() call for the context V of cross-attention
__global__ void k_transpose_pad(const half_t* __restrict__ v, half_t* __restrict__ vt, int B, int Tk, int C, int Tkpad) {
    const size_t n = (size_t)B * C * Tkpad;
    EW_LOOP(i, n) {
        const int t = (int)(i % Tkpad);
        const size_t bc = i / Tkpad;
        const size_t b = bc / C, c = bc - b * C;
        vt[i] = t < Tk ? v[(b * Tk + t) * C + c] : (half_t)0;
    }
}
int transpose_pad_keys(const half_t* v, half_t* vt, int B, int Tk, int C, int Tkpad, hipStream_t s) {
    FGDM_LAUNCH(k_transpose_pad, dim3(ew_grid((size_t)B * C * Tkpad)), dim3(EW_BLOCK), 0, s, v, vt, B, Tk, C, Tkpad);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------- sampler updates (fp32, one fused pass)
// e = e_u + s (e_c - e_u)                                        ldm/models/diffusion/ddim.py:243
// pred_x0 = (x - sqrt(1-a_t) e) / sqrt(a_t)                      ddim.py:259
// x_prev  = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noise     ddim.py:263-268
__global__ void k_ddim_step(const float* __restrict__ x, const float* __restrict__ ec, const float* __restrict__ eu,
                            float cfg, float a_t, float a_prev, float sigma, float s1m, const float* __restrict__ noise,
                            float* __restrict__ x_prev, float* __restrict__ pred_x0, float* __restrict__ e_out, size_t n) {
    const float sa = sqrtf(a_t), sp = sqrtf(a_prev), dir = sqrtf(1.0f - a_prev - sigma * sigma);
    EW_LOOP(i, n) {
        float e = ec[i];
        if (eu) { const float u = eu[i]; e = u + cfg * (e - u); }
        const float xi = x[i];
        const float p0 = (xi - s1m * e) / sa;
        float xp = sp * p0 + dir * e;
        if (noise) xp += sigma * noise[i];
        if (x_prev) x_prev[i] = xp;
        if (pred_x0) pred_x0[i] = p0;
        if (e_out) e_out[i] = e;
    }
}
int ddim_step(const float* x, const float* e_cond, const float* e_uncond, float cfg_scale, float a_t, float a_prev,
              float sigma_t, float sqrt_one_minus_at, const float* noise, float* x_prev, float* pred_x0, float* e_out,
              size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_ddim_step, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, x, e_cond, e_uncond, cfg_scale, a_t, a_prev,
                       sigma_t, sqrt_one_minus_at, noise, x_prev, pred_x0, e_out, n);
    return LAUNCH_OK();
}

// Adams-Bashforth combinations of plms.py:224-232; order = number of old eps available (1..3)
__global__ void k_plms(const float* __restrict__ e, const float* __restrict__ e1, const float* __restrict__ e2,
                       const float* __restrict__ e3, int order, float* __restrict__ out, size_t n) {
    EW_LOOP(i, n) {
        float r;
        if (order == 1) r = (3.f * e[i] - e1[i]) / 2.f;
        else if (order == 2) r = (23.f * e[i] - 16.f * e1[i] + 5.f * e2[i]) / 12.f;
        else r = (55.f * e[i] - 59.f * e1[i] + 37.f * e2[i] - 9.f * e3[i]) / 24.f;
        out[i] = r;
    }
}
int plms_combine(const float* e_t, const float* e1, const float* e2, const float* e3, int order, float* e_prime,
                 size_t n, hipStream_t s) {
    if (order < 1 || order > 3) return FGDM_ERR_ARG;
    FGDM_LAUNCH(k_plms, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, e_t, e1, e2, e3, order, e_prime, n);
    return LAUNCH_OK();
}

__global__ void k_axpby(const float* __restrict__ a, float ca, const float* __restrict__ b, float cb,
                        float* __restrict__ y, size_t n) {
    EW_LOOP(i, n) y[i] = ca * a[i] + (b ? cb * b[i] : 0.f);
}
int axpby(const float* a, float ca, const float* b, float cb, float* y, size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_axpby, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, a, ca, b, cb, y, n);
    return LAUNCH_OK();
}

// inpainting blend img_orig * mask + (1 - mask) * img (ddim.py:151-154, ddpm.py:1419-1421); mask pre-expanded
__global__ void k_mask_blend(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m,
                             float* __restrict__ y, size_t n) {
    EW_LOOP(i, n) y[i] = a[i] * m[i] + (1.0f - m[i]) * b[i];
}
int mask_blend(const float* a, const float* b, const float* m, float* y, size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_mask_blend, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, a, b, m, y, n);
    return LAUNCH_OK();
}

// ancestral step (ddpm.py:284-297, 1318-1323): x0 = c_r x - c_rm1 eps; mean = c1 x0 + c2 x; out = mean + std * noise
__global__ void k_ancestral(const float* __restrict__ x, const float* __restrict__ eps, float cr, float crm1, float c1,
                            float c2, float std, const float* __restrict__ noise, float* __restrict__ out, size_t n) {
    EW_LOOP(i, n) {
        const float x0 = cr * x[i] - crm1 * eps[i];
        float r = c1 * x0 + c2 * x[i];
        if (noise) r += std * noise[i];
        out[i] = r;
    }
}
int ancestral_step(const float* x, const float* eps, float sqrt_recip, float sqrt_recipm1, float coef1, float coef2,
                   float std, const float* noise, float* out, size_t n, hipStream_t s) {
    FGDM_LAUNCH(k_ancestral, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, x, eps, sqrt_recip, sqrt_recipm1, coef1,
                       coef2, std, noise, out, n);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------- first-stage decoder helpers
// post_quant_conv (1x1, 4 -> 4; ldm/models/autoencoder.py:331) applied to scale * z, fused with the layout change:
// z [B, 4, HW] fp32  ->  y [B, HW, 4] fp16.   wb = 16 weights [co][ci] then 4 biases (device, fp32)
__global__ void k_vae_prequant(const float* __restrict__ z, const float* __restrict__ wb, float scale,
                               half_t* __restrict__ y, int B, int HW) {
    const size_t n = (size_t)B * HW;
    EW_LOOP(i, n) {
        const size_t b = i / HW, p = i - b * HW;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = scale * z[(b * 4 + c) * HW + p];
        h4 o;
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            float acc = wb[16 + co];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc += wb[co * 4 + ci] * v[ci];
            o[co] = (half_t)acc;
        }
        *(h4*)(y + i * 4) = o;
    }
}
int vae_prequant(const float* z, const float* wb, float scale, half_t* y, int B, int HW, hipStream_t s) {
    FGDM_LAUNCH(k_vae_prequant, dim3(ew_grid((size_t)B * HW)), dim3(EW_BLOCK), 0, s, z, wb, scale, y, B, HW);
    return LAUNCH_OK();
}

// P[r][:] = softmax(S[r][:]) over `cols` fp32 logits -> fp16 probabilities; one 256-thread block per row
// (AttnBlock, ldm/modules/diffusionmodules/model.py:190-192: single head, the T x T score matrix of one image)
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ S, half_t* __restrict__ P, int cols) {
    __shared__ float red[8];
    const float* row = S + (size_t)blockIdx.x * cols;
    half_t* out = P + (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, row[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = tid; c < cols; c += 256) sum += __expf(row[c] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int c = tid; c < cols; c += 256) out[c] = (half_t)(__expf(row[c] - m) * inv);
}
int softmax_rows(const float* S, half_t* P, int rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return FGDM_ERR_ARG;
    FGDM_LAUNCH(k_softmax_rows, dim3(rows), dim3(256), 0, s, S, P, cols);
    return LAUNCH_OK();
}
