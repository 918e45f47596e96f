// CLIP text encoder pieces that are not GEMMs / LayerNorm (FrozenCLIPEmbedder's transformers.CLIPTextModel,
// ldm/modules/encoders/modules.py:137-162): token + position embedding gather, and causal multi-head attention over
// the 77-token prompt.  The whole encoder is ~7 GFLOP per prompt and runs once per image, so the attention is a plain
// LDS-resident VALU kernel (one workgroup per (head, prompt)); the projections and the MLP go through the igemm path.
#include "common.h"

// out[b*T + t][:] = tok[ids[b][t]][:] + pos[t][:]: fp32 tables, fp32 sum, fp32 out.  nn.Embedding is not an autocast op: the
// reference's residual stream starts in fp32 and stays fp32 (fp16 branch outputs are promoted when added to it).
__global__ void k_embed_tokens(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                               const float* __restrict__ pos, float* __restrict__ out, int rows, int T, int W, int vocab) {
    const int P = W >> 2;
    const size_t n = (size_t)rows * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / P;
        const int o = (int)(i - r * P);
        long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 a = *(const f32x4*)(tok + (size_t)id * W + (o << 2));
        const f32x4 b = *(const f32x4*)(pos + (size_t)(r % T) * W + (o << 2));
        f32x4 y = {a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
        *(f32x4*)(out + r * W + (o << 2)) = y;
    }
}
int embed_tokens(const int64_t* ids, const float* tok, const float* pos, float* out, int rows, int T, int W, int vocab,
                 hipStream_t s) {
    if (rows <= 0 || T <= 0 || (W & 7) || vocab <= 0) return FGDM_ERR_ARG;
    size_t g = ((size_t)rows * (W >> 2) + 255) / 256;
    FGDM_LAUNCH(k_embed_tokens, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, ids, tok, pos, out, rows, T, W, vocab);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

__global__ void k_f16_to_f32(const half_t* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (float)x[i];
}
int f16_to_f32(const half_t* x, float* y, size_t n, hipStream_t s) {
    size_t g = (n + 255) / 256;
    FGDM_LAUNCH(k_f16_to_f32, dim3((unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g))), dim3(256), 0, s, x, y, n);
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}

// O[b, i, h*D + :] = softmax_j( scale * q_i . k_j  (j <= i when causal) ) v_j ; q/k/v are column blocks of one
// [B*T, ld] matrix (the stacked q|k|v projection).  One 256-thread workgroup per (head, prompt); T <= 128.
template <int D>
__global__ __launch_bounds__(256) void small_attn_kernel(const half_t* __restrict__ qkv, int ld, int koff, int voff,
                                                          half_t* __restrict__ out, int ldo, int T, int causal, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD = D + 2;                      // odd dword stride: conflict-free row-parallel reads
    half_t* Qs = (half_t*)smem;
    half_t* Ks = Qs + (size_t)T * LD;
    half_t* Vs = Ks + (size_t)T * LD;
    float* S = (float*)(Vs + (size_t)T * LD);      // [T][T + 1]
    const int SL = T + 1;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const half_t* base = qkv + (size_t)b * T * ld + h * D;
    for (int i = tid; i < T * (D / 2); i += 256) {
        const int r = i / (D / 2), c = (i - r * (D / 2)) * 2;
        *(h2*)(Qs + r * LD + c) = *(const h2*)(base + (size_t)r * ld + c);
        *(h2*)(Ks + r * LD + c) = *(const h2*)(base + (size_t)r * ld + koff + c);
        *(h2*)(Vs + r * LD + c) = *(const h2*)(base + (size_t)r * ld + voff + c);
    }
    __syncthreads();
    for (int idx = tid; idx < T * T; idx += 256) {
        const int i = idx / T, j = idx - i * T;
        float acc = -INFINITY;
        if (!causal || j <= i) {
            acc = 0.f;
#pragma unroll 8
            for (int c = 0; c < D; c += 2) {
                const h2 q = *(const h2*)(Qs + i * LD + c), k = *(const h2*)(Ks + j * LD + c);
                acc += (float)q[0] * (float)k[0] + (float)q[1] * (float)k[1];
            }
            acc *= scale;
        }
        S[i * SL + j] = acc;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < T; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < T; j += 64) m = fmaxf(m, S[i * SL + j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) { const float p = __expf(S[i * SL + j] - m); S[i * SL + j] = p; sum += p; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const float inv = 1.0f / sum;
        for (int j = lane; j < T; j += 64) S[i * SL + j] *= inv;
    }
    __syncthreads();
    for (int idx = tid; idx < T * D; idx += 256) {
        const int i = idx / D, c = idx - i * D;
        const int jn = causal ? i + 1 : T;
        float acc = 0.f;
        for (int j = 0; j < jn; ++j) acc += S[i * SL + j] * (float)Vs[j * LD + c];
        out[((size_t)b * T + i) * ldo + h * D + c] = (half_t)acc;
    }
}

int small_attention_launch(const half_t* qkv, int ld, int koff, int voff, half_t* out, int ldo, int B, int heads, int T,
                           int d, int causal, hipStream_t s) {
    if (d != 64 || T <= 0 || T > 128 || B <= 0 || heads <= 0) return FGDM_ERR_ARG;
    const size_t smem = (size_t)T * (64 + 2) * 3 * sizeof(half_t) + (size_t)T * (T + 1) * sizeof(float);
    auto k = small_attn_kernel<64>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 66 * 6 + 128 * 129 * 4) != hipSuccess)
            return FGDM_ERR_HIP;
        attr_set = true;
    }
    FGDM_LAUNCH(k, dim3(heads, B), dim3(256), smem, s, qkv, ld, koff, voff, out, ldo, T, causal, 1.0f / sqrtf((float)d));
    return hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP;
}
