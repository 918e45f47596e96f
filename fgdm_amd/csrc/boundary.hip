// Stage boundary of the two-factor FG-DM chain: decoded image -> uint8 -> bilinear resize -> ControlNet hint.
#include "common.h"

#define EW_BLOCK 256
static inline int ew_grid(size_t n) {
    size_t g = (n + EW_BLOCK - 1) / EW_BLOCK;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
#define EW_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < (n); i += (size_t)gridDim.x * EW_BLOCK)
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? FGDM_OK : FGDM_ERR_HIP)

// ---------------------------------------------------------------- stage boundary (decoded image -> ControlNet hint)
// Byte/integer work, HBM-bound, once per image.  THIS FILE IS COMPILED WITH -ffp-contract=off (fgdm_amd/build.py):
// every fp32 operation must be a separately rounded IEEE operation so that results are bit-identical to the
// reference's numpy/torch fp32 expressions (HIP's __fmul_rn/__fadd_rn are plain operators and are fused into FMAs
// under the -ffp-contract=fast the other sources use; a `#pragma clang fp contract(off)` does not stop that).
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
// mode 0: uint8(255 * clamp((x + 1) / 2, 0, 1))      scripts/txt2img_fgdm_inference.py:245,249-252
// mode 1: uint8(clip(x * 127.5 + 127.5, 0, 255))      controlnet/initialize_cn.py:101
__global__ void k_image_to_u8(const float* __restrict__ x, uint8_t* __restrict__ y, int B, int C, int HW, int mode) {
    const size_t n = (size_t)B * HW * C;
    EW_LOOP(i, n) {
        const int c = (int)(i % C);
        const size_t bp = i / C;
        const size_t b = bp / HW, p = bp - b * HW;
        const float v = x[(b * C + c) * HW + p];
        float r;
        if (mode == 0) {
            r = mul_rn(add_rn(v, 1.0f), 0.5f);
            r = fminf(fmaxf(r, 0.0f), 1.0f);
            r = mul_rn(255.0f, r);
        } else {
            r = add_rn(mul_rn(v, 127.5f), 127.5f);
            r = fminf(fmaxf(r, 0.0f), 255.0f);
        }
        y[i] = (uint8_t)(int)r;     // truncation, like ndarray.astype(np.uint8) on values in [0, 255]
    }
}
int image_to_u8(const float* x, uint8_t* y, int B, int C, int HW, int mode, hipStream_t s) {
    if (B <= 0 || C <= 0 || HW <= 0 || (mode != 0 && mode != 1)) return FGDM_ERR_ARG;
    FGDM_LAUNCH(k_image_to_u8, dim3(ew_grid((size_t)B * HW * C)), dim3(EW_BLOCK), 0, s, x, y, B, C, HW, mode);
    return LAUNCH_OK();
}

// cv2.resize(..., interpolation=cv2.INTER_LINEAR) on 8-bit images: OpenCV's generic fixed-point path
// (resize.cpp: 11-bit coefficients, horizontal pass in int, vertical pass ((b*(D>>4))>>16 ... +2)>>2).
__device__ __forceinline__ void cv_coeff(int d, double scale, int& s, float& f) {
    f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f = sub_rn(f, (float)s);
}
__device__ __forceinline__ int cv_short(float w) {
    int v = __float2int_rn(mul_rn(w, 2048.0f));
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
__global__ void k_resize_linear_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int B, int H, int W, int C,
                                   int Ho, int Wo, double scale_y, double scale_x) {
    const size_t n = (size_t)B * Ho * Wo * C;
    EW_LOOP(i, n) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int dx = (int)(r % Wo); r /= Wo;
        const int dy = (int)(r % Ho);
        const size_t b = r / Ho;
        int sx, sy; float fx, fy;
        cv_coeff(dx, scale_x, sx, fx);
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= W - 1) { fx = 0.f; sx = W - 1; }
        const int sx1 = min(sx + 1, W - 1);
        const int a0 = cv_short(sub_rn(1.0f, fx)), a1 = cv_short(fx);
        cv_coeff(dy, scale_y, sy, fy);
        const int b0 = cv_short(sub_rn(1.0f, fy)), b1 = cv_short(fy);
        const int y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
        const uint8_t* p0 = src + ((b * H + y0) * W) * C + c;
        const uint8_t* p1 = src + ((b * H + y1) * W) * C + c;
        const int D0 = (int)p0[(size_t)sx * C] * a0 + (int)p0[(size_t)sx1 * C] * a1;
        const int D1 = (int)p1[(size_t)sx * C] * a0 + (int)p1[(size_t)sx1 * C] * a1;
        int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
        dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}
int resize_linear_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return FGDM_ERR_ARG;
    const double sy = 1.0 / ((double)Ho / (double)H), sx = 1.0 / ((double)Wo / (double)W);
    FGDM_LAUNCH(k_resize_linear_u8, dim3(ew_grid((size_t)B * Ho * Wo * C)), dim3(EW_BLOCK), 0, s, src, dst, B, H, W, C,
                       Ho, Wo, sy, sx);
    return LAUNCH_OK();
}

// control = float(img) / 255.0, 'b h w c -> b c h w'   (controlnet/initialize_cn.py:78-80)
__global__ void k_u8_to_hint(const uint8_t* __restrict__ src, float* __restrict__ dst, int B, int HW, int C) {
    const size_t n = (size_t)B * C * HW;
    EW_LOOP(i, n) {
        const size_t p = i % HW;
        const size_t bc = i / HW;
        const size_t c = bc % C, b = bc / C;
        dst[i] = __fdiv_rn((float)src[(b * HW + p) * C + c], 255.0f);
    }
}
int u8_to_hint(const uint8_t* src, float* dst, int B, int HW, int C, hipStream_t s) {
    if (B <= 0 || HW <= 0 || C <= 0) return FGDM_ERR_ARG;
    FGDM_LAUNCH(k_u8_to_hint, dim3(ew_grid((size_t)B * C * HW)), dim3(EW_BLOCK), 0, s, src, dst, B, HW, C);
    return LAUNCH_OK();
}
