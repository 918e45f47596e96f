"""Stage boundary of the two-factor FG-DM chain  --  CPU oracle, TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Integer/byte restatement (numpy) of what happens between the condition stage and the ControlNet stage:
  * decoded image -> uint8: `clamp((x+1)/2, 0, 1)`, `255. * x`, `.astype(np.uint8)` (truncation)
        scripts/txt2img_fgdm_inference.py:245,249-252
  * `cv2.resize(img, (512, 512), interpolation=cv2.INTER_LINEAR)`          scripts/txt2img_fgdm_inference.py:258
  * `control = torch.from_numpy(img).float().cuda() / 255.0`, `b h w c -> b c h w`   controlnet/initialize_cn.py:78-80
  * final image -> uint8: `(x * 127.5 + 127.5).clip(0, 255).astype(np.uint8)`        controlnet/initialize_cn.py:101

PARITY UNPINNED for the resize: `cv2` (opencv-python, un-pinned in the reference's environment file) is a third-party
dependency that is absent from this image, the reference has no fixture for it, and it cannot be run here.  What is
restated is OpenCV 4.x `modules/imgproc/src/resize.cpp`, the generic fixed-point path for 8-bit INTER_LINEAR
(`resizeGeneric_` + `HResizeLinear<uchar,int,short>` + `VResizeLinear<uchar,int,short,FixedPtCast<..., 22>>`):
    fx = float((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; borders: fx = 0 and sx clamped (columns), rows clamped
    coefficients  a = saturate_cast<short>(w * 2048)  (round-half-even)
    horizontal    D[dx] = S[sx] * a0 + S[sx+1] * a1                                     (int)
    vertical      dst = ((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2   (uchar)
OpenCV builds that dispatch to IPP may differ from this path by one LSB.  The float<->uint8 conversions are plain IEEE
fp32 arithmetic and are exact restatements.
"""
import numpy as np


def image_to_uint8(x, mode=0):
    """x: float32 [B, C, H, W] -> uint8 [B, H, W, C].  mode 0: stage-A saving path, mode 1: initialize_cn.process."""
    x = np.asarray(x, dtype=np.float32)
    if mode == 0:
        y = np.clip((x + np.float32(1.0)) / np.float32(2.0), np.float32(0.0), np.float32(1.0))
        y = np.float32(255.0) * y
    else:
        y = np.clip(x * np.float32(127.5) + np.float32(127.5), np.float32(0.0), np.float32(255.0))
    return np.ascontiguousarray(y.transpose(0, 2, 3, 1)).astype(np.uint8)


def _coeffs(dst, src):
    """Per destination index: (source index, short coefficient pair) as cv2 computes them (before row clamping)."""
    scale = 1.0 / (float(dst) / float(src))                      # double, as in cv::resize
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _short(w):
    return np.clip(np.rint(w.astype(np.float32) * np.float32(2048.0)), -32768, 32767).astype(np.int64)


def resize_linear_u8(src, Ho, Wo):
    """src uint8 [B, H, W, C] -> uint8 [B, Ho, Wo, C], cv2.INTER_LINEAR generic fixed-point path."""
    src = np.asarray(src, dtype=np.uint8)
    B, H, W, C = src.shape
    sx, fx = _coeffs(Wo, W)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= W - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, W - 1, sx)
    a0, a1 = _short(np.float32(1.0) - fx), _short(fx)
    sx1 = np.minimum(sx + 1, W - 1)
    S = src.astype(np.int64)
    D = S[:, :, sx, :] * a0[None, None, :, None] + S[:, :, sx1, :] * a1[None, None, :, None]     # [B, H, Wo, C]
    sy, fy = _coeffs(Ho, H)
    b0, b1 = _short(np.float32(1.0) - fy), _short(fy)
    y0 = np.clip(sy, 0, H - 1)
    y1 = np.clip(sy + 1, 0, H - 1)
    D0 = D[:, y0] >> 4
    D1 = D[:, y1] >> 4
    out = (((b0[None, :, None, None] * D0) >> 16) + ((b1[None, :, None, None] * D1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def uint8_to_hint(img):
    """uint8 [B, H, W, C] -> float32 [B, C, H, W] in [0, 1] (initialize_cn.py:78-80)."""
    x = np.asarray(img, dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def hint_from_image(x, Ho=512, Wo=512):
    """decoded stage-A image (float32 NCHW, [-1, 1]) -> ControlNet hint (float32 NCHW, [0, 1])."""
    return uint8_to_hint(resize_linear_u8(image_to_uint8(x, 0), Ho, Wo))
