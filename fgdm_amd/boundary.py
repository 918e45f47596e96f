"""Stage boundary of the two-factor FG-DM chain on the device.

Mirrors, without a host round trip, what scripts/txt2img_fgdm_inference.py:244-262 and
controlnet/initialize_cn.py:78-80,101 do with numpy / cv2 on the host: decoded condition image -> uint8 ->
bilinear resize -> ControlNet hint in [0, 1], and final image -> uint8.  All arithmetic runs in the HIP library
(fgdm_image_to_uint8 / fgdm_resize_linear_uint8 / fgdm_uint8_to_hint); there is no CPU fallback."""
import ctypes as C

import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed ({rc})')


def image_to_uint8(img, mode=0):
    """fp32 NCHW -> uint8 NHWC.  mode 0: `uint8(255 * clamp((x+1)/2, 0, 1))` (txt2img_fgdm_inference.py:245-252);
    mode 1: `uint8(clip(x*127.5+127.5, 0, 255))` (initialize_cn.py:101)."""
    img = img.to('cuda', torch.float32).contiguous()
    B, Cc, H, W = img.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.uint8, device=img.device)
    _check(_lib.load().fgdm_image_to_uint8(C.c_void_p(img.data_ptr()), B, Cc, H, W, int(mode),
                                            C.c_void_p(out.data_ptr()), _stream()), 'fgdm_image_to_uint8')
    return out


def resize_linear_uint8(u8, Ho, Wo):
    """cv2.resize(img, (Wo, Ho), interpolation=cv2.INTER_LINEAR) for a uint8 NHWC batch."""
    u8 = u8.to('cuda', torch.uint8).contiguous()
    B, H, W, Cc = u8.shape
    out = torch.empty(B, Ho, Wo, Cc, dtype=torch.uint8, device=u8.device)
    _check(_lib.load().fgdm_resize_linear_uint8(C.c_void_p(u8.data_ptr()), B, H, W, Cc, Ho, Wo,
                                                 C.c_void_p(out.data_ptr()), _stream()), 'fgdm_resize_linear_uint8')
    return out


def uint8_to_hint(u8):
    """uint8 NHWC -> fp32 NCHW in [0, 1]: `control = img.float() / 255` + rearrange (initialize_cn.py:78-80)."""
    u8 = u8.to('cuda', torch.uint8).contiguous()
    B, H, W, Cc = u8.shape
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=u8.device)
    _check(_lib.load().fgdm_uint8_to_hint(C.c_void_p(u8.data_ptr()), B, H, W, Cc, C.c_void_p(out.data_ptr()), _stream()),
           'fgdm_uint8_to_hint')
    return out


def hint_from_image(img, size=512):
    """Decoded stage-A image (fp32 NCHW in [-1, 1]) -> (hint fp32 NCHW [B,3,size,size], uint8 NHWC image as saved)."""
    u8 = image_to_uint8(img, 0)
    big = resize_linear_uint8(u8, size, size)
    return uint8_to_hint(big), u8
