"""Stage boundary (decoded image -> uint8 -> cv2-style bilinear resize -> hint).
CPU part: sanity of the oracle (oracle/boundary.py; the resize is "parity unpinned": cv2 is absent here).
GPU part: the HIP kernels against the oracle, bit-exact (byte / integer work)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import boundary as ob


def _img(B=2, H=32, W=48, seed=3):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, 3, H, W), dtype=np.float32) * 0.8
    x[0, 0, 0, :4] = [-1.0, 1.0, -1.0000001, 0.99999994]      # edge values of the clamp
    return x


def test_uint8_conversions():
    x = _img()
    u0 = ob.image_to_uint8(x, 0)
    want = (255.0 * np.clip((x.astype(np.float64) + 1) / 2, 0, 1)).transpose(0, 2, 3, 1)
    assert u0.dtype == np.uint8 and u0.shape == (2, 32, 48, 3)
    assert np.max(np.abs(u0.astype(np.float64) - np.floor(want))) <= 1          # fp32 vs fp64 rounding at most one step
    assert u0[0, 0, 0, 0] == 0 and u0[0, 0, 1, 0] == 255
    u1 = ob.image_to_uint8(x, 1)
    assert np.max(np.abs(u1.astype(np.int64) - u0.astype(np.int64))) <= 1
    h = ob.uint8_to_hint(u0)
    assert h.dtype == np.float32 and h.shape == (2, 3, 32, 48) and h.min() >= 0 and h.max() <= 1
    assert np.array_equal(np.round(h * 255).astype(np.uint8).transpose(0, 2, 3, 1), u0)


@pytest.mark.parametrize('H,W,Ho,Wo', [(32, 48, 64, 96), (256, 256, 512, 512), (20, 20, 50, 30), (64, 64, 32, 32)])
def test_resize_matches_float_bilinear_within_one(H, W, Ho, Wo):
    """The fixed-point restatement stays within 1 LSB of exact half-pixel-centre bilinear interpolation
    (torch F.interpolate, align_corners=False), and reproduces constant images exactly."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    got = ob.resize_linear_u8(src, Ho, Wo)
    t = torch.from_numpy(src).permute(0, 3, 1, 2).double()
    ref = F.interpolate(t, size=(Ho, Wo), mode='bilinear', align_corners=False).permute(0, 2, 3, 1).numpy()
    assert got.shape == (2, Ho, Wo, 3)
    assert np.max(np.abs(got.astype(np.float64) - ref)) <= 1.0
    const = np.full((1, H, W, 3), 201, dtype=np.uint8)
    assert np.all(ob.resize_linear_u8(const, Ho, Wo) == 201)


def test_resize_2x_coefficients():
    """2x upsampling uses exactly the weights (0.25, 0.75) away from the borders and replicates the border pixel."""
    src = np.zeros((1, 4, 4, 1), dtype=np.uint8)
    src[0, :, :, 0] = np.arange(4)[None, :] * 64
    got = ob.resize_linear_u8(src, 8, 8)[0, 3, :, 0]
    assert list(got) == [0, 16, 48, 80, 112, 144, 176, 192]


@pytest.mark.gpu
def test_gpu_boundary_kernels_bit_exact():
    from fgdm_amd import boundary as fb
    x = _img(B=3, H=40, W=56, seed=9)
    xt = torch.from_numpy(x).cuda()
    for mode in (0, 1):
        assert np.array_equal(fb.image_to_uint8(xt, mode).cpu().numpy(), ob.image_to_uint8(x, mode)), mode
    u8 = ob.image_to_uint8(x, 0)
    for Ho, Wo in ((80, 112), (100, 75), (20, 28)):
        got = fb.resize_linear_uint8(torch.from_numpy(u8).cuda(), Ho, Wo).cpu().numpy()
        assert np.array_equal(got, ob.resize_linear_u8(u8, Ho, Wo)), (Ho, Wo)
    assert np.array_equal(fb.uint8_to_hint(torch.from_numpy(u8).cuda()).cpu().numpy(), ob.uint8_to_hint(u8))


@pytest.mark.gpu
def test_gpu_hint_from_image_full_size():
    """256x256 condition image -> 512x512 hint, the size the inference script uses."""
    from fgdm_amd import boundary as fb
    x = _img(B=2, H=256, W=256, seed=11)
    hint, u8 = fb.hint_from_image(torch.from_numpy(x).cuda(), 512)
    assert np.array_equal(u8.cpu().numpy(), ob.image_to_uint8(x, 0))
    assert np.array_equal(hint.cpu().numpy(), ob.hint_from_image(x, 512, 512))
