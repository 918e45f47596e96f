"""TEST INFRASTRUCTURE: torch-CPU stand-ins for the fused sampler kernels (csrc/elementwise.hip), used only by
the CPU host-logic tests to drive fgdm_amd.samplers / fgdm_amd.models without a GPU.  Same formulas as the
oracle (oracle/samplers.py); never imported by the product."""
import math

import torch


def ddim_step(x, e_cond, e_uncond, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None, want_pred_x0=True):
    e = e_cond if e_uncond is None else e_uncond + cfg_scale * (e_cond - e_uncond)
    f = lambda v: torch.full((x.shape[0], 1, 1, 1), float(v), dtype=torch.float32)
    pred = (x - f(sqrt_one_minus_at) * e) / f(a_t).sqrt()
    xp = f(a_prev).sqrt() * pred + (1. - f(a_prev) - f(sigma_t) ** 2).sqrt() * e
    if noise is not None:
        xp = xp + f(sigma_t) * noise
    return xp, (pred if want_pred_x0 else None)


def cfg_combine(e_cond, e_uncond, s):
    return e_uncond + s * (e_cond - e_uncond)


def plms_combine(e_t, old):
    if len(old) == 1:
        return (3 * e_t - old[-1]) / 2
    if len(old) == 2:
        return (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
    return (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24


def axpby(a, ca, b, cb):
    return ca * a + (cb * b if b is not None else 0)


def mask_blend(a, b, m):
    return a * m + (1. - m) * b


def ancestral_step(x, eps, cr, crm1, c1, c2, std, noise=None):
    x0 = float(cr) * x - float(crm1) * eps
    out = float(c1) * x0 + float(c2) * x
    return out + float(std) * noise if noise is not None else out
