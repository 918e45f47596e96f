"""Sampling loops.  TEST INFRASTRUCTURE.

Restates
  * DDIMSampler.sample / ddim_sampling / p_sample_ddim   ldm/models/diffusion/ddim.py:58-286
  * PLMSSampler.plms_sampling / p_sample_plms            ldm/models/diffusion/plms.py:115-236
  * ControlNet DDIMSampler (dict cond, sequential CFG)   controlnet/cldm/ddim_hacked.py:123-231
  * LatentDiffusion.p_sample_loop / p_sample / p_mean_variance / q_posterior /
    predict_start_from_noise / q_sample                  ldm/models/diffusion/ddpm.py:1382-1430, 1295-1323,
                                                         1260-1292, 290-297, 284-288, 342-345
``model_fn(x, t, cond)`` plays the role of ``model.apply_model``.
"""
import numpy as np
import torch

from . import schedule


def _full(b, v):
    return torch.full((b, 1, 1, 1), float(v), dtype=torch.float32)


def _cfg_eps(model_fn, x, t, cond, uc, scale, cfg_mode):
    if uc is None or scale == 1.0:
        return model_fn(x, t, cond)
    if cfg_mode == 'batched':              # ddim.py:222-243
        x_in = torch.cat([x] * 2)
        t_in = torch.cat([t] * 2)
        c_in = torch.cat([uc, cond])
        e_u, e_c = model_fn(x_in, t_in, c_in).chunk(2)
    else:                                  # ddim_hacked.py:190-192: two sequential calls
        e_c = model_fn(x, t, cond)
        e_u = model_fn(x, t, uc)
    return e_u + scale * (e_c - e_u)


def _x_prev(x, e_t, tab, index, temperature, noise_fn):
    # ddim.py:254-268
    b = x.shape[0]
    a_t, a_prev = _full(b, tab['alphas'][index]), _full(b, tab['alphas_prev'][index])
    sigma_t = _full(b, tab['sigmas'][index])
    s1m = _full(b, tab['sqrt_one_minus_alphas'][index])
    pred_x0 = (x - s1m * e_t) / a_t.sqrt()
    dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
    noise = sigma_t * noise_fn(x.shape) * temperature
    return a_prev.sqrt() * pred_x0 + dir_xt + noise, pred_x0


def _q_sample(sched, x0, t, noise):
    a = torch.from_numpy(sched['sqrt_alphas_cumprod'])[t].reshape(-1, 1, 1, 1)
    b = torch.from_numpy(sched['sqrt_one_minus_alphas_cumprod'])[t].reshape(-1, 1, 1, 1)
    return a * x0 + b * noise


def ddim_sample(model_fn, sched, S, shape, cond, x_T, eta=0.0, scale=1.0, uc=None,
                cfg_mode='batched', temperature=1.0, noise_fn=torch.randn,
                mask=None, x0=None, log_every_t=100, callback=None, img_callback=None):
    tab = schedule.ddim_tables(sched['alphas_cumprod'], S, eta)
    ts = tab['timesteps']
    b = shape[0]
    img = x_T
    inter = {'x_inter': [img], 'pred_x0': [img]}
    total = ts.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:
            img_orig = _q_sample(sched, x0, t, noise_fn(x0.shape))
            img = img_orig * mask + (1. - mask) * img
        e_t = _cfg_eps(model_fn, img, t, cond, uc, scale, cfg_mode)
        img, pred_x0 = _x_prev(img, e_t, tab, index, temperature, noise_fn)
        if callback:
            callback(i)
        if img_callback:
            img_callback(pred_x0, i)
        if index % log_every_t == 0 or index == total - 1:
            inter['x_inter'].append(img)
            inter['pred_x0'].append(pred_x0)
    return img, inter


def plms_sample(model_fn, sched, S, shape, cond, x_T, scale=1.0, uc=None,
                temperature=1.0, noise_fn=torch.randn, log_every_t=100):
    tab = schedule.ddim_tables(sched['alphas_cumprod'], S, 0.0)      # eta must be 0, plms.py:25-26
    ts = tab['timesteps']
    b = shape[0]
    img = x_T
    inter = {'x_inter': [img], 'pred_x0': [img]}
    total = ts.shape[0]
    time_range = np.flip(ts)
    old_eps = []
    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        t_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        e_t = _cfg_eps(model_fn, img, t, cond, uc, scale, 'batched')
        if len(old_eps) == 0:                                        # plms.py:219-223
            x_prev, _ = _x_prev(img, e_t, tab, index, temperature, noise_fn)
            e_next = _cfg_eps(model_fn, x_prev, t_next, cond, uc, scale, 'batched')
            e_prime = (e_t + e_next) / 2
        elif len(old_eps) == 1:
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img, pred_x0 = _x_prev(img, e_prime, tab, index, temperature, noise_fn)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if index % log_every_t == 0 or index == total - 1:
            inter['x_inter'].append(img)
            inter['pred_x0'].append(pred_x0)
    return img, inter


def p_sample_loop(model_fn, sched, cond, shape, x_T, timesteps=None, noise_fn=torch.randn,
                  temperature=1.0, clip_denoised=False, log_every_t=200):
    """Ancestral DDPM sampler (no CFG), ddpm.py:1382-1430."""
    T = len(sched['betas'])
    timesteps = T if timesteps is None else timesteps
    b = shape[0]
    img = x_T
    inter = [img]
    g = lambda name, t: torch.from_numpy(sched[name])[t].reshape(b, 1, 1, 1)
    for i in reversed(range(0, timesteps)):
        t = torch.full((b,), i, dtype=torch.long)
        eps = model_fn(img, t, cond)
        x_recon = g('sqrt_recip_alphas_cumprod', t) * img - g('sqrt_recipm1_alphas_cumprod', t) * eps
        if clip_denoised:
            x_recon = x_recon.clamp(-1., 1.)
        mean = g('posterior_mean_coef1', t) * x_recon + g('posterior_mean_coef2', t) * img
        logvar = g('posterior_log_variance_clipped', t)
        noise = noise_fn(img.shape) * temperature
        nonzero = (1 - (t == 0).float()).reshape(b, 1, 1, 1)
        img = mean + nonzero * (0.5 * logvar).exp() * noise
        if i % log_every_t == 0 or i == timesteps - 1:
            inter.append(img)
    return img, inter
