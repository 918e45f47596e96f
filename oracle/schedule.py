"""Noise schedules and DDIM tables.  TEST INFRASTRUCTURE.

Restates
  * make_beta_schedule("linear")      ldm/modules/diffusionmodules/util.py:21-27
  * DDPM.register_schedule            ldm/models/diffusion/ddpm.py:175-227
  * make_ddim_timesteps               util.py:46-60
  * make_ddim_sampling_parameters     util.py:63-74
  * DDIMSampler.make_schedule         ldm/models/diffusion/ddim.py:26-55
All tables are float64 numpy rounded to float32 exactly where the reference
rounds (``to_torch`` = torch.tensor(..., dtype=float32)).
"""
import numpy as np


def register_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.012, v_posterior=0.0):
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
    post_var = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    return dict(
        betas=f32(betas),
        alphas_cumprod=f32(ac),
        alphas_cumprod_prev=f32(ac_prev),
        sqrt_alphas_cumprod=f32(np.sqrt(ac)),
        sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - ac)),
        log_one_minus_alphas_cumprod=f32(np.log(1.0 - ac)),
        sqrt_recip_alphas_cumprod=f32(np.sqrt(1.0 / ac)),
        sqrt_recipm1_alphas_cumprod=f32(np.sqrt(1.0 / ac - 1)),
        posterior_variance=f32(post_var),
        posterior_log_variance_clipped=f32(np.log(np.maximum(post_var, 1e-20))),
        posterior_mean_coef1=f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        posterior_mean_coef2=f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    )


def ddim_timesteps(num_ddim, num_ddpm=1000, method='uniform'):
    if method == 'uniform':
        c = num_ddpm // num_ddim
        ts = np.asarray(list(range(0, num_ddpm, c)))
    elif method == 'quad':
        ts = (np.linspace(0, np.sqrt(num_ddpm * .8), num_ddim) ** 2).astype(int)
    else:
        raise NotImplementedError(method)
    return ts + 1


def ddim_tables(alphas_cumprod_f32, num_ddim, eta=0.0, method='uniform'):
    """alphas / alphas_prev / sigmas / sqrt(1-alphas) as float32 (the values the
    reference feeds to torch.full(..., dtype default float32), ddim.py:254-257)."""
    ts = ddim_timesteps(num_ddim, len(alphas_cumprod_f32), method)
    ac = np.asarray(alphas_cumprod_f32, dtype=np.float32)
    alphas = ac[ts]                                                  # float32 gather
    alphas_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())         # float64 holding f32 values
    # util.py:69: float64 ndarray (alphas_prev) op float32 tensor (alphas) -> numpy float64 math on
    # float32-valued inputs; the result is rounded to float32 only by torch.full at ddim.py:256.
    a32 = alphas.astype(np.float32)
    ap32 = alphas_prev.astype(np.float32)
    a64, ap64 = a32.astype(np.float64), ap32.astype(np.float64)
    sig = eta * np.sqrt((1 - ap64) / (1 - a64) * (1 - a64 / ap64))
    return dict(timesteps=ts, alphas=a32, alphas_prev=ap32, sigmas=sig.astype(np.float32),
                sqrt_one_minus_alphas=np.sqrt(np.float32(1.0) - a32).astype(np.float32))
