"""Host-side mirrors of the reference samplers; same constructor / sample() / p_sample_* signatures.

  DDIMSampler         <- ldm/models/diffusion/ddim.py:13-412
  PLMSSampler         <- ldm/models/diffusion/plms.py:11-236
  ControlDDIMSampler  <- controlnet/cldm/ddim_hacked.py:10-317 (dict conditionings)

The loop stays on the host (callbacks, intermediates, kwargs pass-through keep their reference semantics);
every per-step tensor update is one fused HIP kernel (csrc/elementwise.hip) and every model evaluation is
`model.apply_model` -- for fgdm_amd.models.* that is the HIP engine.  CFG is evaluated as ONE 2B batch
(cat([uncond, cond])) like ddim.py:222-243; the ControlNet sampler's two sequential calls
(ddim_hacked.py:190-192) are batched the same way when both branches use the same hint (identical math).
Not carried over (dead or unreachable in the reference): inference_loss / return_conds / x2.
"""
import numpy as np
import torch

from . import engine as _k
from . import schedule


def _randn(shape, device):
    return torch.randn(shape, device=device)


class _SamplerBase:
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != torch.device(self.model.device):
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    # ddim.py:26-55
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = schedule.ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        if verbose:
            print(f'Selected timesteps for ddim sampler: {self.ddim_timesteps}')
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        f32 = lambda v: torch.as_tensor(v).clone().detach().to(torch.float32).to(self.model.device)
        ac_h = ac.detach().float().cpu()
        self.register_buffer('betas', f32(self.model.betas))
        self.register_buffer('alphas_cumprod', f32(ac))
        self.register_buffer('alphas_cumprod_prev', f32(self.model.alphas_cumprod_prev))
        self.register_buffer('sqrt_alphas_cumprod', f32(torch.sqrt(ac_h)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', f32(torch.sqrt(1. - ac_h)))
        self.register_buffer('log_one_minus_alphas_cumprod', f32(torch.log(1. - ac_h)))
        self.register_buffer('sqrt_recip_alphas_cumprod', f32(torch.sqrt(1. / ac_h)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', f32(torch.sqrt(1. / ac_h - 1)))
        sig, a, ap, s1m = schedule.ddim_tables(ac_h.numpy(), self.ddim_timesteps, ddim_eta)
        if verbose:
            print(f'Selected alphas for ddim sampler: a_t: {a}; a_(t-1): {ap}')
            print(f'For the chosen value of eta, which is {ddim_eta}, this results in the following sigma_t '
                  f'schedule for ddim sampler {sig}')
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev, self.ddim_sqrt_one_minus_alphas = sig, a, ap, s1m
        acp_h = self.model.alphas_cumprod_prev.detach().float().cpu()
        self.register_buffer('ddim_sigmas_for_original_num_steps',
                             ddim_eta * torch.sqrt((1 - acp_h) / (1 - ac_h) * (1 - ac_h / acp_h)))

    def _tables(self, use_original_steps):
        """(alphas, alphas_prev, sqrt_one_minus_alphas, sigmas) as host float arrays (ddim.py:249-252)."""
        if not use_original_steps:
            return self.ddim_alphas, self.ddim_alphas_prev, self.ddim_sqrt_one_minus_alphas, self.ddim_sigmas
        h = lambda v: v.detach().float().cpu().numpy()
        return (h(self.model.alphas_cumprod), h(self.model.alphas_cumprod_prev),
                h(self.model.sqrt_one_minus_alphas_cumprod), h(self.ddim_sigmas_for_original_num_steps))

    @staticmethod
    def _batch_of(conditioning):
        if isinstance(conditioning, dict):
            v = conditioning[list(conditioning.keys())[0]]
            while isinstance(v, (list, tuple)):
                v = v[0]
            return v.shape[0]
        return conditioning.shape[0]

    # ---- model evaluation with classifier-free guidance; returns (e_cond, e_uncond_or_None)
    def _eval_pair(self, x, t, c, uc, scale, **kwargs):
        if uc is None or scale == 1.:
            return self.model.apply_model(x, t, c, **kwargs), None
        x_in, t_in = torch.cat([x] * 2), torch.cat([t] * 2)
        c_in = self._batched_cond(uc, c)
        if c_in is None:        # conditionings that cannot share one batch: two calls (ddim_hacked.py:190-191)
            return self.model.apply_model(x, t, c, **kwargs), self.model.apply_model(x, t, uc, **kwargs)
        # cfg_pairs: rows b and b + B of this batch differ only in the context (lets the engine share the network prefix).
        # The hint is understood by this repo's mirrors only: any other `model` (e.g. the reference's LatentDiffusion,
        # which forwards **kwargs into UNetModel.forward) is called exactly as the reference calls it.
        if getattr(self.model, 'engine', None) is not None:
            kwargs = dict(kwargs, cfg_pairs=True)
        e_u, e_c = self.model.apply_model(x_in, t_in, c_in, **kwargs).chunk(2)
        return e_c.contiguous(), e_u.contiguous()

    @staticmethod
    def _cat_cond(uc, c):
        return torch.cat([uc, c])

    @staticmethod
    def _leaves(c, out):
        if torch.is_tensor(c):
            out.append(c)
        elif isinstance(c, dict):
            for k in sorted(c):
                _SamplerBase._leaves(c[k], out)
        elif isinstance(c, (list, tuple)):
            for v in c:
                _SamplerBase._leaves(v, out)
        return out

    def _batched_cond(self, uc, c):
        """cat([uc, c]) is loop-invariant: build it once per (uc, c) pair and hand the SAME tensor object to every step,
        so the engine can keep the context's K/V projections (Engine.apply_model).  The memo holds the source tensors, so
        their ids cannot be recycled, and is invalidated by any in-place modification (_version)."""
        leaves = self._leaves(uc, []) + self._leaves(c, [])
        key = tuple((id(t), t._version) for t in leaves)
        memo = getattr(self, '_cin_memo', None)
        if memo is not None and memo[0] == key:
            return memo[2]
        c_in = self._cat_cond(uc, c)
        self._cin_memo = (key, leaves, c_in)
        return c_in

    def _x_prev(self, x, e_cond, e_uncond, scale, index, tabs, temperature, noise_dropout, repeat_noise,
                want_pred_x0=True):
        alphas, alphas_prev, s1m, sigmas = tabs
        sigma = float(sigmas[index])
        # the reference draws noise_like(x.shape) on EVERY step, also when sigma == 0 (ddim.py:265); drawing it
        # keeps the generator stream aligned with the reference (matters when q_sample also draws: mask blending)
        shape = (1, *x.shape[1:]) if repeat_noise else x.shape
        noise = _randn(shape, x.device)
        if sigma == 0.0:
            noise = None
        else:
            if repeat_noise:
                noise = noise.expand_as(x).contiguous()
            if temperature != 1.:
                noise = noise * temperature                       # rare options: plain tensor ops on the noise only
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        return _k.ddim_step(x.contiguous(), e_cond, e_uncond, scale, float(alphas[index]), float(alphas_prev[index]),
                            sigma, float(s1m[index]), noise, want_pred_x0)

    def _blend_mask(self, img, mask, x0, ts):
        img_orig = self.model.q_sample(x0, ts)
        m = mask.to(img.dtype).expand_as(img).contiguous()
        return _k.mask_blend(img_orig.contiguous(), img.contiguous(), m)

    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """q(x_t | x_0) at (ddim) index t (ddim.py:379-393)."""
        if use_original_steps:
            sa, s1m = self.sqrt_alphas_cumprod.cpu().numpy(), self.sqrt_one_minus_alphas_cumprod.cpu().numpy()
        else:
            sa, s1m = np.sqrt(self.ddim_alphas), self.ddim_sqrt_one_minus_alphas
        noise = torch.randn_like(x0) if noise is None else noise
        ti = np.asarray(t.detach().cpu())
        assert (ti == ti.flat[0]).all(), 'one encode index per call'
        return _k.axpby(x0.contiguous(), float(sa[ti.flat[0]]), noise.contiguous(), float(s1m[ti.flat[0]]))


class DDIMSampler(_SamplerBase):
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, inference_loss=False, **kwargs):
        if conditioning is not None:
            cbs = self._batch_of(conditioning)
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning,
                                  inference_loss=inference_loss, **kwargs)

    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, inference_loss=False,
                      **kwargs):
        device = self.model.betas.device
        b = shape[0]
        img = _randn(shape, device) if x_T is None else x_T.to(device, torch.float32)
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif not ddim_use_original_steps:
            n = self.ddim_timesteps.shape[0]
            timesteps = self.ddim_timesteps[:int(min(timesteps / n, 1) * n) - 1]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = list(reversed(range(0, timesteps))) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img = self._blend_mask(img, mask, x0, ts)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                              quantize_denoised=quantize_denoised, temperature=temperature,
                                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                                              corrector_kwargs=corrector_kwargs,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, i=i,
                                              inference_loss=inference_loss, **kwargs)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates

    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, i=0, inference_loss=False,
                      **kwargs):
        if inference_loss or kwargs.get('return_conds'):
            raise NotImplementedError('attention-alignment guidance / joint condition update need a model with a '
                                      'tuple output; no shipped model provides one (dead code in the reference)')
        if quantize_denoised:
            raise NotImplementedError('quantize_denoised needs a VQ first stage (not part of the latent-diffusion path)')
        uc, scale = unconditional_conditioning, unconditional_guidance_scale
        e_u = None
        if uc is None or scale == 1.:
            e_c = self.model.apply_model(x, t, c, **kwargs)
        elif 'composable_diffusion' in kwargs:                      # ddim.py:204-212
            kw = {k: v for k, v in kwargs.items() if k != 'composable_diffusion'}
            n = kwargs['composable_diffusion'] + 1
            out = self.model.apply_model(torch.cat([x] * n), torch.cat([t] * n), torch.cat([uc, c]), **kw)
            e_c = _k.axpby(out[:1].contiguous(), float(2 - n), None, 0.0)          # e_u + sum_k (e_k - e_u)
            for k in range(1, n):
                e_c = _k.axpby(e_c, 1.0, out[k:k + 1].contiguous(), 1.0)
        elif 'augmented_conditoning' in kwargs:                     # ddim.py:213-220 (sic)
            kw = {k: v for k, v in kwargs.items() if k not in ('augmented_conditoning', 'ac')}
            e_un, e_t, e_ac = self.model.apply_model(torch.cat([x] * 3), torch.cat([t] * 3),
                                                     torch.cat([uc, c, kwargs['ac']]), **kw).chunk(3)
            e_c = _k.cfg_combine(e_t.contiguous(), e_ac.contiguous(), scale)
            e_u = e_un.contiguous()
        else:
            e_c, e_u = self._eval_pair(x, t, c, uc, scale, **kwargs)
        if score_corrector is not None:
            assert self.model.parameterization == "eps"
            e_c = _k.cfg_combine(e_c, e_u, scale) if e_u is not None else e_c
            e_c, e_u = score_corrector.modify_score(self.model, e_c, x, t, c, **(corrector_kwargs or {})), None
        return self._x_prev(x, e_c, e_u, scale, index, self._tables(use_original_steps), temperature,
                            noise_dropout, repeat_noise)

    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        timesteps = timesteps[:t_start]
        total_steps = timesteps.shape[0]
        x_dec = x_latent
        for i, step in enumerate(np.flip(timesteps)):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index, use_original_steps=use_original_steps,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        return x_dec


class PLMSSampler(_SamplerBase):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')       # plms.py:25-26
        super().make_schedule(ddim_num_steps, ddim_discretize, ddim_eta, verbose)

    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        if conditioning is not None:
            cbs = self._batch_of(conditioning)
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for PLMS sampling is {size}')
        return self.plms_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    def plms_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None):
        device = self.model.betas.device
        b = shape[0]
        img = _randn(shape, device) if x_T is None else x_T.to(device, torch.float32)
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif not ddim_use_original_steps:
            n = self.ddim_timesteps.shape[0]
            timesteps = self.ddim_timesteps[:int(min(timesteps / n, 1) * n) - 1]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = list(reversed(range(0, timesteps))) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        old_eps = []
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img = self._blend_mask(img, mask, x0, ts)
            img, pred_x0, e_t = self.p_sample_plms(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                                   quantize_denoised=quantize_denoised, temperature=temperature,
                                                   noise_dropout=noise_dropout, score_corrector=score_corrector,
                                                   corrector_kwargs=corrector_kwargs,
                                                   unconditional_guidance_scale=unconditional_guidance_scale,
                                                   unconditional_conditioning=unconditional_conditioning,
                                                   old_eps=old_eps, t_next=ts_next)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates

    def p_sample_plms(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, old_eps=None, t_next=None):
        if quantize_denoised:
            raise NotImplementedError('quantize_denoised needs a VQ first stage')
        uc, scale = unconditional_conditioning, unconditional_guidance_scale
        tabs = self._tables(use_original_steps)

        def model_output(xx, tt):
            e_c, e_u = self._eval_pair(xx, tt, c, uc, scale)
            e = _k.cfg_combine(e_c, e_u, scale) if e_u is not None else e_c
            if score_corrector is not None:
                assert self.model.parameterization == "eps"
                e = score_corrector.modify_score(self.model, e, xx, tt, c, **(corrector_kwargs or {}))
            return e

        step = lambda e: self._x_prev(x, e, None, 1.0, index, tabs, temperature, noise_dropout, repeat_noise)
        e_t = model_output(x, t)
        if len(old_eps) == 0:            # pseudo improved Euler (plms.py:219-223): second evaluation at t_next
            x_prev, _ = step(e_t)
            e_prime = _k.axpby(e_t, 0.5, model_output(x_prev, t_next), 0.5)
        else:                            # Adams-Bashforth 2 / 3 / 4 (plms.py:224-232)
            e_prime = _k.plms_combine(e_t, old_eps)
        x_prev, pred_x0 = step(e_prime)
        return x_prev, pred_x0, e_t


class ControlDDIMSampler(DDIMSampler):
    """controlnet/cldm/ddim_hacked.py: conditionings are dicts {'c_concat': [hint] | None, 'c_crossattn': [ctx]}."""

    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None,
               ucg_schedule=None, **kwargs):
        if conditioning is not None:
            cbs = self._batch_of(conditioning)
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning,
                                  dynamic_threshold=dynamic_threshold, ucg_schedule=ucg_schedule)

    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None,
                      ucg_schedule=None):
        self._ucg = ucg_schedule
        self._dyn = dynamic_threshold
        if ucg_schedule is not None:
            n = self.ddpm_num_timesteps if ddim_use_original_steps else len(self.ddim_timesteps)
            assert timesteps is not None or len(ucg_schedule) == n
        return super().ddim_sampling(cond, shape, x_T=x_T, ddim_use_original_steps=ddim_use_original_steps,
                                     callback=callback, timesteps=timesteps, quantize_denoised=quantize_denoised,
                                     mask=mask, x0=x0, img_callback=img_callback, log_every_t=log_every_t,
                                     temperature=temperature, noise_dropout=noise_dropout,
                                     score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                     unconditional_guidance_scale=unconditional_guidance_scale,
                                     unconditional_conditioning=unconditional_conditioning)

    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None,
                      i=0, inference_loss=False, **kwargs):
        if dynamic_threshold is not None or getattr(self, '_dyn', None) is not None:
            raise NotImplementedError()                              # ddim_hacked.py:222-223
        if getattr(self, '_ucg', None) is not None:                 # ddim_hacked.py:159-161
            unconditional_guidance_scale = self._ucg[i]
        if self.model.parameterization == "v":
            raise NotImplementedError('v-parameterization is not used by cldm_v15 configs')
        return super().p_sample_ddim(x, c, t, index, repeat_noise=repeat_noise, use_original_steps=use_original_steps,
                                     quantize_denoised=quantize_denoised, temperature=temperature,
                                     noise_dropout=noise_dropout, score_corrector=score_corrector,
                                     corrector_kwargs=corrector_kwargs,
                                     unconditional_guidance_scale=unconditional_guidance_scale,
                                     unconditional_conditioning=unconditional_conditioning)

    def encode(self, x0, c, t_enc, use_original_steps=False, return_intermediates=None,
               unconditional_guidance_scale=1.0, unconditional_conditioning=None, callback=None):
        """DDIM inversion x_0 -> x_{t_enc} (ddim_hacked.py:234-279); requires make_schedule() first."""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        assert t_enc <= timesteps.shape[0]
        if use_original_steps:
            a_next = self.alphas_cumprod[:t_enc].detach().cpu().numpy().astype(np.float32)
            a_cur = self.alphas_cumprod_prev[:t_enc].detach().cpu().numpy().astype(np.float32)
        else:
            a_next = np.asarray(self.ddim_alphas[:t_enc], dtype=np.float32)
            a_cur = np.asarray(self.ddim_alphas_prev[:t_enc], dtype=np.float32)
        x_next = x0
        intermediates, inter_steps = [], []
        for i in range(t_enc):
            t = torch.full((x0.shape[0],), int(timesteps[i]), device=self.model.device, dtype=torch.long)
            if unconditional_guidance_scale == 1.:
                e = self.model.apply_model(x_next, t, c)
            else:
                assert unconditional_conditioning is not None
                e_c, e_u = self._eval_pair(x_next, t, c, unconditional_conditioning, unconditional_guidance_scale)
                e = _k.cfg_combine(e_c, e_u, unconditional_guidance_scale) if e_u is not None else e_c
            f32 = np.float32
            cx = np.sqrt(a_next[i] / a_cur[i])
            ce = np.sqrt(a_next[i]) * (np.sqrt(f32(1) / a_next[i] - f32(1)) - np.sqrt(f32(1) / a_cur[i] - f32(1)))
            x_next = _k.axpby(x_next.contiguous(), float(cx), e, float(ce))
            if return_intermediates and i % (t_enc // return_intermediates) == 0 and i < t_enc - 1:
                intermediates.append(x_next)
                inter_steps.append(i)
            elif return_intermediates and i >= t_enc - 2:
                intermediates.append(x_next)
                inter_steps.append(i)
            if callback:
                callback(i)
        out = {'x_encoded': x_next, 'intermediate_steps': inter_steps}
        if return_intermediates:
            out.update({'intermediates': intermediates})
        return x_next, out

    @staticmethod
    def _cat_cond(uc, c):
        """One 2B batch is possible when both branches carry the same hint tensors (guess_mode=False)."""
        if not (isinstance(uc, dict) and isinstance(c, dict)):
            return torch.cat([uc, c])
        hu, hc = uc.get('c_concat'), c.get('c_concat')
        if (hu is None) != (hc is None):
            return None
        if hu is not None:
            if len(hu) != len(hc) or any(a.data_ptr() != b.data_ptr() or a.shape != b.shape for a, b in zip(hu, hc)):
                return None
        if len(uc['c_crossattn']) != len(c['c_crossattn']):
            return None
        return {'c_concat': hc, 'c_crossattn': [torch.cat([a, b]) for a, b in zip(uc['c_crossattn'], c['c_crossattn'])]}


# ----------------------------------------------------------------------------------------------- DPM-Solver++
class _DiscreteVPSchedule:
    """NoiseScheduleVP('discrete', alphas_cumprod=...) of ldm/models/diffusion/dpm_solver/dpm_solver.py:98-160:
    log alpha_t is the piecewise-linear interpolation of 0.5 log(alphas_cumprod) over t_k = k/N, k = 1..N (T = 1).
    Host-side float32 scalars, like the reference's float32 tensors."""

    def __init__(self, alphas_cumprod):
        ac = np.asarray(alphas_cumprod, dtype=np.float32)
        self.log_alpha = (np.float32(0.5) * np.log(ac)).astype(np.float32)
        self.total_N = len(ac)
        self.T = 1.0
        self.t_array = np.linspace(0., 1., self.total_N + 1, dtype=np.float32)[1:]

    def marginal_log_mean_coeff(self, t):
        """interpolate_fn (dpm_solver.py:1132-1171) for one scalar: linear between keypoints, linear extrapolation."""
        t = np.float32(t)
        xp, yp = self.t_array, self.log_alpha
        K = len(xp)
        idx = int(np.searchsorted(xp, t, side='left'))        # number of keypoints strictly below t
        if idx == 0:
            i0 = 0
        elif idx >= K:
            i0 = K - 2
        else:
            i0 = idx - 1
        x0, x1, y0, y1 = xp[i0], xp[i0 + 1], yp[i0], yp[i0 + 1]
        return np.float32(y0 + (t - x0) * (y1 - y0) / (x1 - x0))

    def marginal_alpha(self, t):
        return np.float32(np.exp(self.marginal_log_mean_coeff(t)))

    def marginal_std(self, t):
        return np.float32(np.sqrt(np.float32(1.) - np.exp(np.float32(2.) * self.marginal_log_mean_coeff(t))))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return np.float32(lm - np.float32(0.5) * np.log(np.float32(1.) - np.exp(np.float32(2.) * lm)))


class DPMSolverSampler:
    """ldm/models/diffusion/dpm_solver/sampler.py:8-82: DPM-Solver++ (data prediction), multistep order 2,
    time_uniform steps from T = 1 to 1/N, lower_order_final, classifier-free guidance as one 2B batch.
    The model is evaluated at FRACTIONAL timesteps (t - 1/N) * 1000 (dpm_solver.py:278-287)."""

    def __init__(self, model, **kwargs):
        self.model = model
        self.alphas_cumprod = torch.as_tensor(model.alphas_cumprod).clone().detach().to(torch.float32).to(model.device)

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        if conditioning is not None:
            cbs = _SamplerBase._batch_of(conditioning)
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        C, H, W = shape
        size = (batch_size, C, H, W)
        device = self.model.betas.device
        x = _randn(size, device) if x_T is None else x_T.to(device, torch.float32)
        ns = _DiscreteVPSchedule(self.alphas_cumprod.detach().cpu().numpy())
        steps, order = S, 2
        assert steps >= order
        ts = np.linspace(ns.T, 1. / ns.total_N, steps + 1, dtype=np.float32)       # get_time_steps('time_uniform')
        scale, uc = unconditional_guidance_scale, unconditional_conditioning

        def data_prediction(xx, t):
            # model_wrapper (classifier-free, dpm_solver.py:321-343) + data_prediction_fn (:386-399)
            t_in = torch.full((xx.shape[0],), float((np.float32(t) - np.float32(1. / ns.total_N)) * np.float32(1000.)),
                              device=xx.device, dtype=torch.float32)
            if scale == 1. or uc is None:
                e = self.model.apply_model(xx, t_in, conditioning)
            else:
                kw = {'cfg_pairs': True} if getattr(self.model, 'engine', None) is not None else {}
                out = self.model.apply_model(torch.cat([xx] * 2), torch.cat([t_in] * 2), torch.cat([uc, conditioning]), **kw)
                e_u, e_c = out.chunk(2)
                e = _k.cfg_combine(e_c.contiguous(), e_u.contiguous(), scale)
            a, sg = ns.marginal_alpha(t), ns.marginal_std(t)
            return _k.axpby(xx.contiguous(), float(np.float32(1.) / a), e, float(-sg / a))     # (x - sigma eps) / alpha

        def first_update(xx, s, t, m_s):                         # dpm_solver.py:504-528 (predict_x0 branch)
            h = ns.marginal_lambda(t) - ns.marginal_lambda(s)
            c1 = ns.marginal_std(t) / ns.marginal_std(s)
            c2 = ns.marginal_alpha(t) * np.float32(np.expm1(-h))
            return _k.axpby(xx, float(c1), m_s, float(-c2))

        def second_update(xx, m_prev, t_prev, t):               # dpm_solver.py:755-789 ('dpm_solver' type, predict_x0)
            (m1, m0), (t1, t0) = m_prev, t_prev
            l1, l0, lt = ns.marginal_lambda(t1), ns.marginal_lambda(t0), ns.marginal_lambda(t)
            h0, h = l0 - l1, lt - l0
            r0 = h0 / h
            c1 = ns.marginal_std(t) / ns.marginal_std(t0)
            c2 = ns.marginal_alpha(t) * (np.float32(np.exp(-h)) - np.float32(1.))
            # x_t = c1 x - c2 m0 - 0.5 c2 (m0 - m1) / r0
            y = _k.axpby(xx, float(c1), m0, float(-c2 - np.float32(0.5) * c2 / r0))
            return _k.axpby(y, 1.0, m1, float(np.float32(0.5) * c2 / r0))

        m_prev = [data_prediction(x, ts[0])]
        t_prev = [ts[0]]
        x = first_update(x, t_prev[-1], ts[1], m_prev[-1])      # init_order = 1
        m_prev.append(data_prediction(x, ts[1]))
        t_prev.append(ts[1])
        for step in range(order, steps + 1):
            t = ts[step]
            step_order = min(order, steps + 1 - step) if steps < 15 else order      # lower_order_final
            if step_order == 1:
                x = first_update(x, t_prev[-1], t, m_prev[-1])
            else:
                x = second_update(x, m_prev, t_prev, t)
            t_prev = [t_prev[1], t]
            m_prev = [m_prev[1], m_prev[1]]
            if step < steps:
                m_prev[-1] = data_prediction(x, t)
            if callback:
                callback(step)
        return x.to(device), None
