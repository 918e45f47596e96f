"""Host-side mirrors of the reference's model objects for the sampling path.

  LatentDiffusion  <-  ldm/models/diffusion/ddpm.py  (register_schedule :175-227, apply_model :1035-1136,
                       q_sample :342-345, predict_start_from_noise :284-288, q_posterior :290-297,
                       p_mean_variance :1260-1292, p_sample :1295-1323, p_sample_loop :1382-1430, sample :1433-1448)
  ControlLDM       <-  controlnet/cldm/cldm.py:816-849 (apply_model with dict conds, control_scales)

They expose what the samplers and the inference scripts require of `model`: num_timesteps, betas,
alphas_cumprod(_prev), device, parameterization, apply_model, q_sample, control_scales, cuda()/to()/eval(),
ema_scope(), load_state_dict().  All network arithmetic runs in the HIP engine; the per-step latent updates run
in the sampler kernels.  decode_first_stage (ddpm.py:832-889) runs in the engine too when the
model is built with a `first_stage_config`, and get_learned_conditioning (FrozenCLIPEmbedder's transformer,
ldm/modules/encoders/modules.py:137-162) when it is built with a `cond_stage_config`; only the BPE
tokenizer (host code of the third-party `transformers` package) stays outside: `model.tokenizer`.
"""
import contextlib

import numpy as np
import torch

from . import _lib, config as _cfg, schedule
from . import engine as _k


def _randn(shape, device):
    return torch.randn(shape, device=device)


class _Buffers:
    """schedule tables as float32 torch tensors on the model's device (attribute names of register_schedule)."""

    def _register_schedule(self, device, **kw):
        tabs = schedule.ddpm_tables(**kw)
        for k, v in tabs.items():
            setattr(self, k, torch.from_numpy(v).to(device))
        self._host = tabs
        self.num_timesteps = int(tabs['betas'].shape[0])


class DiffusionWrapper:
    """ddpm.py:1829-1848: holds `.diffusion_model` (here: the HIP engine) and the conditioning key."""

    def __init__(self, engine, conditioning_key):
        self.diffusion_model = engine
        self.conditioning_key = conditioning_key


class LatentDiffusion(_Buffers):
    def __init__(self, unet_config=None, engine=None, use_adapter=True, n_controlnets=0, num_prompts=1, timesteps=1000,
                 beta_schedule='linear', linear_start=0.00085, linear_end=0.012, cosine_s=8e-3, given_betas=None,
                 v_posterior=0.0, parameterization='eps', conditioning_key='crossattn', scale_factor=0.18215,
                 channels=4, image_size=32, log_every_t=200, clip_denoised=False, device=0, first_stage_config=None,
                 cond_stage_config=None, **ignored):
        if parameterization != 'eps':
            raise NotImplementedError('only eps-parameterization is used by the shipped configs (models/config.yaml)')
        if conditioning_key != 'crossattn':
            raise NotImplementedError("only conditioning_key='crossattn' is on the hot path (models/config.yaml:15)")
        self.engine = engine if engine is not None else _k.Engine(device=device, **self.engine_args(
            unet_config=unet_config, use_adapter=use_adapter, n_controlnets=n_controlnets, num_prompts=num_prompts,
            first_stage_config=first_stage_config, cond_stage_config=cond_stage_config))
        self.device = self.engine.device
        self.model = DiffusionWrapper(self.engine, conditioning_key)
        self.parameterization = parameterization
        self.v_posterior = v_posterior
        self.scale_factor = scale_factor
        self.channels = channels
        self.image_size = image_size
        self.log_every_t = log_every_t
        self.clip_denoised = clip_denoised
        self.shorten_cond_schedule = False
        self.cond_stage_model = None        # optional callable(list[str]) -> [B,77,768] overriding the engine's text encoder
        self.tokenizer = None               # callable(list[str]) -> int64 ids [B,77]; default: transformers.CLIPTokenizer
        self.clip_version = 'openai/clip-vit-large-patch14'
        self.max_length = 77
        self.first_stage_decode = None      # optional callable(z / scale_factor) -> image overriding the engine's decoder
        self._register_schedule(self.device, kind=beta_schedule, timesteps=timesteps, linear_start=linear_start,
                                linear_end=linear_end, cosine_s=cosine_s, v_posterior=v_posterior,
                                given_betas=given_betas)
        self._finalized = False

    @classmethod
    def engine_args(cls, unet_config=None, use_adapter=True, n_controlnets=0, num_prompts=1, first_stage_config=None,
                    cond_stage_config=None, **ignored):
        """Constructor arguments of the reference model -> keyword arguments of fgdm_amd.engine.Engine / make_config (a pure
        function: works without a GPU).  unet_config as the scripts pass it: {target: ...UNetModel, params: {...}}
        (models/config.yaml:33-48; OmegaConf or dict) or the bare parameter dict; the target and its flags select the
        adapter variant the engine builds."""
        kind, cfg, uflags = _cfg.unet_params(unet_config)
        if kind == 'controlled' or uflags.get('no_prompting'):
            use_adapter = False                       # ControlledUnetModel / no_prompting: the plain SD UNet
        elif uflags.get('use_time_adapter'):
            use_adapter = 'time'
        if kind == 'adapt' or uflags.get('num_prompts', 1) > 1:
            num_prompts = max(num_prompts, int(uflags.get('num_prompts', num_prompts)))
        return dict(cfg=cfg, use_adapter=use_adapter, n_controlnets=n_controlnets, num_prompts=num_prompts,
                    vae=cls._ddconfig(first_stage_config), clip=cls._clipconfig(cond_stage_config))

    @staticmethod
    def _ddconfig(first_stage_config):
        """first_stage_config as in models/config.yaml:50-71 ({'target': ..., 'params': {'ddconfig': {...}}}), a bare
        ddconfig dict, True (SD-v1 decoder) or None (no decoder in the engine)."""
        if not first_stage_config:
            return None
        if first_stage_config is True:
            return True
        fc = dict(first_stage_config)
        if 'params' in fc:
            fc = dict(fc['params'])
        return dict(fc.get('ddconfig', fc))

    @staticmethod
    def _clipconfig(cond_stage_config):
        """cond_stage_config as in models/config.yaml:73-74 ({'target': '...FrozenCLIPEmbedder'}), True (SD-v1 text tower),
        a CLIPTextConfig-style dict, or None (no text encoder in the engine)."""
        if not cond_stage_config:
            return None
        if cond_stage_config is True:
            return True
        cc = dict(cond_stage_config)
        if 'target' in cc:
            if not str(cc['target']).endswith('FrozenCLIPEmbedder'):
                raise NotImplementedError(f"cond_stage_config target {cc['target']}: only FrozenCLIPEmbedder is shipped")
            return True
        return cc

    # ---- nn.Module-ish surface the scripts touch (scripts/txt2img_fgdm_inference.py:23-38,179-180,216-218)
    def cuda(self, *a, **k):
        return self

    def cpu(self, *a, **k):          # create_model(...).cpu() (controlnet/cldm/model.py:26): the engine lives on its GPU
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    @contextlib.contextmanager
    def ema_scope(self, context=None):
        yield None          # use_ema: False in every shipped config

    def load_state_dict(self, sd, strict=False):
        missing = self.engine.load_state_dict(sd, strict=strict)
        want = self.engine.param_shapes()
        unexpected = [k for k in sd if k not in want]
        if not missing:
            self.engine.finalize()
            self._finalized = True
        return missing, unexpected

    def _tokenize(self, text):
        """FrozenCLIPEmbedder.forward's tokenizer call (modules.py:153-155)."""
        if self.tokenizer is None:
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(self.clip_version)
            except Exception as e:      # no vocabulary files on this machine (no network)
                raise RuntimeError(f'CLIP tokenizer files for {self.clip_version} are not available ({e}); set '
                                   'model.tokenizer to a callable list[str] -> int64 ids [B, 77]') from e
            self.tokenizer = lambda t: tok(t, truncation=True, max_length=self.max_length, return_length=True,
                                           return_overflowing_tokens=False, padding='max_length',
                                           return_tensors='pt')['input_ids']
        return self.tokenizer(text)

    def get_learned_conditioning(self, c):
        """ddpm.py get_learned_conditioning -> cond_stage_model.encode(c): list of prompts (or ready token ids) ->
        [B, 77, 768]."""
        if self.cond_stage_model is not None:
            return self.cond_stage_model(c)
        if not getattr(self.engine, 'has_clip', False):
            raise NotImplementedError('this model was built without cond_stage_config; pass one (or set '
                                      'model.cond_stage_model to a callable returning [B,77,768])')
        ids = c if torch.is_tensor(c) or isinstance(c, np.ndarray) else self._tokenize(list(c))
        return self.engine.clip_encode(ids)

    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False, n=None):
        """ddpm.py:832-889 for an AutoencoderKL first stage: decode(z / scale_factor)."""
        if predict_cids:
            raise NotImplementedError('predict_cids needs a VQ first stage; every shipped config uses AutoencoderKL')
        if self.first_stage_decode is not None:
            return self.first_stage_decode(1. / self.scale_factor * z)
        if not self.engine.has_vae:
            raise NotImplementedError('this model was built without first_stage_config; pass one (or set '
                                      'model.first_stage_decode to a callable)')
        return self.engine.vae_decode(z, 1. / self.scale_factor)

    # ---- apply_model (ddpm.py:1035-1044,1130-1136 non-tiled branch; DiffusionWrapper crossattn mode)
    @staticmethod
    def _context(cond):
        if isinstance(cond, dict):
            cc = cond['c_crossattn']
        elif isinstance(cond, (list, tuple)):
            cc = list(cond)
        else:
            cc = [cond]
        return cc[0] if len(cc) == 1 else torch.cat(cc, 1)

    def apply_model(self, x_noisy, t, cond, return_ids=False, **kwargs):
        if return_ids:
            raise NotImplementedError('return_ids / return_conds needs a model with two outputs; no shipped model has one')
        flags = _lib.FLAG_NO_CONTROL
        if kwargs.get('use_original', False):
            flags |= _lib.FLAG_USE_ORIGINAL
        # AdaptUNetModel.forward(x, t, context, control=None, conds=None) (openaimodel.py:1263): `control` replaces the
        # adapter's prompt (UNetModel calls it `pcond`), `conds` feed the extra adapters
        if 'conds' in kwargs or getattr(self.engine, '_conds_key', None) is not None:
            self.engine.set_adapter_conds(kwargs.get('conds'))
        pcond = kwargs.get('pcond', kwargs.get('control'))
        # cfg_pairs (set by the samplers for cat([x] * 2) batches): the engine evaluates the network prefix on the first
        # half only, which is exact only if the two halves of a user-supplied prompt image are the same rows too
        if kwargs.get('cfg_pairs', False) and (pcond is None or self._halves_equal(pcond)):
            flags |= _lib.FLAG_CFG_PAIRS
        return self.engine.apply_model(x_noisy, t, self._context(cond), flags=flags, pcond=pcond)

    def _halves_equal(self, v):
        """True iff v[:B/2] == v[B/2:] (checked once per tensor object / version: the samplers pass the same object every
        step).  The memo keeps the tensor alive so its id cannot be recycled."""
        key = (id(v), v._version, tuple(v.shape))
        memo = getattr(self, '_halves_memo', None)
        if memo is None or memo[0] != key:
            n = v.shape[0]
            eq = n % 2 == 0 and bool(torch.equal(v[:n // 2], v[n // 2:]))
            self._halves_memo = memo = (key, v, eq)
        return memo[2]

    # ---- closed-form pieces
    def _at(self, name, t):
        return [float(v) for v in self._host[name][np.asarray(t.detach().cpu())]]

    def q_sample(self, x_start, t, noise=None):
        """sqrt(acp_t) x0 + sqrt(1-acp_t) noise; per-sample t handled by grouping equal timesteps."""
        noise = torch.randn_like(x_start) if noise is None else noise
        a, b = self._at('sqrt_alphas_cumprod', t), self._at('sqrt_one_minus_alphas_cumprod', t)
        if len(set(a)) == 1:
            return _k.axpby(x_start.contiguous(), a[0], noise.contiguous(), b[0])
        return torch.cat([_k.axpby(x_start[i:i + 1].contiguous(), a[i], noise[i:i + 1].contiguous(), b[i])
                          for i in range(x_start.shape[0])])

    def predict_start_from_noise(self, x_t, t, noise):
        a, b = self._at('sqrt_recip_alphas_cumprod', t), self._at('sqrt_recipm1_alphas_cumprod', t)
        assert len(set(a)) == 1, 'per-sample timesteps: call per sample'
        return _k.axpby(x_t.contiguous(), a[0], noise.contiguous(), -b[0])

    def p_sample(self, x, c, t, clip_denoised=False, repeat_noise=False, return_x0=False, temperature=1.,
                 noise_dropout=0., noise=None, **kwargs):
        """One ancestral step; all samples of a call share t (as in p_sample_loop, ddpm.py:1407)."""
        if clip_denoised or noise_dropout > 0.:
            raise NotImplementedError('clip_denoised / noise_dropout are not used by the latent configs')
        ti = int(t[0])
        assert bool((t == ti).all()), 'p_sample expects one timestep per call'
        eps = self.apply_model(x, t, c, **kwargs)
        h = self._host
        if noise is None:
            noise = _randn((1, *x.shape[1:]) if repeat_noise else x.shape, x.device)
            noise = noise.expand_as(x).contiguous()
        std = 0.0 if ti == 0 else float(np.exp(0.5 * h['posterior_log_variance_clipped'][ti])) * temperature
        out = _k.ancestral_step(x.contiguous(), eps, h['sqrt_recip_alphas_cumprod'][ti],
                                h['sqrt_recipm1_alphas_cumprod'][ti], h['posterior_mean_coef1'][ti],
                                h['posterior_mean_coef2'][ti], std, noise if ti != 0 else None)
        if return_x0:
            return out, self.predict_start_from_noise(x, t, eps)
        return out

    def p_sample_loop(self, cond, shape, return_intermediates=False, x_T=None, verbose=True, callback=None,
                      timesteps=None, quantize_denoised=False, mask=None, x0=None, img_callback=None, start_T=None,
                      log_every_t=None, **kwargs):
        if quantize_denoised:
            raise NotImplementedError('quantize_denoised needs a VQ first stage')
        log_every_t = log_every_t or self.log_every_t
        b = shape[0]
        img = _randn(shape, self.device) if x_T is None else x_T.to(self.device, torch.float32)
        inter = [img]
        timesteps = self.num_timesteps if timesteps is None else timesteps
        if start_T is not None:
            timesteps = min(timesteps, start_T)
        if mask is not None:
            assert x0 is not None and x0.shape[2:3] == mask.shape[2:3]
        for i in reversed(range(0, timesteps)):
            ts = torch.full((b,), i, device=self.device, dtype=torch.long)
            img = self.p_sample(img, cond, ts, clip_denoised=self.clip_denoised, **kwargs)
            if mask is not None:
                img_orig = self.q_sample(x0, ts)
                img = _k.mask_blend(img_orig.contiguous(), img.contiguous(),
                                    mask.to(img.dtype).expand_as(img).contiguous())
            if i % log_every_t == 0 or i == timesteps - 1:
                inter.append(img)
            if callback:
                callback(i)
            if img_callback:
                img_callback(img, i)
        return (img, inter) if return_intermediates else img

    def sample(self, cond, batch_size=16, return_intermediates=False, x_T=None, verbose=True, timesteps=None,
               quantize_denoised=False, mask=None, x0=None, shape=None, **kwargs):
        if shape is None:
            shape = (batch_size, self.channels, self.image_size, self.image_size)
        if cond is not None:
            if isinstance(cond, dict):
                cond = {k: (v[:batch_size] if not isinstance(v, list) else [u[:batch_size] for u in v])
                        for k, v in cond.items()}
            else:
                cond = [c[:batch_size] for c in cond] if isinstance(cond, list) else cond[:batch_size]
        return self.p_sample_loop(cond, shape, return_intermediates=return_intermediates, x_T=x_T, verbose=verbose,
                                  timesteps=timesteps, quantize_denoised=quantize_denoised, mask=mask, x0=x0, **kwargs)


class ControlLDM(LatentDiffusion):
    """controlnet/cldm/cldm.py:816-849.  Several control models (BASELINE configs 4/5) are an extension: their
    residual lists are summed; cond['c_concat'] then carries one hint per control model."""

    def __init__(self, unet_config=None, n_controlnets=1, only_mid_control=False, control_key='hint',
                 control_stage_config=None, **kw):
        kw.setdefault('use_adapter', False)       # ControlledUnetModel is the plain SD UNet (cldm.py:26)
        kw.setdefault('image_size', 64)
        self._check_control_stage(unet_config, control_stage_config)
        super().__init__(unet_config=unet_config, n_controlnets=n_controlnets, **kw)
        self.control_key = control_key
        self.only_mid_control = only_mid_control
        self.n_controlnets = n_controlnets
        self.control_scales = [1.0] * 13

    @staticmethod
    def _check_control_stage(unet_config, control_stage_config):
        """cldm_v15_canny.yaml:21-36: control_stage_config must describe the twin of the UNet's encoder"""
        if control_stage_config is None:
            return
        ckind, ccfg, cflags = _cfg.unet_params(control_stage_config)
        _, ucfg, _ = _cfg.unet_params(unet_config)
        ucfg = dict(_k.SD_V1 if ucfg is None else ucfg)
        diff = [k for k in ccfg if k != 'out_channels' and tuple(np.atleast_1d(ccfg[k])) != tuple(np.atleast_1d(ucfg[k]))]
        if ckind not in (None, 'controlnet') or diff or cflags.get('hint_channels', 3) != 3:
            raise NotImplementedError(f'control_stage_config must be a ControlNet matching unet_config (differs in {diff})')

    @classmethod
    def engine_args(cls, unet_config=None, n_controlnets=1, control_stage_config=None, **kw):
        kw.setdefault('use_adapter', False)
        cls._check_control_stage(unet_config, control_stage_config)
        return super().engine_args(unet_config=unet_config, n_controlnets=n_controlnets, **kw)

    def _scales(self):
        sc = list(self.control_scales)
        if len(sc) == 13:
            sc = sc * self.n_controlnets
        assert len(sc) == 13 * self.n_controlnets, 'control_scales: 13 per control model'
        return sc

    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        assert isinstance(cond, dict)
        ctx = self._context(cond)
        hints = cond.get('c_concat')
        flags = _lib.FLAG_ONLY_MID_CONTROL if self.only_mid_control else 0
        if kwargs.get('cfg_pairs', False):           # set by the samplers for cat([x] * 2) batches sharing one hint
            flags |= _lib.FLAG_CFG_PAIRS
        if hints is None:
            return self.engine.apply_model(x_noisy, t, ctx, flags=flags | _lib.FLAG_NO_CONTROL)
        if self.n_controlnets == 1:
            hints = [hints[0] if len(hints) == 1 else torch.cat(hints, 1)]     # torch.cat(cond['c_concat'], 1)
        assert len(hints) == self.n_controlnets
        for k, h in enumerate(hints):
            self.engine.set_hint(k, h)
        return self.engine.apply_model(x_noisy, t, ctx, control_scales=self._scales(), flags=flags)

    def get_unconditional_conditioning(self, N):
        return self.get_learned_conditioning([""] * N)
