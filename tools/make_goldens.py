#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own modules (CPU, fp32).

Runs only in the build container, where /root/reference is mounted.  It imports
the reference's Python modules (with tiny in-process stubs for third-party
packages that are not installed: omegaconf, torchvision, pytorch_lightning,
taming -- see SURVEY.md section 8c), loads the deterministic synthetic weights from
``fgdm_amd.synth`` into them via load_state_dict, runs them on seeded synthetic
inputs and writes small ``.npz`` fixtures under tests/golden/.  Only data
(inputs + expected outputs + key/shape lists) is written; no reference source
text is stored.

Usage:  python tools/make_goldens.py [--only NAME]
"""
import argparse
import json
import os
import sys
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('FGDM_REFERENCE', '/root/reference')
GOLD = os.path.join(ROOT, 'tests', 'golden')

from fgdm_amd import synth  # noqa: E402


def install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class ListConfig(list):
        pass
    stub('omegaconf', ListConfig=ListConfig, OmegaConf=object)
    stub('omegaconf.listconfig', ListConfig=ListConfig)
    tv = stub('torchvision')
    tv.utils = stub('torchvision.utils', save_image=lambda *a, **k: None, make_grid=lambda *a, **k: None)
    tv.transforms = stub('torchvision.transforms')

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device('cpu')

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass
    stub('pytorch_lightning', LightningModule=LightningModule)
    stub('pytorch_lightning.utilities')
    stub('pytorch_lightning.utilities.distributed', rank_zero_only=lambda f: f)
    stub('taming')
    stub('taming.modules')
    stub('taming.modules.vqvae')
    stub('taming.modules.vqvae.quantize', VectorQuantizer2=object)
    sys.path.insert(0, REF)


def load_synth(module, prefix='', seed=synth.DEFAULT_SEED):
    """Fill every parameter of a reference module from the synthetic generator."""
    sd = module.state_dict()
    new = {k: torch.from_numpy(synth.make_tensor(prefix + k, tuple(v.shape), seed)) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return {prefix + k: tuple(v.shape) for k, v in sd.items()}


sys.path.insert(0, os.path.join(ROOT, 'tests'))
import golden_inputs as gi  # noqa: E402


def ref_cfg(cfg, **extra):
    """oracle-style config dict -> reference UNetModel kwargs."""
    d = dict(image_size=32, use_spatial_transformer=True, use_checkpoint=False, legacy=False)
    d.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()})
    d.update(extra)
    return d


AC_SUFFIX = ''        # '_ac' while a generator runs under the CUDA-autocast emulation (oracle/autocast.py)


def save(name, **arrs):
    name = name + AC_SUFFIX
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)')


# --------------------------------------------------------------------------- G1/G2
def g_schedule():
    from ldm.modules.diffusionmodules.util import make_beta_schedule, make_ddim_timesteps, \
        make_ddim_sampling_parameters, timestep_embedding
    betas = make_beta_schedule('linear', 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)
    arrs = dict(betas=betas.astype(np.float32), alphas_cumprod=ac.astype(np.float32))
    ac_t = torch.tensor(ac, dtype=torch.float32)
    for S in (20, 50):
        for eta in (0.0, 1.0):
            ts = make_ddim_timesteps('uniform', S, 1000, verbose=False)
            sig, al, alp = make_ddim_sampling_parameters(ac_t, ts, eta, verbose=False)
            tag = f'S{S}_eta{int(eta)}'
            arrs[f'ts_{tag}'] = ts
            arrs[f'alphas_{tag}'] = np.asarray(al, dtype=np.float32)
            arrs[f'alphas_prev_{tag}'] = np.asarray(alp, dtype=np.float32)
            arrs[f'sigmas_{tag}'] = np.asarray(sig, dtype=np.float32)
            arrs[f'sqrt1m_{tag}'] = np.asarray(np.sqrt(1. - al), dtype=np.float32)
    t = torch.tensor([1, 21, 981, 500], dtype=torch.long)
    arrs['temb_t'] = t
    arrs['temb_320'] = timestep_embedding(t, 320)
    save('schedule', **arrs)


def g_ddpm_schedule():
    """DDPM.register_schedule buffers via the reference's LatentDiffusion class machinery."""
    import ldm.models.diffusion.ddpm as ddpm

    class Bare(ddpm.DDPM):
        def __init__(self):
            nn.Module.__init__(self)
            self.parameterization = 'eps'
            self.v_posterior = 0.0
    m = Bare()
    m.register_schedule(beta_schedule='linear', timesteps=1000, linear_start=0.00085, linear_end=0.012)
    names = ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
             'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod',
             'sqrt_recipm1_alphas_cumprod', 'posterior_variance', 'posterior_log_variance_clipped',
             'posterior_mean_coef1', 'posterior_mean_coef2']
    save('ddpm_schedule', **{n: getattr(m, n) for n in names})


# --------------------------------------------------------------------------- keys
def g_param_keys():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from controlnet.cldm.cldm import ControlNet, ControlledUnetModel
    out = {}
    with torch.device('meta'):
        sd = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
        out['unet_fgdm'] = sd(UNetModel(**ref_cfg(gi.SD_CFG)))
        out['unet_plain'] = sd(UNetModel(**ref_cfg(gi.SD_CFG, no_prompting=True)))
        out['unet_time_adapter'] = sd(UNetModel(**ref_cfg(gi.SD_CFG, use_time_adapter=True)))
        from ldm.modules.diffusionmodules.openaimodel import AdaptUNetModel
        out['adapt_unet_3'] = sd(AdaptUNetModel(**ref_cfg(gi.SD_CFG, num_prompts=3)))
        out['controlnet'] = sd(ControlNet(**ref_cfg(gi.SD_CFG, hint_channels=3)))
        out['controlled_unet'] = sd(ControlledUnetModel(**ref_cfg(gi.SD_CFG)))
        out['unet_small'] = sd(UNetModel(**ref_cfg(gi.SMALL_CFG, no_prompting=True)))
        out['controlnet_small'] = sd(ControlNet(**ref_cfg(gi.SMALL_CFG, hint_channels=3)))
        out['unet_narrow'] = sd(UNetModel(**ref_cfg(gi.NARROW_CFG, no_prompting=True)))
        from ldm.models.autoencoder import AutoencoderKL
        vae = sd(AutoencoderKL(ddconfig=vae_ddconfig(), lossconfig={'target': 'torch.nn.Identity'}, embed_dim=4))
        out['vae_decoder'] = {'first_stage_model.' + k: v for k, v in vae.items()
                              if k.startswith(('decoder.', 'post_quant_conv.'))}
    with open(os.path.join(GOLD, 'param_keys.json'), 'w') as f:
        json.dump(out, f)
    print('wrote param_keys.json', {k: len(v) for k, v in out.items()})


def vae_ddconfig():
    from oracle import vae as ovae
    c = ovae.SD_VAE
    return dict(double_z=True, z_channels=c['z_channels'], resolution=c['resolution'], in_channels=c['in_channels'],
                out_ch=c['out_ch'], ch=c['ch'], ch_mult=list(c['ch_mult']), num_res_blocks=c['num_res_blocks'],
                attn_resolutions=[], dropout=0.0)


def g_vae():
    """First-stage decoder: the reference's AutoencoderKL.decode on z / scale_factor."""
    from ldm.models.autoencoder import AutoencoderKL
    from oracle import vae as ovae
    m = AutoencoderKL(ddconfig=vae_ddconfig(), lossconfig={'target': 'torch.nn.Identity'}, embed_dim=4).eval()
    load_synth(m, 'first_stage_model.')
    out = {}
    with torch.no_grad():
        for key in ('vae/z8', 'vae/z16'):
            z = gi.get(key)
            out['img_' + key.split('/')[1]] = m.decode(1.0 / ovae.SCALE_FACTOR * z)
        # the attention block alone (single head, d = 512) on the conv_in features of z8
        z = m.post_quant_conv(1.0 / ovae.SCALE_FACTOR * gi.get('vae/z8'))
        h = m.decoder.mid.block_1(m.decoder.conv_in(z), None)
        out['mid_attn_z8'] = m.decoder.mid.attn_1(h)
    save('vae', **out)


def g_clip():
    """CLIP text encoder.  Third-party: FrozenCLIPEmbedder = transformers.CLIPTextModel
    (ldm/modules/encoders/modules.py:137-162); run here from the installed transformers with synthetic weights."""
    # transformers probes optional packages with importlib.util.find_spec: hide the spec-less stubs while importing it
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split('.')[0] in ('torchvision', 'omegaconf', 'taming',
                                                                                      'pytorch_lightning')}
    try:
        import transformers
        from transformers import CLIPTextConfig, CLIPTextModel
        CLIPTextModel(CLIPTextConfig(num_hidden_layers=1, hidden_size=64, intermediate_size=64, num_attention_heads=1,
                                     vocab_size=8))          # force the lazy sub-imports now
    finally:
        sys.modules.update(hidden)
    from oracle import clip as oclip
    c = oclip.SD_CLIP
    m = CLIPTextModel(CLIPTextConfig(hidden_act='quick_gelu', attention_dropout=0.0, **c)).eval()
    sd = m.state_dict()
    inner = 'text_model.' if any(k.startswith('text_model.') for k in sd) else ''
    new = {}
    for k, v in sd.items():
        name = oclip.PREFIX + k[len(inner):]
        if 'position_ids' in k:
            new[k] = v
        else:
            new[k] = torch.from_numpy(synth.make_tensor(name, tuple(v.shape)))
    m.load_state_dict(new, strict=True)
    ids = gi.clip_ids()
    with torch.no_grad():
        z = m(input_ids=ids).last_hidden_state
    keys = {oclip.PREFIX + k[len(inner):]: list(v.shape) for k, v in sd.items() if 'position_ids' not in k}
    with open(os.path.join(GOLD, 'clip_keys.json'), 'w') as f:
        json.dump({'transformers_version': transformers.__version__, 'keys': keys}, f)
    save('clip', z=z, ids=ids)


# --------------------------------------------------------------------------- G3 per-op
def g_ops():
    from ldm.modules.diffusionmodules.openaimodel import ResBlock, Downsample, Upsample
    from ldm.modules.diffusionmodules.util import normalization
    from ldm.modules.attention import CrossAttention, FeedForward, BasicTransformerBlock, SpatialTransformer, Normalize
    from ldm.modules.encoders.adapter import ResnetBlock, Adapter
    arrs = {}
    with torch.no_grad():
        emb = gi.get('ops/emb')
        for tag, (cin, cout) in dict(res_320_320=(320, 320), res_320_640=(320, 640),
                                     res_2560_1280=(2560, 1280), res_960_320=(960, 320)).items():
            m = ResBlock(cin, 1280, 0.0, out_channels=cout, use_checkpoint=False).eval()
            load_synth(m, tag + '.')
            arrs[tag + '_y'] = m(gi.get(f'ops/{tag}_x'), emb)
        m = Downsample(320, True, out_channels=320).eval()
        load_synth(m, 'down.')
        arrs['down_y'] = m(gi.get('ops/down_x'))
        m = Upsample(640, True, out_channels=640).eval()
        load_synth(m, 'up.')
        arrs['up_y'] = m(gi.get('ops/up_x'))
        for tag, mk in dict(gn5=normalization, gn6=Normalize).items():
            m = mk(320).eval()
            load_synth(m, tag + '.')
            arrs[tag + '_y'] = m(gi.get(f'ops/{tag}_x'))
        # attention: self (T=64, C=320, d=40), cross (ctx 77x768), self d=160
        ctx = gi.get('ops/ctx')
        x = gi.get('ops/attn_x')
        m = CrossAttention(320, heads=8, dim_head=40).eval()
        load_synth(m, 'attn_self.')
        arrs['attn_self_y'] = m(x)[0]
        m = CrossAttention(320, context_dim=768, heads=8, dim_head=40).eval()
        load_synth(m, 'attn_cross.')
        arrs['attn_cross_y'] = m(x, context=ctx)[0]
        m = CrossAttention(1280, heads=8, dim_head=160).eval()
        load_synth(m, 'attn_self160.')
        arrs['attn_self160_y'] = m(gi.get('ops/attn160_x'))[0]
        m = FeedForward(320, glu=True).eval()
        load_synth(m, 'ff.')
        x = gi.get('ops/ff_x')
        arrs['ff_y'] = m(x)
        m = BasicTransformerBlock(320, 8, 40, context_dim=768, checkpoint=False).eval()
        load_synth(m, 'tblock.')
        arrs['tblock_y'] = m(x, context=ctx)
        m = SpatialTransformer(640, 8, 80, depth=1, context_dim=768).eval()
        for b in m.transformer_blocks:
            b.checkpoint = False
        load_synth(m, 'st.')
        arrs['st_y'] = m(gi.get('ops/st_x'), context=ctx)
        for tag, (ch, dh) in dict(st320=(320, 40), st1280=(1280, 160)).items():     # head dims 40 and 160
            m = SpatialTransformer(ch, 8, dh, depth=1, context_dim=768).eval()
            for b in m.transformer_blocks:
                b.checkpoint = False
            load_synth(m, tag + '.')
            arrs[tag + '_y'] = m(gi.get(f'ops/{tag}_x'), context=ctx)
        # adapter pieces
        x = gi.get('ops/arb_x')
        m = ResnetBlock(320, 640, down=True, ksize=1, sk=True, use_conv=False).eval()
        load_synth(m, 'arb_down.')
        arrs['arb_down_y'] = m(x)
        m = ResnetBlock(320, 320, down=False, ksize=1, sk=True, use_conv=False).eval()
        load_synth(m, 'arb_same.')
        arrs['arb_same_y'] = m(x)
        m = Adapter(cin=4, channels=[320, 640, 1280, 1280], nums_rb=2, ksize=1, sk=True, use_conv=False).eval()
        load_synth(m, 'adapter.')
        for i, f in enumerate(m(gi.get('ops/adapter_x'))):
            arrs[f'adapter_f{i}'] = f
    save('ops', **arrs)


# --------------------------------------------------------------------------- G4 full nets
def g_unet_full():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    arrs = {}
    with torch.no_grad():
        m = UNetModel(**ref_cfg(gi.SD_CFG)).eval()
        load_synth(m, 'model.diffusion_model.')
        ctx = gi.get('unet/ctx')
        t = torch.tensor([981, 1], dtype=torch.long)
        arrs['t'] = t
        for hw in (8, 16):
            x = gi.get(f'unet/x{hw}')
            arrs[f'eps_orig{hw}'] = m(x, t, context=ctx, use_original=True)
            arrs[f'eps_fgdm{hw}'] = m(x, t, context=ctx)
        mt = UNetModel(**ref_cfg(gi.SD_CFG, use_time_adapter=True)).eval()
        load_synth(mt, 'model.diffusion_model.')
        for hw in (8, 16):
            arrs[f'eps_tadapt{hw}'] = mt(gi.get(f'unet/x{hw}'), t, context=ctx)
    save('unet_full', **arrs)


def g_adapt_unet():
    """AdaptUNetModel (openaimodel.py:887-1320) with num_prompts = 3: two extra adapters over `conds`, `control` prompt."""
    from ldm.modules.diffusionmodules.openaimodel import AdaptUNetModel
    arrs = {}
    with torch.no_grad():
        m = AdaptUNetModel(**ref_cfg(gi.SD_CFG, num_prompts=3)).eval()
        load_synth(m, 'model.diffusion_model.')
        ctx = gi.get('unet/ctx')
        t = torch.tensor([981, 1], dtype=torch.long)
        x = gi.get('unet/x16')
        conds = [gi.get('adapt/cond0'), gi.get('adapt/cond1')]
        arrs['eps_conds'] = m(x, t, context=ctx, conds=conds)
        arrs['eps_conds_control'] = m(x, t, context=ctx, conds=conds, control=gi.get('adapt/control'))
        arrs['eps_plain'] = m(x, t, context=ctx)
    save('adapt_unet', **arrs)


def g_controlnet_full():
    from controlnet.cldm.cldm import ControlNet, ControlledUnetModel
    arrs = {}
    with torch.no_grad():
        cn = ControlNet(**ref_cfg(gi.SD_CFG, hint_channels=3)).eval()
        load_synth(cn, 'control_model.')
        cu = ControlledUnetModel(**ref_cfg(gi.SD_CFG)).eval()
        load_synth(cu, 'model.diffusion_model.')
        ctx = gi.get('cn/ctx')
        t = torch.tensor([981, 21], dtype=torch.long)
        x = gi.get('cn/x')
        hint = gi.hint(2, 64, 45)
        arrs['t'] = t
        ctrl = cn(x=x, hint=hint, timesteps=t, context=ctx)
        for i, c in enumerate(ctrl):
            arrs[f'ctrl{i}'] = c
        arrs['eps_ctrl'] = cu(x=x, timesteps=t, context=ctx,
                              control=[c * s for c, s in zip(ctrl, gi.CTRL_SCALES)], only_mid_control=False)
        arrs['eps_noctrl'] = cu(x=x, timesteps=t, context=ctx, control=None)
        arrs['guided'] = cn.input_hint_block(gi.hint(1, 64, 46), None, None)
    save('controlnet_full', **arrs)


# --------------------------------------------------------------------------- G5 reduced nets at 64x64
def g_small_nets():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from controlnet.cldm.cldm import ControlNet, ControlledUnetModel
    arrs = {}
    with torch.no_grad():
        ctx = gi.get('small/ctx')
        t = torch.tensor([801, 801], dtype=torch.long)
        x = gi.get('small/x')
        arrs['t'] = t
        m = UNetModel(**ref_cfg(gi.SMALL_CFG, no_prompting=True)).eval()
        load_synth(m, 'small.')
        arrs['eps_small'] = m(x, t, context=ctx)
        m = UNetModel(**ref_cfg(gi.NARROW_CFG, no_prompting=True)).eval()
        load_synth(m, 'narrow.')
        arrs['eps_narrow32'] = m(x[:, :, :32, :32].contiguous(), t, context=ctx)
        cn = ControlNet(**ref_cfg(gi.SMALL_CFG, hint_channels=3)).eval()
        load_synth(cn, 'small_cn.')
        cu = ControlledUnetModel(**ref_cfg(gi.SMALL_CFG)).eval()
        load_synth(cu, 'small.')
        ctrl = cn(x=x, hint=gi.hint(2, 512, 47), timesteps=t, context=ctx)
        arrs['eps_small_ctrl'] = cu(x=x, timesteps=t, context=ctx, control=list(ctrl))
    save('small_nets', **arrs)


# --------------------------------------------------------------------------- G6 sampler trajectories
class FakeModel:
    """Duck-typed `model` for the reference samplers: analytic eps, real schedule buffers."""

    def __init__(self, fn):
        import ldm.models.diffusion.ddpm as ddpm

        class Bare(ddpm.DDPM):
            def __init__(s):
                nn.Module.__init__(s)
                s.parameterization = 'eps'
                s.v_posterior = 0.0
        self._m = Bare()
        self._m.register_schedule(beta_schedule='linear', timesteps=1000, linear_start=0.00085, linear_end=0.012)
        for n in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_one_minus_alphas_cumprod',
                  'sqrt_alphas_cumprod'):
            setattr(self, n, getattr(self._m, n))
        self.num_timesteps = 1000
        self.device = torch.device('cpu')
        self.parameterization = 'eps'
        self.fn = fn
        self.calls = 0

    def apply_model(self, x, t, c, **kw):
        self.calls += 1
        return self.fn(x, t, c)

    def q_sample(self, x0, t, noise=None):
        return self._m.q_sample(x0, t, noise)


def analytic_eps(x, t, c):
    """Cheap deterministic stand-in for the UNet: depends on x, t and the conditioning."""
    if isinstance(c, dict):
        cc = c['c_crossattn'][0]
        hint = c['c_concat'][0] if c.get('c_concat') is not None else None
    else:
        cc, hint = c, None
    s = cc.mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
    tt = (t.float() / 1000.0).reshape(-1, 1, 1, 1)
    e = 0.7 * x + 0.2 * torch.sin(3.0 * x + s) + 0.1 * tt * torch.roll(x, 1, dims=3) + 0.05 * s
    if hint is not None:
        e = e + 0.1 * torch.nn.functional.avg_pool2d(hint, 8).mean(dim=1, keepdim=True)
    return e


def _cpu_sampler(cls):
    class S(cls):
        def register_buffer(self, name, attr):   # reference hard-codes "cuda" (ddim.py:20-24)
            setattr(self, name, attr)
    return S


def g_samplers():
    import contextlib
    import io
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    from controlnet.cldm.ddim_hacked import DDIMSampler as CNSampler
    import ldm.models.diffusion.ddpm as ddpm
    arrs = {}
    shape = (4, 8, 8)
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    sink = io.StringIO()
    with torch.no_grad(), contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        for S, scale, eta in ((50, 7.5, 0.0), (20, 9.0, 0.0), (20, 7.5, 1.0), (10, 1.0, 0.0)):
            fm = FakeModel(analytic_eps)
            smp = _cpu_sampler(DDIMSampler)(fm)
            torch.manual_seed(123)
            out, inter = smp.sample(S, 2, shape, conditioning=c, x_T=x_T, eta=eta, verbose=False,
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                    log_every_t=5)
            tag = f'ddim_S{S}_s{scale}_eta{eta}'
            arrs[tag] = out
            arrs[tag + '_xinter'] = torch.stack(inter['x_inter'])
            arrs[tag + '_predx0'] = torch.stack(inter['pred_x0'])
            arrs[tag + '_calls'] = np.asarray([fm.calls])
        fm = FakeModel(analytic_eps)
        smp = _cpu_sampler(PLMSSampler)(fm)
        out, inter = smp.sample(50, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, log_every_t=5)
        arrs['plms_S50'] = out
        arrs['plms_S50_xinter'] = torch.stack(inter['x_inter'])
        arrs['plms_S50_calls'] = np.asarray([fm.calls])
        # ControlNet sampler: dict conds, sequential CFG
        hint = gi.hint(2, 64, 48)
        fm = FakeModel(analytic_eps)
        smp = _cpu_sampler(CNSampler)(fm)
        cond = {'c_concat': [hint], 'c_crossattn': [c]}
        ucond = {'c_concat': [hint], 'c_crossattn': [uc]}
        out, inter = smp.sample(20, 2, shape, cond, verbose=False, eta=0.0, x_T=x_T,
                                unconditional_guidance_scale=9.0, unconditional_conditioning=ucond)
        arrs['cn_ddim_S20'] = out
        arrs['cn_ddim_S20_calls'] = np.asarray([fm.calls])
        # inpainting-style mask blend with eta=0 (q_sample noise still drawn: ddim.py:151-154)
        fm = FakeModel(analytic_eps)
        smp = _cpu_sampler(DDIMSampler)(fm)
        mask, x0 = gi.get('samp/mask'), gi.get('samp/x0')
        torch.manual_seed(321)
        out, _ = smp.sample(10, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False, mask=mask, x0=x0,
                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
        arrs['ddim_mask_S10'] = out
        # ancestral p_sample_loop through the reference LatentDiffusion methods (12-step excerpt)
        class LD(ddpm.LatentDiffusion):
            def __init__(s):
                nn.Module.__init__(s)
                s.parameterization = 'eps'
                s.v_posterior = 0.0
                s.clip_denoised = False
                s.log_every_t = 4
                s.num_timesteps_cond = 1
                s.register_schedule(beta_schedule='linear', timesteps=1000, linear_start=0.00085, linear_end=0.012)

            def apply_model(s, x, t, c, **kw):
                return analytic_eps(x, t, c)
        ld = LD()
        torch.manual_seed(99)
        img, inter = ld.p_sample_loop(c, (2,) + shape, return_intermediates=True, x_T=x_T, verbose=False,
                                      timesteps=12)
        arrs['ancestral_T12'] = img
        arrs['ancestral_T12_inter'] = torch.stack(inter)
    save('samplers', **arrs)


def g_samplers2():
    """DPM-Solver++ (scripts/txt2img.py --dpm_solver) and DDIM inversion (ddim_hacked.encode) with the analytic model."""
    import contextlib
    import io
    from ldm.models.diffusion.dpm_solver import DPMSolverSampler
    from controlnet.cldm.ddim_hacked import DDIMSampler as CNSampler
    arrs = {}
    shape = (4, 8, 8)
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    sink = io.StringIO()
    with torch.no_grad(), contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        for S, scale in ((20, 7.5), (10, 7.5), (12, 1.0)):
            fm = FakeModel(analytic_eps)
            smp = _cpu_sampler(DPMSolverSampler)(fm)
            out, _ = smp.sample(S, 2, shape, conditioning=c, x_T=x_T, verbose=False,
                                unconditional_guidance_scale=scale, unconditional_conditioning=uc)
            arrs[f'dpm_S{S}_s{scale}'] = out
            arrs[f'dpm_S{S}_s{scale}_calls'] = np.asarray([fm.calls])
        fm = FakeModel(analytic_eps)
        smp = _cpu_sampler(CNSampler)(fm)
        smp.make_schedule(20, ddim_eta=0.0, verbose=False)
        hint = gi.hint(2, 64, 48)
        cond = {'c_concat': [hint], 'c_crossattn': [c]}
        ucond = {'c_concat': [hint], 'c_crossattn': [uc]}
        x0 = gi.get('samp/x0')
        # the reference's CFG branch of encode() concatenates the conditionings with torch.cat, which only works for
        # tensor conditionings: use tensors for the guided case and dict conds for the unguided one
        enc, info = smp.encode(x0, c, 12, unconditional_guidance_scale=5.0, unconditional_conditioning=uc,
                               return_intermediates=3)
        arrs['encode_cfg'] = enc
        arrs['encode_cfg_inter'] = torch.stack(info['intermediates'])
        arrs['encode_cfg_steps'] = np.asarray(info['intermediate_steps'])
        enc, _ = smp.encode(x0, cond, 15)
        arrs['encode_plain'] = enc
    save('samplers2', **arrs)


class Corrector:
    """a score_corrector in the sense of ddim.py:245-247: any object with modify_score(model, e_t, x, t, c, **kwargs)"""

    def modify_score(self, model, e_t, x, t, c, gain=1.0):
        return e_t + gain * 0.1 * torch.tanh(x) * (t.float() / 1000.0).reshape(-1, 1, 1, 1)


def g_samplers3():
    """Sampler branches that exist in the reference but that no script exercises, with the analytic model:
    composable_diffusion (ddim.py:204-212), augmented_conditoning + ac (:213-220), score_corrector (:245-247),
    ddim_sampling(timesteps=) truncation (:131-134), use_original_steps / decode (:395-412, :249-252),
    stochastic_encode (:379-393), ControlNet sampler ucg_schedule (ddim_hacked.py:159-161)."""
    import contextlib
    import io
    from ldm.models.diffusion.ddim import DDIMSampler
    from controlnet.cldm.ddim_hacked import DDIMSampler as CNSampler
    arrs = {}
    shape = (4, 8, 8)
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    sink = io.StringIO()
    with torch.no_grad(), contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        # composable diffusion: ONE latent, two prompts and one unconditional context
        fm = FakeModel(analytic_eps)
        out, _ = _cpu_sampler(DDIMSampler)(fm).sample(10, 1, shape, conditioning=c, x_T=x_T[:1], eta=0.0, verbose=False,
                                                     unconditional_guidance_scale=7.5, unconditional_conditioning=uc[:1],
                                                     composable_diffusion=2)
        arrs['compose'] = out
        arrs['compose_calls'] = np.asarray([fm.calls])
        # augmented conditioning: third context `ac`, two nested guidance combinations
        fm = FakeModel(analytic_eps)
        ac = gi.get('samp/ac')
        out, _ = _cpu_sampler(DDIMSampler)(fm).sample(10, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                                     unconditional_guidance_scale=3.0, unconditional_conditioning=uc,
                                                     augmented_conditoning=True, ac=ac)
        arrs['augmented'] = out
        arrs['augmented_calls'] = np.asarray([fm.calls])
        fm = FakeModel(analytic_eps)
        out, _ = _cpu_sampler(DDIMSampler)(fm).sample(10, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                                     unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                                     score_corrector=Corrector(), corrector_kwargs={'gain': 0.5})
        arrs['corrector'] = out
        # ddim_sampling(timesteps=10) on a 20-step schedule: the first 9 ddim timesteps
        fm = FakeModel(analytic_eps)
        smp = _cpu_sampler(DDIMSampler)(fm)
        smp.make_schedule(20, ddim_eta=0.0, verbose=False)
        out, inter = smp.ddim_sampling(c, (2,) + shape, x_T=x_T, timesteps=10, unconditional_guidance_scale=7.5,
                                       unconditional_conditioning=uc, log_every_t=1)
        arrs['truncated'] = out
        arrs['truncated_n'] = np.asarray([fm.calls])
        # decode from ddim index 12, and from step 15 of the ORIGINAL 1000-step schedule (the reference reads the sigmas for
        # that from self.model, ddim.py:252, so the model has to carry them)
        x_lat = gi.get('samp/x0')
        fm = FakeModel(analytic_eps)
        smp = _cpu_sampler(DDIMSampler)(fm)
        smp.make_schedule(20, ddim_eta=0.0, verbose=False)
        arrs['decode_ddim12'] = smp.decode(x_lat, c, 12, unconditional_guidance_scale=5.0, unconditional_conditioning=uc)
        fm.ddim_sigmas_for_original_num_steps = smp.ddim_sigmas_for_original_num_steps
        arrs['decode_orig15'] = smp.decode(x_lat, c, 15, unconditional_guidance_scale=5.0, unconditional_conditioning=uc,
                                           use_original_steps=True)
        noise = gi.get('samp/noise')
        arrs['stoch_ddim7'] = smp.stochastic_encode(x_lat, torch.tensor([7, 7]), noise=noise)
        arrs['stoch_orig300'] = smp.stochastic_encode(x_lat, torch.tensor([300, 300]), use_original_steps=True, noise=noise)
        # ControlNet sampler with a guidance-scale schedule
        hint = gi.hint(2, 64, 48)
        fm = FakeModel(analytic_eps)
        ucg = [1.0 + 0.8 * i for i in range(10)]
        out, _ = _cpu_sampler(CNSampler)(fm).sample(10, 2, shape, {'c_concat': [hint], 'c_crossattn': [c]}, verbose=False, eta=0.0,
                                                   x_T=x_T, unconditional_guidance_scale=9.0, ucg_schedule=ucg,
                                                   unconditional_conditioning={'c_concat': [hint], 'c_crossattn': [uc]})
        arrs['cn_ucg'] = out
        arrs['cn_ucg_schedule'] = np.asarray(ucg)
    save('samplers3', **arrs)


def g_sampler_unet():
    """End-to-end compounding: reference DDIMSampler driving the reference reduced UNet (SMALL_CFG) at 16x16."""
    import contextlib
    import io
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.models.diffusion.ddim import DDIMSampler
    arrs = {}
    m = UNetModel(**ref_cfg(gi.SMALL_CFG, no_prompting=True)).eval()
    load_synth(m, 'small.')
    fm = FakeModel(lambda x, t, c: m(x, t, context=c))
    smp = _cpu_sampler(DDIMSampler)(fm)
    shape = (4, 16, 16)
    x_T, c, uc = gi.get('sunet/x_T'), gi.get('sunet/c'), gi.get('sunet/uc')
    sink = io.StringIO()
    with torch.no_grad(), contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        out, inter = smp.sample(10, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, log_every_t=1)
    arrs.update(out=out, xinter=torch.stack(inter['x_inter']))
    save('sampler_unet', **arrs)


def _reference_control_ldm(cn, cu):
    """The reference's ControlLDM (controlnet/cldm/cldm.py:816-849) around already-built control / diffusion models: the heavy
    LatentDiffusion constructor (first stage, text encoder: not on the path, need the network) is bypassed, the methods are the
    reference's own -- so `apply_model` below IS ControlLDM.apply_model, not a re-wiring of it."""
    from controlnet.cldm.cldm import ControlLDM

    class Wrapper(nn.Module):
        def __init__(self, m):
            super().__init__()
            self.diffusion_model = m

    class LDM(ControlLDM):
        def __init__(s):
            nn.Module.__init__(s)
            s.model = Wrapper(cu)
            s.control_model = cn
            s.control_key = 'hint'
            s.only_mid_control = False
            s.control_scales = [1.0] * 13
    return LDM()


def g_full_size_check():
    """Certifies tests/golden/full_size.npz against ControlLDM.apply_model itself without the 25 minutes of the
    50-step trajectory: the CFG pair at t = 981 through the reference class must reproduce the stored fixture bit for bit."""
    import json
    from controlnet.cldm.cldm import ControlNet, ControlledUnetModel
    cn = ControlNet(**ref_cfg(gi.SD_CFG, hint_channels=3)).eval()
    load_synth(cn, 'control_model.')
    cu = ControlledUnetModel(**ref_cfg(gi.SD_CFG)).eval()
    load_synth(cu, 'model.diffusion_model.')
    ldm = _reference_control_ldm(cn, cu)
    x = torch.from_numpy(synth.latents(1, 64, 64, seed=42))
    c = torch.from_numpy(synth.context(1, seed=43))
    uc = torch.from_numpy(synth.context(1, seed=44))
    hint = torch.from_numpy(synth.hint(1, 512, seed=45))
    name = 'full_size' + AC_SUFFIX
    want = np.load(os.path.join(GOLD, name + '.npz'))['eps_pair_t981']
    with torch.no_grad():
        got = ldm.apply_model(torch.cat([x, x]), torch.full((2,), 981, dtype=torch.long),
                              {'c_concat': [torch.cat([hint, hint])], 'c_crossattn': [torch.cat([uc, c])]})
    got = got.float().numpy()
    diff = float(np.abs(got - want.astype(np.float32)).max())
    rec = {'fixture': name + '.npz:eps_pair_t981', 'through': 'controlnet.cldm.cldm.ControlLDM.apply_model (cldm.py:836-849)',
           'max_abs_diff': diff, 'bit_identical': bool(np.array_equal(got, want.astype(np.float32)))}
    print('   ', rec)
    assert diff == 0.0, rec
    with open(os.path.join(GOLD, name + '_apply_model_check.json'), 'w') as f:
        json.dump(rec, f, indent=1)


def g_full_size():
    """G7: the metric's own workload at full size -- SD-v1.5-width ControlledUnetModel + ControlNet, latent 64x64,
    hint 512x512 -- through the reference's modules wired as ControlLDM.apply_model does (cldm.py:836-849):
      * one classifier-free-guidance pair (same x and hint, uncond / cond context) at t = 981 and t = 21;
      * a complete 50-step eta = 0 DDIM sampling, CFG 9.0, with the reference's ControlNet sampler (ddim_hacked.py:55-231):
        the latent after steps 1, 2, 5, 10, 20, 30, 40, 50 and (sum, L2 norm) of every step's latent and pred_x0."""
    import contextlib
    import io
    import time
    from controlnet.cldm.cldm import ControlNet, ControlledUnetModel
    from controlnet.cldm.ddim_hacked import DDIMSampler as CNSampler
    cn = ControlNet(**ref_cfg(gi.SD_CFG, hint_channels=3)).eval()
    load_synth(cn, 'control_model.')
    cu = ControlledUnetModel(**ref_cfg(gi.SD_CFG)).eval()
    load_synth(cu, 'model.diffusion_model.')
    apply_model = _reference_control_ldm(cn, cu).apply_model      # the reference's own method body (cldm.py:836-849)
    x = torch.from_numpy(synth.latents(1, 64, 64, seed=42))
    c = torch.from_numpy(synth.context(1, seed=43))
    uc = torch.from_numpy(synth.context(1, seed=44))
    hint = torch.from_numpy(synth.hint(1, 512, seed=45))
    arrs = {}
    t0 = time.time()
    with torch.no_grad():
        for tv in (981, 21):
            t = torch.full((2,), tv, dtype=torch.long)
            arrs[f'eps_pair_t{tv}'] = apply_model(torch.cat([x, x]), t, {'c_concat': [torch.cat([hint, hint])],
                                                                       'c_crossattn': [torch.cat([uc, c])]})
        print(f'   CFG pairs done, {time.time() - t0:.0f} s', flush=True)
        fm = FakeModel(lambda xx, tt, cc: apply_model(xx, tt, cc))
        smp = _cpu_sampler(CNSampler)(fm)
        keep = {1, 2, 5, 10, 20, 30, 40, 50}
        sums = []

        def img_cb(pred_x0, i):
            pass
        sink = io.StringIO()
        traj = {}
        # the sampler reports pred_x0 through img_callback and x through the intermediates (log_every_t = 1)
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            out, inter = smp.sample(50, 1, (4, 64, 64), {'c_concat': [hint], 'c_crossattn': [c]}, verbose=False, eta=0.0,
                                    x_T=x, unconditional_guidance_scale=9.0, log_every_t=1,
                                    unconditional_conditioning={'c_concat': [hint], 'c_crossattn': [uc]})
        xs = inter['x_inter'][1:]           # [0] is x_T
        ps = inter['pred_x0'][1:]
        assert len(xs) == 50, len(xs)
        for i, (xi, pi) in enumerate(zip(xs, ps)):
            xi, pi = xi.float(), pi.float()
            sums.append([float(xi.double().sum()), float(xi.double().norm()), float(pi.double().sum()), float(pi.double().norm())])
            if i + 1 in keep:
                arrs[f'x_step{i + 1}'] = xi
        arrs['traj_sums'] = np.asarray(sums)
        arrs['out'] = out.float()
        arrs['calls'] = np.asarray([fm.calls])
    print(f'   50-step trajectory done, {time.time() - t0:.0f} s', flush=True)
    save('full_size', **arrs)


ALL = dict(schedule=g_schedule, ddpm_schedule=g_ddpm_schedule, param_keys=g_param_keys, ops=g_ops,
           unet_full=g_unet_full, controlnet_full=g_controlnet_full, small_nets=g_small_nets,
           samplers=g_samplers, samplers2=g_samplers2, samplers3=g_samplers3, sampler_unet=g_sampler_unet, vae=g_vae, clip=g_clip, adapt_unet=g_adapt_unet, full_size=g_full_size,
           full_size_check=g_full_size_check)


# generators that are ALSO run with the reference's modules under the emulated torch.autocast("cuda") policy
# (scripts/txt2img_fgdm_inference.py:212-217 wraps the whole sampling loop in it) -> tests/golden/<name>_ac.npz
AC = ('ops', 'unet_full', 'controlnet_full', 'small_nets', 'sampler_unet', 'adapt_unet', 'vae', 'clip', 'full_size', 'full_size_check')


def main():
    global AC_SUFFIX
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--ac', action='store_true', help='only the autocast-policy variants (<name>_ac.npz)')
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit(f'reference checkout not found at {REF}; goldens can only be regenerated in the build container')
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    torch.set_num_threads(os.cpu_count() or 1)
    from oracle import autocast
    for name, fn in ALL.items():
        if a.only and a.only != name:
            continue
        if not a.ac:
            print(f'== {name}')
            fn()
        if name in AC:
            print(f'== {name} under the autocast policy')
            AC_SUFFIX = '_ac'
            try:
                with autocast.emulate() as mode:
                    fn()
                print('   ', mode.stats)
            finally:
                AC_SUFFIX = ''


if __name__ == '__main__':
    main()
