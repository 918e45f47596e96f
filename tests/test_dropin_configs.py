"""Script-level drop-in (CPU part): the configs the inference scripts hand to `instantiate_from_config` / `create_model`
(scripts/txt2img_fgdm_inference.py:23-38, controlnet/initialize_cn.py:25-43, controlnet/seg2image_inference.py:36-40), typed in
here as the literals of models/config.yaml:1-74 and controlnet/models/cldm_v15_canny.yaml:1-85, go through the mirrors' own
argument handling and must produce exactly the reference's state-dict key tables (tests/golden/param_keys.json, made from
the reference's modules).  No GPU needed: `engine_args` / `make_config` / `param_shapes` are host code."""
import json
import os

import pytest

from common import GOLD
from fgdm_amd import config, engine, models

UNET_PARAMS = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                   num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
                   context_dim=768, use_checkpoint=True, legacy=False)
FIRST_STAGE = {'target': 'ldm.models.autoencoder.AutoencoderKL',
               'params': {'embed_dim': 4, 'monitor': 'val/rec_loss',
                          'ddconfig': dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                                           ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0),
                          'lossconfig': {'target': 'torch.nn.Identity'}}}
COND_STAGE = {'target': 'ldm.modules.encoders.modules.FrozenCLIPEmbedder'}
# models/config.yaml (the FG-DM condition factor)
FGDM_YAML = {'model': {'base_learning_rate': 1.0e-05, 'target': 'ldm.models.diffusion.ddpm.LatentDiffusion', 'params': dict(
    linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000, first_stage_key='image',
    cond_stage_key='caption', image_size=32, channels=4, cond_stage_trainable=False, conditioning_key='crossattn',
    monitor='val/loss_simple_ema', scale_factor=0.18215, use_ema=False, freeze_backbone=True, apply_distill_loss=True,
    distill_every_n_step=10,
    scheduler_config={'target': 'ldm.lr_scheduler.LambdaLinearScheduler', 'params': {'warm_up_steps': [10000]}},
    unet_config={'target': 'ldm.modules.diffusionmodules.openaimodel.UNetModel', 'params': UNET_PARAMS},
    first_stage_config=FIRST_STAGE, cond_stage_config=COND_STAGE)}}
# controlnet/models/cldm_v15_canny.yaml (the image factor)
CLDM_YAML = {'model': {'target': 'controlnet.cldm.cldm.ControlLDM', 'params': dict(
    linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000, first_stage_key='image',
    cond_stage_key='caption', control_key='hint', image_size=64, channels=4, cond_stage_trainable=False,
    conditioning_key='crossattn', monitor='val/loss_simple_ema', scale_factor=0.18215, use_ema=False, only_mid_control=False,
    control_stage_config={'target': 'controlnet.cldm.cldm.ControlNet',
                          'params': dict({k: v for k, v in UNET_PARAMS.items() if k != 'out_channels'}, hint_channels=3)},
    unet_config={'target': 'controlnet.cldm.cldm.ControlledUnetModel', 'params': UNET_PARAMS},
    first_stage_config=dict(FIRST_STAGE, target='controlnet.ldm.models.autoencoder.AutoencoderKL'),
    cond_stage_config={'target': 'controlnet.ldm.modules.encoders.modules.FrozenCLIPEmbedder'})}}


def _ref():
    return json.load(open(os.path.join(GOLD, 'param_keys.json')))


def _shapes(cls, yaml_model):
    target, params = config.split(yaml_model)
    args = cls.engine_args(**params)
    return engine.param_shapes(engine.make_config(**args))


def _check(mine, theirs, prefix):
    got = {k[len(prefix):]: list(v) for k, v in mine.items() if k.startswith(prefix)}
    assert list(got.keys()) == list(theirs.keys())
    assert got == theirs


def test_fgdm_yaml_gives_the_reference_key_table():
    ref = _ref()
    sh = _shapes(models.LatentDiffusion, FGDM_YAML['model'])
    _check(sh, ref['unet_fgdm'], 'model.diffusion_model.')                       # UNetModel with the FG-DM adapter
    assert {k: list(v) for k, v in sh.items() if k.startswith('first_stage_model.')} == ref['vae_decoder']
    clip = json.load(open(os.path.join(GOLD, 'clip_keys.json')))['keys']
    assert {k: list(v) for k, v in sh.items() if k.startswith('cond_stage_model.')} == clip
    # variants selected by UNetModel flags (openaimodel.py:551-556)
    y = json.loads(json.dumps(FGDM_YAML['model']))
    y['params']['unet_config']['params']['no_prompting'] = True
    _check(_shapes(models.LatentDiffusion, y), ref['unet_plain'], 'model.diffusion_model.')
    y['params']['unet_config']['params'].pop('no_prompting')
    y['params']['unet_config']['params']['use_time_adapter'] = True
    _check(_shapes(models.LatentDiffusion, y), ref['unet_time_adapter'], 'model.diffusion_model.')
    y['params']['unet_config'] = {'target': 'ldm.modules.diffusionmodules.openaimodel.AdaptUNetModel',
                                  'params': dict(UNET_PARAMS, num_prompts=3)}
    _check(_shapes(models.LatentDiffusion, y), ref['adapt_unet_3'], 'model.diffusion_model.')


def test_cldm_yaml_gives_the_reference_key_tables():
    ref = _ref()
    sh = _shapes(models.ControlLDM, CLDM_YAML['model'])
    _check(sh, ref['controlled_unet'], 'model.diffusion_model.')
    _check(sh, ref['controlnet'], 'control_model.')
    assert any(k.startswith('first_stage_model.decoder.') for k in sh) and any(k.startswith('cond_stage_model.') for k in sh)


def test_engine_config_accepts_the_wrapped_node():
    """make_config(cfg) did cfg['in_channels'] on the {target, params} node and raised KeyError."""
    node = {'target': 'ldm.modules.diffusionmodules.openaimodel.UNetModel', 'params': UNET_PARAMS}
    a, b = engine.make_config(node), engine.make_config(engine.SD_V1)
    assert bytes(a) == bytes(b)

    class AttrDict(dict):                 # OmegaConf-like: mapping with attribute access
        __getattr__ = dict.__getitem__
    assert bytes(engine.make_config(AttrDict(target=node['target'], params=AttrDict(UNET_PARAMS)))) == bytes(b)


def test_unsupported_settings_are_refused_by_name():
    for bad in (dict(transformer_depth=2), dict(use_spatial_transformer=False), dict(num_head_channels=64),
                dict(use_scale_shift_norm=True), dict(resblock_updown=True), dict(made_up_option=1)):
        with pytest.raises((NotImplementedError, KeyError)):
            config.unet_params({'target': 'ldm.modules.diffusionmodules.openaimodel.UNetModel', 'params': dict(UNET_PARAMS, **bad)})
    with pytest.raises(NotImplementedError):
        models.ControlLDM.engine_args(unet_config=FGDM_YAML['model']['params']['unet_config'],
                                      control_stage_config={'target': 'controlnet.cldm.cldm.ControlNet',
                                                            'params': dict(UNET_PARAMS, model_channels=256, hint_channels=3)})
    with pytest.raises(KeyError):
        config.instantiate_from_config({'params': {}})
    assert config.instantiate_from_config('__is_first_stage__') is None


def test_dropin_module_map_resolves_the_scripts_imports():
    import fgdm_amd.dropin as dropin
    dropin.install()
    from ldm.util import instantiate_from_config                                  # scripts/txt2img_fgdm_inference.py:17
    from ldm.models.diffusion.ddim import DDIMSampler                             # :18
    from ldm.models.diffusion.plms import PLMSSampler                             # :19
    import controlnet.initialize_cn as initialize_cn                              # :25
    from cldm.model import create_model, load_state_dict                          # controlnet/seg2image_inference.py:18
    from cldm.ddim_hacked import DDIMSampler as CNSampler                         # :19
    from fgdm_amd import samplers
    assert instantiate_from_config is config.instantiate_from_config and create_model is config.create_model
    assert DDIMSampler is samplers.DDIMSampler and CNSampler is samplers.ControlDDIMSampler
    assert callable(initialize_cn.process) and callable(load_state_dict) and PLMSSampler is samplers.PLMSSampler


def test_yaml_files_round_trip(tmp_path):
    """create_model(path): the YAML is read with OmegaConf when present, plain yaml otherwise."""
    import yaml
    p = tmp_path / 'cldm.yaml'
    p.write_text(yaml.safe_dump(CLDM_YAML))
    cfg = config.load_config(str(p))
    assert cfg['model']['target'] == 'controlnet.cldm.cldm.ControlLDM'
    args = models.ControlLDM.engine_args(**config.split(cfg['model'])[1])
    assert args['n_controlnets'] == 1 and args['use_adapter'] is False and args['vae'] and args['clip'] is True
