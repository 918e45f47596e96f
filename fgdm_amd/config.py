"""The reference's config plumbing for the sampling path: `{target: dotted.path, params: {...}}` nodes (OmegaConf or plain
dicts), `instantiate_from_config` (ldm/util.py:78-93) and `create_model` (controlnet/cldm/model.py:24-28).

The scripts build their models with `instantiate_from_config(config.model)` (scripts/txt2img_fgdm_inference.py:23-38) and
`create_model('./models/cldm_v15_canny.yaml')` (controlnet/initialize_cn.py:25-43, controlnet/seg2image_inference.py:36-40);
here the model targets of the shipped YAML files resolve to the MI355X mirrors, and the nested unet / control / first-stage
/ cond-stage nodes are read for their hyper-parameters (the engine implements exactly those module types)."""
import importlib
from collections.abc import Mapping

# reference model classes -> mirrors (resolved lazily: fgdm_amd.models imports this module)
_MODEL_TARGETS = {
    'ldm.models.diffusion.ddpm.LatentDiffusion': 'LatentDiffusion',
    'controlnet.cldm.cldm.ControlLDM': 'ControlLDM',
    'cldm.cldm.ControlLDM': 'ControlLDM',
}
_UNET_TARGETS = {
    'ldm.modules.diffusionmodules.openaimodel.UNetModel': 'unet',
    'ldm.modules.diffusionmodules.openaimodel.AdaptUNetModel': 'adapt',
    'controlnet.cldm.cldm.ControlledUnetModel': 'controlled',
    'cldm.cldm.ControlledUnetModel': 'controlled',
    'controlnet.cldm.cldm.ControlNet': 'controlnet',
    'cldm.cldm.ControlNet': 'controlnet',
}
# UNetModel / ControlNet kwargs the engine implements with exactly one value (the shipped configs')
_FIXED = {'use_spatial_transformer': True, 'transformer_depth': 1, 'legacy': False, 'dims': 2, 'num_classes': None,
          'use_scale_shift_norm': False, 'resblock_updown': False, 'conv_resample': True, 'use_new_attention_order': False,
          'num_head_channels': -1, 'num_heads_upsample': -1, 'n_embed': None, 'dropout': 0, 'use_fp16': False,
          'disable_self_attentions': None, 'num_attention_blocks': None, 'disable_middle_self_attn': False,
          'use_linear_in_transformer': False}
# ... and the ones without numerical meaning on the sampling path
_IGNORED = {'image_size', 'use_checkpoint', 'hint_channels', 'no_prompting', 'use_time_adapter', 'num_prompts',
            'return_conds', 'distill', 'freeze_backbone'}
_NET_KEYS = ('in_channels', 'out_channels', 'model_channels', 'attention_resolutions', 'num_res_blocks', 'channel_mult',
             'num_heads', 'context_dim')


def to_dict(node):
    """OmegaConf DictConfig / Mapping / object-with-keys -> plain nested python containers."""
    if node is None:
        return None
    try:
        from omegaconf import OmegaConf
        if OmegaConf.is_config(node):
            return OmegaConf.to_container(node, resolve=True)
    except Exception:
        pass
    if isinstance(node, Mapping):
        return {k: to_dict(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)) or type(node).__name__ == 'ListConfig':
        return [to_dict(v) for v in node]
    return node


def split(node):
    """-> (target or None, params dict) for a `{target, params}` node or a bare params dict."""
    d = to_dict(node)
    if isinstance(d, dict) and 'target' in d:
        return d['target'], dict(d.get('params') or {})
    return None, dict(d or {})


def unet_params(node, default=None):
    """Hyper-parameters of a UNetModel / ControlledUnetModel / ControlNet node (models/config.yaml:33-48,
    controlnet/models/cldm_v15_canny.yaml:21-53) -> (kind, engine cfg dict, remaining flags).  Anything the engine does
    not implement is refused by name instead of being ignored."""
    if node is None:
        return None, dict(default) if default is not None else None, {}
    target, p = split(node)
    kind = _UNET_TARGETS.get(target) if target is not None else None
    if target is not None and kind is None:
        raise NotImplementedError(f'unet_config / control_stage_config target {target}: not a module of the sampling path')
    for k, want in _FIXED.items():
        if k in p and p[k] != want and not (want in (0, None, False) and not p[k]):
            raise NotImplementedError(f'{k}={p[k]!r}: the engine implements {k}={want!r} (the shipped configs)')
    unknown = [k for k in p if k not in _FIXED and k not in _IGNORED and k not in _NET_KEYS]
    if unknown:
        raise NotImplementedError(f'unsupported UNet parameters {unknown}')
    missing = [k for k in _NET_KEYS if k not in p and not (k == 'out_channels' and kind == 'controlnet')]
    if missing:
        raise KeyError(f'UNet config lacks {missing}')
    cfg = {k: (tuple(p[k]) if isinstance(p[k], (list, tuple)) else p[k]) for k in _NET_KEYS if k in p}
    cfg.setdefault('out_channels', cfg['in_channels'])
    flags = {k: p[k] for k in ('no_prompting', 'use_time_adapter', 'num_prompts', 'hint_channels') if k in p}
    return kind, cfg, flags


def instantiate_from_config(config, **overrides):
    """ldm/util.py:78-93.  Model targets of the shipped configs resolve to the mirrors in fgdm_amd.models; anything else is
    imported and called exactly as the reference does."""
    if isinstance(config, str):
        if config in ('__is_first_stage__', '__is_unconditional__'):
            return None
        raise KeyError('Expected key `target` to instantiate.')
    target, params = split(config)
    if target is None:
        raise KeyError('Expected key `target` to instantiate.')
    params.update(overrides)
    if target in _MODEL_TARGETS:
        from . import models
        return getattr(models, _MODEL_TARGETS[target])(**params)
    module, cls = target.rsplit('.', 1)
    return getattr(importlib.import_module(module), cls)(**params)


def load_config(path_or_node):
    """A YAML path (OmegaConf.load in the reference; plain yaml here when omegaconf is absent) or a ready node -> dict."""
    if isinstance(path_or_node, (str, bytes)) or hasattr(path_or_node, '__fspath__'):
        try:
            from omegaconf import OmegaConf
            return to_dict(OmegaConf.load(path_or_node))
        except ImportError:
            import yaml
            with open(path_or_node) as f:
                return yaml.safe_load(f)
    return to_dict(path_or_node)


def create_model(config_path, **overrides):
    """controlnet/cldm/model.py:24-28: instantiate_from_config(OmegaConf.load(config_path).model).cpu()"""
    cfg = load_config(config_path)
    model = instantiate_from_config(cfg['model'] if 'model' in cfg else cfg, **overrides).cpu()
    print(f'Loaded model config from [{config_path}]')
    return model
