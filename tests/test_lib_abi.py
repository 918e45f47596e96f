"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/fgdm.h declares, and the
engine's parameter table matches the reference state_dict keys/shapes (golden param_keys.json).
No compute calls: these run without a GPU."""
import json
import os
import re

import pytest

import golden_inputs as gi
from common import GOLD
from fgdm_amd import _lib, engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from fgdm_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'fgdm.h')).read()
    declared = set(re.findall(r'\b(fgdm_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def _strip(d, prefix):
    return {k[len(prefix):]: tuple(v) for k, v in d.items() if k.startswith(prefix)}


def test_param_table_matches_reference_keys(lib):
    ref = json.load(open(os.path.join(GOLD, 'param_keys.json')))
    # FG-DM UNet (with adapter) + one ControlNet
    got = eng.param_shapes(eng.make_config(gi.SD_CFG, use_adapter=True, n_controlnets=1))
    unet = _strip(got, 'model.diffusion_model.')
    assert list(unet.keys()) == list(ref['unet_fgdm'].keys())
    assert all(tuple(ref['unet_fgdm'][k]) == v for k, v in unet.items())
    cn = _strip(got, 'control_model.')
    assert list(cn.keys()) == list(ref['controlnet'].keys())
    assert all(tuple(ref['controlnet'][k]) == v for k, v in cn.items())
    # TimeAdapter variant (use_time_adapter=True)
    got = eng.param_shapes(eng.make_config(gi.SD_CFG, use_adapter='time'))
    unet = _strip(got, 'model.diffusion_model.')
    assert list(unet.keys()) == list(ref['unet_time_adapter'].keys())
    assert all(tuple(ref['unet_time_adapter'][k]) == v for k, v in unet.items())
    # AdaptUNetModel(num_prompts=3): two further adapters registered right after `adapter`
    got = eng.param_shapes(eng.make_config(gi.SD_CFG, use_adapter=True, num_prompts=3))
    unet = _strip(got, 'model.diffusion_model.')
    assert list(unet.keys()) == list(ref['adapt_unet_3'].keys())
    assert all(tuple(ref['adapt_unet_3'][k]) == v for k, v in unet.items())
    # plain SD UNet (= ControlledUnetModel keys)
    got = eng.param_shapes(eng.make_config(gi.SD_CFG))
    unet = _strip(got, 'model.diffusion_model.')
    assert list(unet.keys()) == list(ref['controlled_unet'].keys())
    # reduced-depth config
    got = eng.param_shapes(eng.make_config(gi.SMALL_CFG, n_controlnets=1))
    assert list(_strip(got, 'model.diffusion_model.').keys()) == list(ref['unet_small'].keys())
    assert list(_strip(got, 'control_model.').keys()) == list(ref['controlnet_small'].keys())
    # first-stage decoder keys follow the UNet / ControlNet tables (AutoencoderKL decoder.* + post_quant_conv.*)
    got = eng.param_shapes(eng.make_config(gi.SD_CFG, vae=True))
    vae = {k: v for k, v in got.items() if k.startswith('first_stage_model.')}
    assert list(vae.keys()) == list(ref['vae_decoder'].keys())
    assert all(tuple(ref['vae_decoder'][k]) == v for k, v in vae.items())
    assert list(got.keys())[-len(vae):] == list(vae.keys())
    # text-encoder keys (transformers.CLIPTextModel under the checkpoints' prefix) come last
    got = eng.param_shapes(eng.make_config(gi.SD_CFG, vae=True, clip=True))
    ck = json.load(open(os.path.join(GOLD, 'clip_keys.json')))['keys']
    clip = {k: v for k, v in got.items() if k.startswith('cond_stage_model.')}
    assert list(clip.keys()) == list(ck.keys())
    assert all(tuple(ck[k]) == v for k, v in clip.items())
    assert list(got.keys())[-len(clip):] == list(clip.keys())
    # several ControlNets get distinct prefixes
    got = eng.param_shapes(eng.make_config(gi.SD_CFG, n_controlnets=3))
    assert any(k.startswith('control_model_1.') for k in got) and any(k.startswith('control_model_2.') for k in got)


def test_unsupported_configs_are_rejected(lib):
    bad = dict(gi.SD_CFG, model_channels=160)          # not a multiple of 64
    with pytest.raises(ValueError):
        eng.param_shapes(eng.make_config(bad))
    with pytest.raises(ValueError):                     # decoder attention at up levels is not implemented
        eng.make_config(gi.SD_CFG, vae=dict(eng.SD_VAE, attn_resolutions=[32]))
    with pytest.raises(ValueError):                     # decoder width must be a multiple of 64
        eng.param_shapes(eng.make_config(gi.SD_CFG, vae=dict(eng.SD_VAE, ch=96)))
    with pytest.raises(ValueError):                     # extra adapters need the FG-DM adapter
        eng.param_shapes(eng.make_config(gi.SD_CFG, use_adapter=False, num_prompts=2))
    with pytest.raises(ValueError):                     # text encoder: head dim must be 64
        eng.param_shapes(eng.make_config(gi.SD_CFG, clip=dict(eng.SD_CLIP, num_attention_heads=8)))
    with pytest.raises(ValueError):                     # adapter needs the SD-v1 topology
        eng.param_shapes(eng.make_config(gi.SMALL_CFG, use_adapter=True))


def test_engine_needs_gpu_no_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        eng.Engine(gi.SMALL_CFG)
