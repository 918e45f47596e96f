"""First-stage (AutoencoderKL) decoder  --  CPU oracle, TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, as a functional fp32 torch-CPU forward over a reference-keyed parameter dict,
  * AutoencoderKL.decode                       ldm/models/autoencoder.py:330-333
  * Decoder.__init__ / Decoder.forward          ldm/modules/diffusionmodules/model.py:462-560
  * ResnetBlock.forward (temb=None)             model.py:121-141
  * AttnBlock.forward (single head, d = C)      model.py:176-203
  * Upsample.forward (nearest 2x + conv3x3)     model.py:53-57
  * Normalize = GroupNorm(32, eps=1e-6)         model.py:38-39;  nonlinearity = swish  model.py:33-35
  * LatentDiffusion.decode_first_stage          ldm/models/diffusion/ddpm.py:832-889 (z / scale_factor, plain branch)
Pinned by tests/golden/vae.npz (tests/test_oracle_golden.py), produced by the reference's own AutoencoderKL.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

# models/config.yaml:50-69 (first_stage_config.ddconfig)
SD_VAE = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=(), z_channels=4,
              embed_dim=4, resolution=256, in_channels=3)
SCALE_FACTOR = 0.18215   # models/config.yaml scale_factor (LatentDiffusion.scale_factor)


def _res_shapes(p, pre, cin, cout):
    p[pre + 'norm1.weight'] = (cin,)
    p[pre + 'norm1.bias'] = (cin,)
    p[pre + 'conv1.weight'] = (cout, cin, 3, 3)
    p[pre + 'conv1.bias'] = (cout,)
    p[pre + 'norm2.weight'] = (cout,)
    p[pre + 'norm2.bias'] = (cout,)
    p[pre + 'conv2.weight'] = (cout, cout, 3, 3)
    p[pre + 'conv2.bias'] = (cout,)
    if cin != cout:
        p[pre + 'nin_shortcut.weight'] = (cout, cin, 1, 1)
        p[pre + 'nin_shortcut.bias'] = (cout,)


def decoder_levels(cfg):
    """[(level, [(cin, cout)] * (num_res_blocks + 1), has_upsample)] in execution order (model.py:498-518, 539-546)."""
    ch, mult, nrb = cfg['ch'], cfg['ch_mult'], cfg['num_res_blocks']
    block_in = ch * mult[-1]
    out = []
    for lvl in reversed(range(len(mult))):
        block_out = ch * mult[lvl]
        blocks = []
        for _ in range(nrb + 1):
            blocks.append((block_in, block_out))
            block_in = block_out
        out.append((lvl, blocks, lvl != 0))
    return out


def decoder_param_shapes(cfg=SD_VAE, prefix='first_stage_model.'):
    """State-dict keys (module registration order) of the decoder half of AutoencoderKL:
    decoder.* (model.py:486-530; `self.up.insert(0, up)` leaves the keys in ascending level order) then
    post_quant_conv (autoencoder.py:303)."""
    if cfg.get('attn_resolutions'):
        raise ValueError('decoder attention at up levels is not used by any shipped config')
    p = OrderedDict()
    d = prefix + 'decoder.'
    top = cfg['ch'] * cfg['ch_mult'][-1]
    p[d + 'conv_in.weight'] = (top, cfg['z_channels'], 3, 3)
    p[d + 'conv_in.bias'] = (top,)
    _res_shapes(p, d + 'mid.block_1.', top, top)
    for n in ('norm', 'q', 'k', 'v', 'proj_out'):
        p[d + f'mid.attn_1.{n}.weight'] = (top,) if n == 'norm' else (top, top, 1, 1)
        p[d + f'mid.attn_1.{n}.bias'] = (top,)
    _res_shapes(p, d + 'mid.block_2.', top, top)
    levels = {lvl: (blocks, up) for lvl, blocks, up in decoder_levels(cfg)}
    for lvl in range(len(cfg['ch_mult'])):
        blocks, up = levels[lvl]
        for i, (ci, co) in enumerate(blocks):
            _res_shapes(p, d + f'up.{lvl}.block.{i}.', ci, co)
        if up:
            c = blocks[-1][1]
            p[d + f'up.{lvl}.upsample.conv.weight'] = (c, c, 3, 3)
            p[d + f'up.{lvl}.upsample.conv.bias'] = (c,)
    c0 = cfg['ch'] * cfg['ch_mult'][0]
    p[d + 'norm_out.weight'] = (c0,)
    p[d + 'norm_out.bias'] = (c0,)
    p[d + 'conv_out.weight'] = (cfg['out_ch'], c0, 3, 3)
    p[d + 'conv_out.bias'] = (cfg['out_ch'],)
    p[prefix + 'post_quant_conv.weight'] = (cfg['z_channels'], cfg['embed_dim'], 1, 1)
    p[prefix + 'post_quant_conv.bias'] = (cfg['z_channels'],)
    return p


def _gn(x, p, name):
    return F.group_norm(x.float(), 32, p[name + '.weight'], p[name + '.bias'], 1e-6)


def _conv(x, p, name, padding=1):
    return F.conv2d(x, p[name + '.weight'], p[name + '.bias'], padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)


def resnet_block(p, pre, x):
    # model.py:121-141 with temb None, dropout 0
    h = _conv(_swish(_gn(x, p, pre + 'norm1')), p, pre + 'conv1')
    h = _conv(_swish(_gn(h, p, pre + 'norm2')), p, pre + 'conv2')
    if (pre + 'nin_shortcut.weight') in p:
        x = _conv(x, p, pre + 'nin_shortcut', padding=0)
    return x + h


def attn_block(p, pre, x):
    # model.py:176-203: one head over all C channels, softmax over keys
    h = _gn(x, p, pre + 'norm')
    q = _conv(h, p, pre + 'q', 0)
    k = _conv(h, p, pre + 'k', 0)
    v = _conv(h, p, pre + 'v', 0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, p, pre + 'proj_out', 0)


def decoder_forward(p, z, cfg=SD_VAE, prefix='first_stage_model.'):
    d = prefix + 'decoder.'
    h = _conv(z, p, d + 'conv_in')
    h = resnet_block(p, d + 'mid.block_1.', h)
    h = attn_block(p, d + 'mid.attn_1.', h)
    h = resnet_block(p, d + 'mid.block_2.', h)
    for lvl, blocks, up in decoder_levels(cfg):
        for i in range(len(blocks)):
            h = resnet_block(p, d + f'up.{lvl}.block.{i}.', h)
        if up:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = _conv(h, p, d + f'up.{lvl}.upsample.conv')
    return _conv(_swish(_gn(h, p, d + 'norm_out')), p, d + 'conv_out')


def decode(p, z, cfg=SD_VAE, prefix='first_stage_model.'):
    """AutoencoderKL.decode (autoencoder.py:330-333)."""
    return decoder_forward(p, _conv(z, p, prefix + 'post_quant_conv', 0), cfg, prefix)


def decode_first_stage(p, z, scale_factor=SCALE_FACTOR, cfg=SD_VAE, prefix='first_stage_model.'):
    """LatentDiffusion.decode_first_stage (ddpm.py:839, 889)."""
    return decode(p, (1.0 / scale_factor) * z, cfg, prefix)
