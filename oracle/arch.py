"""Architecture tables (block list + state-dict keys) for the oracle.  TEST INFRASTRUCTURE.

Restates the constructor logic of
  * ldm/modules/diffusionmodules/openaimodel.py:469-734  (UNetModel.__init__)
  * ldm/modules/encoders/adapter.py:316-333              (Adapter.__init__)
  * controlnet/cldm/cldm.py:545-790                      (ControlNet.__init__)
as plain data so the functional forward in ``oracle/nn.py`` can walk it and so
the parameter names/shapes can be compared with the reference's state_dict
(golden: tests/golden/param_keys.json).
"""
from collections import OrderedDict

SD_UNET = dict(in_channels=4, out_channels=4, model_channels=320,
               attention_resolutions=(4, 2, 1), num_res_blocks=2,
               channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768,
               transformer_depth=1)

ADAPTER_CHANNELS = (320, 640, 1280, 1280)   # hard-coded at openaimodel.py:554-556
HINT_CHANNELS = (16, 16, 32, 32, 96, 96, 256)  # cldm.py:655-671


def unet_blocks(cfg):
    """Return (input_blocks, middle, output_blocks): each block is a list of layers.

    layer = ('conv', cin, cout) | ('res', cin, cout) | ('attn', ch, heads, d_head)
          | ('down', ch) | ('up', ch)
    Mirrors openaimodel.py:558-718.
    """
    mc = cfg['model_channels']
    nrb = cfg['num_res_blocks']
    cm = cfg['channel_mult']
    heads = cfg['num_heads']
    ares = cfg['attention_resolutions']
    inp = [[('conv', cfg['in_channels'], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cm):
        for _ in range(nrb):
            layers = [('res', ch, mult * mc)]
            ch = mult * mc
            if ds in ares:
                layers.append(('attn', ch, heads, ch // heads))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            inp.append([('down', ch)])
            chans.append(ch)
            ds *= 2
    mid = [('res', ch, ch), ('attn', ch, heads, ch // heads), ('res', ch, ch)]
    out = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [('res', ch + ich, mc * mult)]
            ch = mc * mult
            if ds in ares:
                layers.append(('attn', ch, heads, ch // heads))
            if level and i == nrb:
                layers.append(('up', ch))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def _res_params(p, prefix, cin, cout, temb):
    p[prefix + 'in_layers.0.weight'] = (cin,)
    p[prefix + 'in_layers.0.bias'] = (cin,)
    p[prefix + 'in_layers.2.weight'] = (cout, cin, 3, 3)
    p[prefix + 'in_layers.2.bias'] = (cout,)
    p[prefix + 'emb_layers.1.weight'] = (cout, temb)
    p[prefix + 'emb_layers.1.bias'] = (cout,)
    p[prefix + 'out_layers.0.weight'] = (cout,)
    p[prefix + 'out_layers.0.bias'] = (cout,)
    p[prefix + 'out_layers.3.weight'] = (cout, cout, 3, 3)
    p[prefix + 'out_layers.3.bias'] = (cout,)
    if cin != cout:
        p[prefix + 'skip_connection.weight'] = (cout, cin, 1, 1)
        p[prefix + 'skip_connection.bias'] = (cout,)


def _attn_params(p, prefix, ch, ctx):
    inner = ch
    p[prefix + 'norm.weight'] = (ch,)
    p[prefix + 'norm.bias'] = (ch,)
    p[prefix + 'proj_in.weight'] = (inner, ch, 1, 1)
    p[prefix + 'proj_in.bias'] = (inner,)
    t = prefix + 'transformer_blocks.0.'
    p[t + 'attn1.to_q.weight'] = (inner, inner)
    p[t + 'attn1.to_k.weight'] = (inner, inner)
    p[t + 'attn1.to_v.weight'] = (inner, inner)
    p[t + 'attn1.to_out.0.weight'] = (inner, inner)
    p[t + 'attn1.to_out.0.bias'] = (inner,)
    p[t + 'ff.net.0.proj.weight'] = (8 * inner, inner)
    p[t + 'ff.net.0.proj.bias'] = (8 * inner,)
    p[t + 'ff.net.2.weight'] = (inner, 4 * inner)
    p[t + 'ff.net.2.bias'] = (inner,)
    p[t + 'attn2.to_q.weight'] = (inner, inner)
    p[t + 'attn2.to_k.weight'] = (inner, ctx)
    p[t + 'attn2.to_v.weight'] = (inner, ctx)
    p[t + 'attn2.to_out.0.weight'] = (inner, inner)
    p[t + 'attn2.to_out.0.bias'] = (inner,)
    for n in ('norm1', 'norm2', 'norm3'):
        p[t + n + '.weight'] = (inner,)
        p[t + n + '.bias'] = (inner,)
    p[prefix + 'proj_out.weight'] = (ch, inner, 1, 1)
    p[prefix + 'proj_out.bias'] = (ch,)


def _block_params(p, prefix, layers, temb, ctx):
    for j, l in enumerate(layers):
        lp = f'{prefix}{j}.'
        if l[0] == 'conv':
            p[lp + 'weight'] = (l[2], l[1], 3, 3)
            p[lp + 'bias'] = (l[2],)
        elif l[0] == 'res':
            _res_params(p, lp, l[1], l[2], temb)
        elif l[0] == 'attn':
            _attn_params(p, lp, l[1], ctx)
        elif l[0] == 'down':
            p[lp + 'op.weight'] = (l[1], l[1], 3, 3)
            p[lp + 'op.bias'] = (l[1],)
        elif l[0] == 'up':
            p[lp + 'conv.weight'] = (l[1], l[1], 3, 3)
            p[lp + 'conv.bias'] = (l[1],)


def adapter_blocks(cin=4, channels=ADAPTER_CHANNELS, nums_rb=2):
    """[(in_c, out_c, down)] per body block; adapter.py:322-331."""
    body = []
    for i in range(len(channels)):
        for j in range(nums_rb):
            if i != 0 and j == 0:
                body.append((channels[i - 1], channels[i], True))
            else:
                body.append((channels[i], channels[i], False))
    return body


def adapter_param_shapes(cin=4, channels=ADAPTER_CHANNELS, nums_rb=2, prefix='adapter.'):
    """Keys in module-registration order (adapter.py:280-333, ksize=1, sk=True)."""
    p = OrderedDict()
    for k, (ic, oc, down) in enumerate(adapter_blocks(cin, channels, nums_rb)):
        b = f'{prefix}body.{k}.'
        if ic != oc:
            p[b + 'in_conv.weight'] = (oc, ic, 1, 1)
            p[b + 'in_conv.bias'] = (oc,)
        p[b + 'block1.weight'] = (oc, oc, 3, 3)
        p[b + 'block1.bias'] = (oc,)
        p[b + 'block2.weight'] = (oc, oc, 1, 1)
        p[b + 'block2.bias'] = (oc,)
    p[prefix + 'conv_in.weight'] = (channels[0], cin, 3, 3)
    p[prefix + 'conv_in.bias'] = (channels[0],)
    return p


def time_adapter_param_shapes(cin=4, channels=ADAPTER_CHANNELS, nums_rb=2, temb=1280, prefix='adapter.'):
    """TimeAdapter (adapter.py:387-403): openaimodel-style ResBlocks, `down=True` on the first block of levels > 0."""
    p = OrderedDict()
    for k, (ic, oc, down) in enumerate(adapter_blocks(cin, channels, nums_rb)):
        _res_params(p, f'{prefix}body.{k}.', ic, oc, temb)
    p[prefix + 'conv_in.weight'] = (channels[0], cin, 3, 3)
    p[prefix + 'conv_in.bias'] = (channels[0],)
    return p


def unet_param_shapes(cfg, adapter=True, prefix='', num_prompts=1):
    """State-dict keys -> shapes of the reference UNetModel (adapter: False | True (Adapter) | 'time' (TimeAdapter));
    num_prompts > 1: AdaptUNetModel (openaimodel.py:993-999) with `adapters.{k}` registered right after `adapter`."""
    mc = cfg['model_channels']
    temb = 4 * mc
    ctx = cfg['context_dim']
    inp, mid, out = unet_blocks(cfg)
    p = OrderedDict()
    p[prefix + 'time_embed.0.weight'] = (temb, mc)
    p[prefix + 'time_embed.0.bias'] = (temb,)
    p[prefix + 'time_embed.2.weight'] = (temb, temb)
    p[prefix + 'time_embed.2.bias'] = (temb,)
    if adapter == 'time':
        p.update(time_adapter_param_shapes(cfg['in_channels'], temb=temb, prefix=prefix + 'adapter.'))
    elif adapter:
        p.update(adapter_param_shapes(cfg['in_channels'], prefix=prefix + 'adapter.'))
        for kk in range(num_prompts - 1):
            p.update(adapter_param_shapes(cfg['in_channels'], prefix=f'{prefix}adapters.{kk}.'))
    for i, layers in enumerate(inp):
        _block_params(p, f'{prefix}input_blocks.{i}.', layers, temb, ctx)
    _block_params(p, f'{prefix}middle_block.', mid, temb, ctx)
    for i, layers in enumerate(out):
        _block_params(p, f'{prefix}output_blocks.{i}.', layers, temb, ctx)
    p[prefix + 'out.0.weight'] = (mc,)
    p[prefix + 'out.0.bias'] = (mc,)
    p[prefix + 'out.2.weight'] = (cfg['out_channels'], mc, 3, 3)
    p[prefix + 'out.2.bias'] = (cfg['out_channels'],)
    return p


def controlnet_param_shapes(cfg, hint_channels=3, prefix=''):
    """State-dict keys -> shapes of the reference ControlNet (cldm.py:545-790)."""
    mc = cfg['model_channels']
    temb = 4 * mc
    ctx = cfg['context_dim']
    inp, mid, _ = unet_blocks(cfg)
    p = OrderedDict()
    p[prefix + 'time_embed.0.weight'] = (temb, mc)
    p[prefix + 'time_embed.0.bias'] = (temb,)
    p[prefix + 'time_embed.2.weight'] = (temb, temb)
    p[prefix + 'time_embed.2.bias'] = (temb,)
    for i, layers in enumerate(inp):
        _block_params(p, f'{prefix}input_blocks.{i}.', layers, temb, ctx)
    for i, layers in enumerate(inp):
        ch = layers[0][2] if layers[0][0] in ('conv', 'res') else layers[0][1]
        p[f'{prefix}zero_convs.{i}.0.weight'] = (ch, ch, 1, 1)
        p[f'{prefix}zero_convs.{i}.0.bias'] = (ch,)
    chs = (hint_channels,) + HINT_CHANNELS + (mc,)
    for k in range(8):
        p[f'{prefix}input_hint_block.{2 * k}.weight'] = (chs[k + 1], chs[k], 3, 3)
        p[f'{prefix}input_hint_block.{2 * k}.bias'] = (chs[k + 1],)
    _block_params(p, f'{prefix}middle_block.', mid, temb, ctx)
    ch = mid[0][2]
    p[prefix + 'middle_block_out.0.weight'] = (ch, ch, 1, 1)
    p[prefix + 'middle_block_out.0.bias'] = (ch,)
    return p
