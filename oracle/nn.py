"""Functional fp32 CPU forward of the denoiser networks.  TEST INFRASTRUCTURE.

Plain ``torch.nn.functional`` on CPU tensors, parameters looked up by the
reference's state-dict key names.  Restates:
  * timestep_embedding        ldm/modules/diffusionmodules/util.py:160-180
  * ResBlock._forward         ldm/modules/diffusionmodules/openaimodel.py:275-301
  * Downsample / Upsample     openaimodel.py:171-180 / :114-130
  * SpatialTransformer        ldm/modules/attention.py:275-292
  * BasicTransformerBlock     attention.py:234-240
  * CrossAttention            attention.py:177-216  (tuple's 2nd element is discarded by callers)
  * GEGLU / FeedForward       attention.py:37-64    (exact-erf GELU)
  * UNetModel.forward / forward_original   openaimodel.py:808-884 / :753-806
  * Adapter / ResnetBlock     ldm/modules/encoders/adapter.py:334-346 / :301-313
  * ControlNet.forward        controlnet/cldm/cldm.py:792-813
  * ControlledUnetModel.forward / ControlLDM.apply_model   cldm.py:27-50 / :836-849
"""
import math
import torch
import torch.nn.functional as F

from . import arch


def timestep_embedding(t, dim, max_period=10000):
    # util.py:171-175: cos first, then sin
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, p, name, eps):
    return F.group_norm(x.float(), 32, p[name + '.weight'], p[name + '.bias'], eps)


def _conv(x, p, name, stride=1, padding=1):
    return F.conv2d(x, p[name + '.weight'], p[name + '.bias'], stride=stride, padding=padding)


def _lin(x, p, name, bias=True):
    return F.linear(x, p[name + '.weight'], p[name + '.bias'] if bias else None)


def time_embed(p, prefix, t, mc):
    e = timestep_embedding(t, mc)
    e = _lin(e, p, prefix + 'time_embed.0')
    e = F.silu(e)
    return _lin(e, p, prefix + 'time_embed.2')


def resblock(p, pre, x, emb, down=False):
    # openaimodel.py:275-301 (no scale-shift norm, dropout p=0); down=True: AvgPool2d(2) on h and x (:276-282, use_conv=False)
    h = F.silu(_gn(x, p, pre + 'in_layers.0', 1e-5))
    if down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(h, p, pre + 'in_layers.2')
    e = _lin(F.silu(emb), p, pre + 'emb_layers.1')
    h = h + e[:, :, None, None]
    h = _conv(F.silu(_gn(h, p, pre + 'out_layers.0', 1e-5)), p, pre + 'out_layers.3')
    if (pre + 'skip_connection.weight') in p:
        x = _conv(x, p, pre + 'skip_connection', padding=0)
    return x + h


def attention(p, pre, x, ctx, heads):
    # attention.py:177-202 ; softmax(q k^T d^-1/2) v ; to_out has a bias, q/k/v do not
    q = _lin(x, p, pre + 'to_q', bias=False)
    ctx = x if ctx is None else ctx
    k = _lin(ctx, p, pre + 'to_k', bias=False)
    v = _lin(ctx, p, pre + 'to_v', bias=False)
    b, n, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    o = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(o, p, pre + 'to_out.0')


def transformer_block(p, pre, x, ctx, heads):
    # attention.py:234-240
    x = attention(p, pre + 'attn1.', F.layer_norm(x, x.shape[-1:], p[pre + 'norm1.weight'], p[pre + 'norm1.bias'], 1e-5), None, heads) + x
    x = attention(p, pre + 'attn2.', F.layer_norm(x, x.shape[-1:], p[pre + 'norm2.weight'], p[pre + 'norm2.bias'], 1e-5), ctx, heads) + x
    h = F.layer_norm(x, x.shape[-1:], p[pre + 'norm3.weight'], p[pre + 'norm3.bias'], 1e-5)
    h = _lin(h, p, pre + 'ff.net.0.proj')
    a, g = h.chunk(2, dim=-1)
    h = a * F.gelu(g)                                   # exact erf GELU, attention.py:43-44
    return _lin(h, p, pre + 'ff.net.2') + x


def spatial_transformer(p, pre, x, ctx, heads):
    # attention.py:275-292 ; GroupNorm eps 1e-6 (attention.py:76-77)
    b, c, hh, ww = x.shape
    x_in = x
    x = _gn(x, p, pre + 'norm', 1e-6)
    x = _conv(x, p, pre + 'proj_in', padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    x = transformer_block(p, pre + 'transformer_blocks.0.', x, ctx, heads)
    x = x.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    x = _conv(x, p, pre + 'proj_out', padding=0)
    return x + x_in


def run_block(p, pre, layers, h, emb, ctx):
    for j, l in enumerate(layers):
        lp = f'{pre}{j}.'
        if l[0] == 'conv':
            h = F.conv2d(h, p[lp + 'weight'], p[lp + 'bias'], padding=1)
        elif l[0] == 'res':
            h = resblock(p, lp, h, emb)
        elif l[0] == 'attn':
            h = spatial_transformer(p, lp, h, ctx, l[2])
        elif l[0] == 'down':
            h = F.conv2d(h, p[lp + 'op.weight'], p[lp + 'op.bias'], stride=2, padding=1)
        elif l[0] == 'up':
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            h = F.conv2d(h, p[lp + 'conv.weight'], p[lp + 'conv.bias'], padding=1)
    return h


def adapter_forward(p, pre, x, cin=4):
    # adapter.py:334-346 + ResnetBlock.forward :301-313 (ksize=1, sk=True, use_conv=False)
    feats = []
    x = F.conv2d(x, p[pre + 'conv_in.weight'], p[pre + 'conv_in.bias'], padding=1)
    body = arch.adapter_blocks(cin)
    nlev = len(arch.ADAPTER_CHANNELS)
    nrb = len(body) // nlev
    for i in range(nlev):
        for j in range(nrb):
            k = i * nrb + j
            ic, oc, down = body[k]
            b = f'{pre}body.{k}.'
            if down:
                x = F.avg_pool2d(x, kernel_size=2, stride=2)
            if ic != oc:
                x = F.conv2d(x, p[b + 'in_conv.weight'], p[b + 'in_conv.bias'])
            h = F.conv2d(x, p[b + 'block1.weight'], p[b + 'block1.bias'], padding=1)
            h = F.relu(h)
            h = F.conv2d(h, p[b + 'block2.weight'], p[b + 'block2.bias'])
            x = h + x
        feats.append(x)
    return feats


def time_adapter_forward(p, pre, x, emb, cin=4):
    # adapter.py:405-417
    feats = []
    x = F.conv2d(x, p[pre + 'conv_in.weight'], p[pre + 'conv_in.bias'], padding=1)
    body = arch.adapter_blocks(cin)
    for k, (ic, oc, down) in enumerate(body):
        x = resblock(p, f'{pre}body.{k}.', x, emb, down=down)
        if k % 2 == 1:
            feats.append(x)
    return feats


def unet_forward(p, cfg, x, t, ctx, prefix='', use_adapter=False, pcond=None,
                 control=None, only_mid_control=False, conds=None):
    """eps = UNet(x, t, ctx).

    use_adapter=False, control=None : UNetModel.forward_original (openaimodel.py:753-806)
    use_adapter=True                : UNetModel.forward with FG-DM adapter (openaimodel.py:808-884);
                                      feature k is added after input block 3k+2 *before* the skip push
    control=[13 tensors]            : ControlledUnetModel.forward (cldm.py:27-50); list is consumed from the end
    conds=[tensors]                 : AdaptUNetModel.forward (openaimodel.py:1263-1320, num_prompts = len(conds) + 1):
                                      `adapters.{k}(conds[k])` features are summed onto the `adapter(prompt)` features
                                      (prompt = pcond, the reference's `control` argument, or x)
    """
    inp, mid, out = arch.unet_blocks(cfg)
    emb = time_embed(p, prefix, t, cfg['model_channels'])
    h = x.float()
    fa = None
    if use_adapter == 'time':      # use_time_adapter=True: fa = self.adapter(prompt, emb)  (openaimodel.py:843-844)
        fa = time_adapter_forward(p, prefix + 'adapter.', h if pcond is None else pcond, emb, cfg['in_channels'])
    elif use_adapter:
        fa = adapter_forward(p, prefix + 'adapter.', h if pcond is None else pcond, cfg['in_channels'])
    if conds is not None:
        for kdx, cond in enumerate(conds):
            fk = adapter_forward(p, f'{prefix}adapters.{kdx}.', cond, cfg['in_channels'])
            fa = [a + b for a, b in zip(fa, fk)]
    hs = []
    k = 0
    for i, layers in enumerate(inp):
        h = run_block(p, f'{prefix}input_blocks.{i}.', layers, h, emb, ctx)
        if fa is not None and (i + 1) % 3 == 0:
            h = h + fa[k]
            k += 1
        hs.append(h)
    if fa is not None:
        assert k == len(fa)
    h = run_block(p, f'{prefix}middle_block.', mid, h, emb, ctx)
    if control is not None:
        control = list(control)
        h = h + control.pop()
    for i, layers in enumerate(out):
        if control is None or only_mid_control:
            h = torch.cat([h, hs.pop()], dim=1)
        else:
            h = torch.cat([h, hs.pop() + control.pop()], dim=1)
        h = run_block(p, f'{prefix}output_blocks.{i}.', layers, h, emb, ctx)
    h = F.silu(_gn(h, p, prefix + 'out.0', 1e-5))
    return F.conv2d(h, p[prefix + 'out.2.weight'], p[prefix + 'out.2.bias'], padding=1)


def hint_block(p, prefix, hint):
    # cldm.py:655-671: 8 conv3x3, SiLU between, stride 2 at convs 2,4,6 (0-based)
    h = hint.float()
    for k in range(8):
        stride = 2 if k in (2, 4, 6) else 1
        h = F.conv2d(h, p[f'{prefix}input_hint_block.{2 * k}.weight'],
                     p[f'{prefix}input_hint_block.{2 * k}.bias'], stride=stride, padding=1)
        if k != 7:
            h = F.silu(h)
    return h


def controlnet_forward(p, cfg, x, hint, t, ctx, prefix=''):
    """13 control residuals; cldm.py:792-813."""
    inp, mid, _ = arch.unet_blocks(cfg)
    emb = time_embed(p, prefix, t, cfg['model_channels'])
    guided = hint_block(p, prefix, hint)
    outs = []
    h = x.float()
    for i, layers in enumerate(inp):
        h = run_block(p, f'{prefix}input_blocks.{i}.', layers, h, emb, ctx)
        if guided is not None:
            h = h + guided
            guided = None
        outs.append(F.conv2d(h, p[f'{prefix}zero_convs.{i}.0.weight'], p[f'{prefix}zero_convs.{i}.0.bias']))
    h = run_block(p, f'{prefix}middle_block.', mid, h, emb, ctx)
    outs.append(F.conv2d(h, p[prefix + 'middle_block_out.0.weight'], p[prefix + 'middle_block_out.0.bias']))
    return outs


def control_ldm_apply(p, cfg, x, t, ctx, hints, scales=None,
                      unet_prefix='model.diffusion_model.', cn_prefixes=('control_model.',)):
    """ControlLDM.apply_model (cldm.py:836-849).

    ``hints`` is a list with one hint per ControlNet; with several ControlNets
    (BASELINE configs 4/5 -- not in the reference) the 13-tensor
    residual lists are summed element-wise before ControlledUnetModel.
    hints=None -> plain UNet (c_concat None branch, cldm.py:842-843).
    """
    if hints is None:
        return unet_forward(p, cfg, x, t, ctx, prefix=unet_prefix)
    total = None
    for hint, cp in zip(hints, cn_prefixes):
        ctrl = controlnet_forward(p, cfg, x, hint, t, ctx, prefix=cp)
        sc = scales if scales is not None else [1.0] * 13
        ctrl = [c * s for c, s in zip(ctrl, sc)]
        total = ctrl if total is None else [a + b for a, b in zip(total, ctrl)]
    return unet_forward(p, cfg, x, t, ctx, prefix=unet_prefix, control=total)
